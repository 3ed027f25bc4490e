"""Host-side helpers of bench.py (no GPU): the byte / FLOP models, the fingerprint that ties a PMC traffic
record to the kernel sources, the CPU count the native helpers use."""
import json
import os
import re
import warnings

import bench
from python_stable_3d_truss_analysis_amd import generate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_source_fingerprint_and_traffic_record():
    sha = bench.kernel_source_sha()
    assert re.fullmatch(r"[0-9a-f]{16}", sha) and sha == bench.kernel_source_sha()
    path = os.path.join(ROOT, "profiles", "potrf_traffic.json")
    with open(path) as fh:
        rec = json.load(fh)
    for key in ("kernel", "batch", "envelope", "joint_order", "hbm_bytes_per_launch", "source_sha", "source"):
        assert key in rec, key
    assert rec["hbm_bytes_per_launch"] > 0 and rec["kernel"].startswith("trs_potrf")
    if rec["source_sha"] != sha:   # bench.py then reports traffic: null; the record wants a new PMC pass
        warnings.warn("profiles/potrf_traffic.json was measured on other kernel sources (tools/profile_pmc.sh)")


def test_byte_and_flop_models():
    n, nJ, nM = 696, 244, 942
    dense = bench.algorithmic_counts(n, nJ, nM)
    assert dense["potrf_flops"] == n ** 3 / 3 + 2 * n ** 2
    assert dense["assemble_bytes_full_contract"] > dense["assemble_bytes"] > 0
    # the executed tile work of the dense factorisation is at least the textbook count (tiles are padded)
    assert bench.potrf_tile_flops(n) >= dense["potrf_flops"]
    # a diagonal-only envelope: one tile per chunk, factorisation work = the 16x16 factorisations alone
    nch = 704 // 16
    ft, last, cend = list(range(nch)), [4 * j + 3 for j in range(nch // 4)], [t + 1 for t in range(nch)]
    narrow = bench.algorithmic_counts(n, nJ, nM, cend, narrow=True)
    assert narrow["potrf_bytes"] < dense["potrf_bytes"] / 10
    assert bench.potrf_tile_flops(n, ft, last, cend, narrow=True) < bench.potrf_tile_flops(n) / 50


def test_vectorised_envelope_counts_equal_the_scalar_models():
    """`bench.envelope_counts_batch` (whole buckets at once, the cube-batch roofline) against the per-truss
    `potrf_tile_flops` / `algorithmic_counts` on random consistent narrow envelopes."""
    import numpy as np
    rng = np.random.default_rng(0)
    rows_pad = 896
    nchm = rows_pad // 16
    B = 40
    n_free = rng.integers(1, rows_pad + 1, size=B)
    nJ, nM = rng.integers(10, 300, size=B), rng.integers(30, 2000, size=B)
    ft, last, cend = np.zeros([B, nchm], int), np.zeros([B, nchm // 4], int), np.zeros([B, nchm], int)
    for b in range(B):
        nch = (int(n_free[b]) + 63) // 64 * 4
        reach = rng.integers(0, 20)
        first = np.maximum(0, np.arange(nch) - rng.integers(0, reach + 1, size=nch))
        f = np.minimum.accumulate(first[::-1])[::-1]                      # non-decreasing, ft[q] <= q
        lastc = [max(q for q in range(nch) if f[q] <= t) for t in range(nch)]
        ft[b, :nch] = f
        last[b, :nch // 4] = [lastc[4 * j + 3] for j in range(nch // 4)]
        cend[b, :nch] = [min(nch, max(lastc[t] + 1, (t | 3) + 1)) for t in range(nch)]
    # which stored tiles hold an entry of K_ff (csrc/trs_common.h kmask): random, diagonal tile always
    kmask = (rng.integers(0, 2 ** 32, size=[B, nchm], dtype=np.int64) | 1).astype(np.int64)
    plain = bench.envelope_counts_batch(n_free, nJ, nM, ft, last, cend, np.ones(B, bool))
    got = bench.envelope_counts_batch(n_free, nJ, nM, ft, last, cend, np.ones(B, bool), kmask=kmask)
    assert (got["stiffness_tiles"] <= got["tiles"]).all() and (got["stiffness_tiles"] < got["tiles"]).any()
    assert (plain["assemble_bytes"] >= got["assemble_bytes"]).all()
    for b in range(B):
        n = int(n_free[b])
        assert plain["potrf_bytes"][b] == 3 * plain["tiles"][b] * 2048 + 4 * 8 * ((n + 63) // 64 * 64)
        want = bench.algorithmic_counts(n, int(nJ[b]), int(nM[b]), cend[b], narrow=True, env_kmask=kmask[b])
        assert got["stiffness_tiles"][b] * 2048 == want["stiffness_tile_bytes"]
        assert got["potrf_tile_flops"][b] == bench.potrf_tile_flops(n, ft[b], last[b], cend[b], narrow=True), b
        assert got["assemble_bytes"][b] == want["assemble_bytes"]
        assert got["potrf_bytes"][b] == want["potrf_bytes"] + want["potrs_bytes"]   # fused substitution
        assert got["recover_bytes"][b] == want["recover_bytes"]
        assert got["tiles"][b] * 2048 == want["slab_tile_bytes"]


def test_available_cpus_respects_affinity_and_is_positive():
    n = generate.available_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    assert generate._load().trs_host_threads(0) >= 1


def test_cube_cpu_baseline_runs_on_a_fixed_sample_of_the_legs_batch():
    """`bench.cube_cpu_baseline` (north-star: the CPU baseline beside the cube-truss line): a fixed sample of the leg's own
    batch from the native host generator, solved by the oracle; the record and the checker's vectors."""
    import numpy as np
    idx = bench.cube_sample_indices(65536)
    assert len(idx) == 64 and idx[0] == 0 and idx[-1] == 65535 and (np.diff(idx) > 0).all()
    assert bench.cube_sample_indices(10).tolist() == list(range(10))
    rec, refs = bench.cube_cpu_baseline(48, seconds=0.2, pool_too=False)
    assert rec["kind"] == "port" and rec["cores"] == 1 and rec["unit"] == "solves/s" and rec["value"] > 0
    assert sorted(refs) == bench.cube_sample_indices(48).tolist()
    ratio = rec["reference_speed_ratio"]
    assert ratio is not None and 0.8 < ratio["ratio"] < 1.25      # the committed calibration against the real reference
    one = refs[0]
    assert one["u"].ndim == 2 and one["u"].shape[1] == 3 and np.isfinite(one["u"]).all() and np.abs(one["N"]).max() > 0


def test_late_leg_watchdog_prints_the_line_and_ends_the_process():
    """`bench.LateLegWatchdog`: a leg that never returns (a stalled device call does not come back to Python) must not
    take the finished line down - the watchdog prints it with a note and ends the process with a NON-ZERO status (a
    stalled device must not read as a clean run: VERDICT r4 item 7); a leg that returns in time leaves no trace, and
    its result is stored under the watchdog's lock (`put`)."""
    import json
    import subprocess
    import sys
    code = (
        "import json, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "line = {'metric': 'm', 'value': 1.0}\n"
        "g = bench.LateLegWatchdog(line, 'quick', 5.0); g.put(lambda: line.__setitem__('quick', 1))\n"
        "g = bench.LateLegWatchdog(line, 'stuck', 0.3)\n"
        "time.sleep(30)\n"
        "print(json.dumps({'unreachable': True}))\n")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert run.returncode == bench.LateLegWatchdog.EXIT_STATUS == 3
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 1                                   # printed once
    out = json.loads(lines[-1])
    assert out["value"] == 1.0 and out["quick"] == 1 and "stuck" in out["watchdog"] and "unreachable" not in out


def test_late_leg_watchdog_leaves_a_finished_leg_alone():
    """A leg that finishes right at the deadline: `put` and the watchdog's dump exclude each other, so the process
    lives on and the main thread prints the line itself."""
    import time
    line = {"value": 1.0}
    guard = bench.LateLegWatchdog(line, "edge", 0.2)
    guard.put(lambda: line.__setitem__("edge", {"ok": True}))
    time.sleep(0.5)                                          # (the watchdog thread has woken up and left by now)
    assert line == {"value": 1.0, "edge": {"ok": True}} and not guard._thread.is_alive()
