#!/usr/bin/env python3
"""Benchmark of the batched Truss.Solve() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): the 942-bar truss (tests/golden/data/bar-942_input_0.json,
n_free = 696) packed once and replicated to `--batch` (default 4096) INDEPENDENT problems per GPU,
inputs resident in HBM before the timed region.  One step = one pass of the whole pipeline over
the batch: joint order (found, applied and priced per truss, on the device) -> dofmap -> assemble -> potrf -> potrs ->
recover (six calls through the C ABI); nothing is shared between the copies or carried over between steps.
Weak scaling: every rank solves its own batch; no collective on the data path.

Order of a run: the legs of configs 3 and 5 (each with a barrier-bracketed region of its own), then the headline - W
warm-up steps in the form of the timed ones, the K timed steps between barriers (`value`), `--repeats` further repeats of
that region (`repeats`: min / median / max) -, then the informational legs (`leg_order` in the line).

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      dominant kernel (the factorisation with the fused substitution): algorithmic bytes or executed
                tile FLOP / measured average launch duration (events recorded on the launch stream inside the
                timed steps), the roof picked by arithmetic intensity; `frac_dense_8d` = SURVEY 8d's dense figure
  cpu_baseline  the numpy oracle (faithful restatement of the reference's per-truss path) timed on
                the host cores of this box, bounded sample, 1 thread (rank 0, N = 1 only)
  north_star_scaling   the three per-N lines north_star asks for in one record: config 2 weak-scaled (= the headline),
                config 3 strong-scaled (`--cube-total` trusses in all), config 5 strong-scaled and streamed
                (`--dataset-total` samples in all), each with the spread over the ranks and its global-index shards
  cube_batch    BASELINE config 3 at EVERY N: `--cube-batch` random cube trusses per GPU generated on the device,
                joint order INSIDE the timed step (batch.RaggedSolver), solves/s + roofline fractions
  dataset       BASELINE config 5 at every N: samples/s of data.dataset_chunks (generation, order, two solves and
                graph features on the device)
  extra.profile_order_hoisted (the headline of rounds 1-5: order found once before the timed region),
  large_truss   the route of trusses beyond the wave-per-matrix kernel's envelopes (8x8x8 grid, n ~ 2 000): stage
                times, executed TFLOP/s and algorithmic GB/s of the work-group factorisation
  pcie_inclusive, given_joint_order, dense_mode_potrf, ga_generation (BASELINE config 4),
  reference_protocol (the reference's own published benchmark: 30 x Truss.Solve() per case)   informational legs at N = 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# FP64 dense peaks of MI355X: 78.6 TFLOP/s matrix (= vector) per the public datasheet quoted in
# SURVEY.md section 8d; /opt/skills/guides/MI355X_MICROARCH.md lists no FP64 row.  HBM 8 TB/s (guide).
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
STAGES = ("dofmap", "assemble", "potrf", "potrs", "recover")


#: the sources that determine the factorisation kernel and what it reads (the PMC traffic record is about it)
FINGERPRINT_FILES = ("potrf.hip", "assemble.hip", "dofmap.hip", "trs_common.h", "trs_chol16.h", "trs_subst.h")


def kernel_source_sha():
    """Fingerprint of the device sources behind the profiled kernel (`FINGERPRINT_FILES` under csrc/): a PMC
    record under profiles/ is only quoted in the bench line while it was taken on exactly these sources (.git
    does not travel to the GPU box, so a commit id cannot be checked there)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "csrc")
    for name in FINGERPRINT_FILES:
        h.update(name.encode())
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_case(name):
    with open(os.path.join(ROOT, "tests", "golden", "data", name + ".json")) as fh:
        return json.load(fh)


def algorithmic_counts(n, nJ, nM, env_cend=None, narrow=False, env_kmask=None):
    """Per-truss algorithmic work (DESIGN.md section 'Kernels').  With an envelope only the tiles
    t .. cend[t]-1 of the rows of chunk t are written / read (csrc/trs_common.h); of those only the tiles that hold
    an entry of K_ff (`env_kmask`, wave-per-matrix kernels) are written by the assembly and read as stiffness
    tiles - the factor fills all of them.  The load vector of a matrix of the wave-per-matrix factorisation
    (`narrow`) travels in uf (8 bytes per row); the work-group factorisation carries it as a 16-wide column chunk
    of the slab."""
    npad = (n + 63) // 64 * 64
    inputs = 8 * nM + 16 * nM + 24 * nJ + nJ + 24 * nJ
    # upper part by 16-row tiles incl. diagonal tiles (+ the load-column chunk in the slab when not narrow)
    row_end = (lambda c: npad) if env_cend is None else (lambda c: 16 * int(env_cend[c // 16]))
    col = 0 if narrow else 16
    upper = sum((row_end(c) + col - (c // 16) * 16) for c in range(npad)) * 8
    k_upper = upper
    if env_kmask is not None and env_cend is not None and narrow:
        k_upper = 2048 * sum(bin((int(env_kmask[t]) & 0xffffffff) & ((1 << max(0, int(env_cend[t]) - t)) - 1)).count("1")
                             for t in range(npad // 16))
    vec = 8 * npad if narrow else 0
    return {
        "potrf_flops": n ** 3 / 3.0 + 2 * n ** 2,       # factor + two triangular solves
        "assemble_bytes": inputs + k_upper + vec,       # K written once (tiles with entries), f, + inputs
        "assemble_bytes_full_contract": inputs + 8 * n * n + 8 * n,  # SURVEY section 8d figure
        "potrf_bytes": k_upper + upper + 2 * vec,       # K tiles with entries read once, U written once; f in, y out
        "potrs_bytes": upper + 2 * vec + (0 if narrow else 8 * n),  # stored part of U read once, y in, u out
        "recover_bytes": 8 * nM + 16 * nM + 24 * nJ + 8 * n + 24 * nJ + 24 * nJ + 8 * nM,
        "slab_tile_bytes": upper, "stiffness_tile_bytes": k_upper,
    }


def npad_of(n):
    return (n + 63) // 64 * 64


def potrf_tile_flops(n, env_ft=None, env_last=None, env_cend=None, narrow=False):
    """FLOPs of the MFMA work the factorisation kernel executes for one matrix (2048 per 16x16x4
    MFMA): a host mirror of its panel loop with the same tile envelope (dense when env is None).
    Per panel: diagonal-block update, block factorisation (64 MFMAs between the tiles + 10 inside each
    of the four 16x16 factorisations), load column, item updates + triangular solves.  The
    wave-per-matrix kernel (`narrow`) issues no MFMA whose operand tile lies outside the envelope."""
    npad = (n + 63) // 64 * 64
    nch = npad // 16
    rs = 2 if narrow else 4
    mfma = 0
    for j in range(npad // 64):
        r0 = 64 * j
        kd = 16 * int(env_ft[4 * j]) if env_ft is not None else 0
        if narrow:
            bks = [16 * int(env_ft[4 * j + u]) for u in range(4)]
            for kk in range(kd, r0, 4):                      # block update, per k-step (the load vector
                mfma += sum(u + 1 for u in range(4) if kk >= bks[u])   # is advanced with VALU FMAs)
        else:
            mfma += (10 + 4) * (r0 - kd) // 4 + 40            # ten block tiles + load-column chunk
        mfma += 64 + 40                                        # F
        lastq = int(env_last[j]) if env_last is not None else nch - 1
        below = lastq - (4 * j + 3)
        for c0 in range(4 * j + 4, 4 * j + 4 + below, rs):
            nv = min(rs, 4 * j + 4 + below - c0)
            ks = 16 * int(env_ft[c0]) if env_ft is not None else 0
            mfma += nv * 4 * (r0 - ks) // 4                    # update
            for q in range(c0, c0 + nv):                       # solve against the block
                e = sum(1 for s in range(4) if q < int(env_cend[4 * j + s])) if narrow else 4
                mfma += 4 * e + 4 * (e * (e - 1) // 2)
    return 2048.0 * mfma


def cpu_baseline(data, seconds=15.0):
    """Time the numpy oracle (kind 'port') single-threaded on this box; bounded sample.  Returns the
    baseline record and the oracle's solution of the case (the checker of the GPU result)."""
    from oracle import truss_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    prepared = orc.prepare(data)   # the truss as the reference holds it once loaded (its protocol times Solve() only)
    ref = orc.solve(prepared)  # warm
    t0, runs = time.perf_counter(), 0
    while True:
        orc.solve(prepared)
        runs += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or runs >= 1000:
            break
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return {"value": runs / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{runs} sequential oracle.solve() calls on bar-942 ({dt:.1f} s), BLAS threads=1; "
                      f"host has {os.cpu_count()} logical CPUs",
            "reference_speed_ratio": oracle_speed_ratio("bar-942")}, ref


def oracle_speed_ratio(key):
    """oracle time / reference time on the same inputs (tests/golden/oracle_speed.json, written in the build
    container by tools/calibrate_oracle.py, which imports the real reference): how far the reported 'port' baseline
    is from the reference's own `Truss.Solve()`; > 1 = the port is slower.  None when the record is missing."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "oracle_speed.json")) as fh:
            rec = json.load(fh)
        return {"ratio": rec["_summary"][key], "note": rec["_summary"]["note"]}
    except Exception:
        return None


def _pool_worker(args):
    data, seconds = args
    from oracle import truss_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    data = orc.prepare(data)
    orc.solve(data)
    t0, runs = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        orc.solve(data)
        runs += 1
    return runs, time.perf_counter() - t0


def cpu_baseline_all_cores(data, seconds=8.0):
    """The same oracle on every host core at once (one forked single-threaded process per CPU this process may
    use - affinity mask and container CPU quota -, bounded sample).  Must run BEFORE this process touches the GPU: the workers are plain forks."""
    import multiprocessing as mp
    from python_stable_3d_truss_analysis_amd.generate import available_cpus
    n = available_cpus()   # affinity mask and cgroup CPU quota, not the logical CPUs of the machine
    with mp.get_context("fork").Pool(n) as pool:
        out = pool.map(_pool_worker, [(data, seconds)] * n)
    runs = sum(r for r, _ in out)
    dt = max(t for _, t in out)
    return {"value": runs / dt, "unit": "solves/s", "cores": n, "kind": "port",
            "sample": f"{runs} oracle.solve() calls on bar-942 by {n} single-threaded processes (of {os.cpu_count()} logical CPUs) in {dt:.1f} s"}


CUBE_SEED = 7
CUBE_CPU_SAMPLE = 64


def cube_sample_indices(B, count=CUBE_CPU_SAMPLE):
    """The fixed sample of the cube leg's batch that the CPU baseline solves (global truss indices of rank 0)."""
    import numpy as np
    return np.unique(np.linspace(0, max(0, B - 1), min(count, max(1, B))).astype(np.int64))


def cube_sample_json(B):
    """The sample's trusses as the reference's JSON dicts, from the native HOST generator (csrc/cubegen.c: bit for
    bit the batch the device generator produces for the same (seed, global index), tests/test_gpu_generate.py)."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    from python_stable_3d_truss_analysis_amd.data import dataset_sizes
    sizes = dataset_sizes(CUBE_SEED, 0, B, (8, 190))
    out = []
    for i in cube_sample_indices(B):
        pk = gen.generate_cube_batch(sizes[i:i + 1], gridRange=(6, 6, 6), seed=CUBE_SEED, first_index=int(i))
        out.append((int(i), gen.packed_to_json(pk, 0)))
    return out


def _cube_pool_worker(args):
    data, passes = args
    from oracle import truss_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    prepared = orc.prepare(data)
    for _ in range(passes):
        orc.solve(prepared)
    return passes


def cube_cpu_baseline(B, seconds=12.0, pool_too=True):
    """north_star: the reference CPU `Truss.Solve()` timed on the same box beside the cube-truss throughput
    (a 64-truss sample of config 3's distribution; reference workload generate.py:342-359).
    The oracle (kind 'port') solves a FIXED sample of the cube leg's own batch - the trusses with the global indices
    `cube_sample_indices(B)` of rank 0 - single-threaded, whole passes over the sample until `seconds` are spent;
    then the same sample on every CPU the process may use (forked single-threaded workers: only before the GPU is
    touched).  Returns (record, {index: oracle result}) - the results check the GPU leg's trusses afterwards."""
    from oracle import truss_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    sample = cube_sample_json(B)
    prepared = [orc.prepare(d) for _, d in sample]
    refs = {i: orc.solve(p) for (i, _), p in zip(sample, prepared)}   # warm + the checker's vectors
    t0, passes = time.perf_counter(), 0
    while True:
        for p in prepared:
            orc.solve(p)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= 50:
            break
    nM = [len(d["member"]) for _, d in sample]
    rec = {"value": passes * len(sample) / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": f"{passes} pass(es) of oracle.solve() over {len(sample)} trusses of the leg's own batch (global "
                     f"indices linspace(0, {B - 1}, {len(sample)}) of rank 0, {min(nM)}..{max(nM)} members, "
                     f"{dt:.1f} s), BLAS threads=1",
           "reference_speed_ratio": oracle_speed_ratio("cube_mean")}
    if pool_too:
        try:
            import multiprocessing as mp
            from python_stable_3d_truss_analysis_amd.generate import available_cpus
            n = available_cpus()
            per = passes
            with mp.get_context("fork").Pool(n) as pool:
                t1 = time.perf_counter()
                done = sum(pool.imap_unordered(_cube_pool_worker, [(d, per) for _, d in sample], chunksize=1))
                dtp = time.perf_counter() - t1
            rec["all_cores"] = {"value": done / dtp, "unit": "solves/s", "cores": n, "kind": "port",
                                "sample": f"the same {len(sample)} trusses x {per} solve(s) each over {n} forked "
                                          f"single-threaded workers ({dtp:.1f} s)"}
        except Exception as exc:  # informational
            rec["all_cores"] = {"error": repr(exc)}
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
    return rec, refs


def envelope_counts_batch(n_free, nJ, nM, ft, last, cend, narrow, kmask=None):
    """Per-truss executed MFMA FLOP of the factorisation and algorithmic bytes of every stage for a bucket of
    NARROW-envelope matrices, from the envelope metadata trs_assemble left (arrays [Bb, ...]): the vectorised
    form of `potrf_tile_flops(narrow=True)` and `algorithmic_counts(narrow=True)` (asserted equal in
    tests/test_bench_helpers.py).  Returns dict of float arrays [Bb]."""
    import numpy as np
    n_free = np.asarray(n_free, dtype=np.int64)
    Bb = len(n_free)
    npad = (n_free + 63) // 64 * 64
    nch, npan = npad // 16, npad // 64
    ft, last, cend = (np.asarray(a, dtype=np.int64) for a in (ft, last, cend))
    width = ft.shape[1]
    mfma = np.zeros(Bb, dtype=np.int64)
    rows = np.arange(Bb)
    for j in range(int(npan.max(initial=0))):
        act = j < npan
        r0 = 64 * j
        cols = np.minimum(4 * j + np.arange(4), width - 1)
        ftj = ft[:, cols]
        kd = 16 * ftj[:, 0]
        for u in range(4):
            mfma += act * (u + 1) * (np.maximum(0, r0 - np.maximum(kd, 16 * ftj[:, u])) // 4)
        mfma += act * 104
        lastq = last[:, min(j, last.shape[1] - 1)]
        below = np.where(act, lastq - (4 * j + 3), 0)
        end = 4 * j + 4 + below
        cj = cend[:, cols]                                             # stored extents of the panel's four chunks
        for i in range(int((below.max(initial=0) + 1) // 2)):
            c0 = 4 * j + 4 + 2 * i
            valid = c0 < end
            nv = np.minimum(2, end - c0)
            ks = 16 * ft[rows, np.minimum(c0, width - 1)]
            mfma += valid * nv * (r0 - ks)
            for q in (c0, c0 + 1):
                has = valid & (q < c0 + nv)
                e = (q < cj).sum(axis=1)
                mfma += has * (4 * e + 2 * e * (e - 1))
    t = np.arange(width)[None, :]
    tiles = ((cend - t) * (t < nch[:, None])).sum(axis=1)
    upper = tiles * 2048
    k_upper = upper
    if kmask is not None:   # stiffness tiles with an entry of K_ff: the ones the assembly writes and the factorisation reads
        km = np.asarray(kmask, dtype=np.int64) & 0xffffffff
        live = km & ((np.int64(1) << np.clip(cend - t, 0, 32)) - 1)
        bits = np.zeros_like(live)
        for b in range(32):
            bits += (live >> b) & 1
        k_upper = (bits * (t < nch[:, None])).sum(axis=1) * 2048
    vec = 8 * npad
    inputs = 24 * np.asarray(nM, dtype=np.int64) + 49 * np.asarray(nJ, dtype=np.int64)
    return {"potrf_tile_flops": 2048.0 * mfma, "tiles": tiles, "stiffness_tiles": k_upper // 2048,
            "assemble_bytes": inputs + k_upper + vec,
            "potrf_bytes": k_upper + 2 * upper + 4 * vec,  # factor + substitution by the same wave
            "recover_bytes": 24 * np.asarray(nM) + 24 * np.asarray(nJ) + 8 * n_free + 48 * np.asarray(nJ) + 8 * np.asarray(nM),
            "order_bytes": 2 * (49 * np.asarray(nJ, dtype=np.int64) + 8 * np.asarray(nM, dtype=np.int64)) + 4 * np.asarray(nJ)}


def cube_workload(B, rank, seed=CUBE_SEED, device=None, first=None):
    """BASELINE config 3: B random cube trusses as `GenerateRandomCubeTrusses(gridRange=(6,6,6), numCube ~ U{8..190},
    LinkType.Random, GenerateMethod.Random)`, generated ON THE DEVICE (csrc/cubegen.hip: bit for bit the native
    host generator csrc/cubegen.c, whose distribution is pinned against the reference in tests/test_generate.py).
    The dataset is keyed by the GLOBAL truss index: rank r draws the trusses r B .. (r + 1) B - 1, or - `first` -
    the trusses first .. first + B - 1.  Returns (sizes, device tensors)."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    from python_stable_3d_truss_analysis_amd.data import dataset_sizes
    first = rank * B if first is None else int(first)
    sizes = dataset_sizes(seed, first, B, (8, 190))
    return gen.generate_cube_batch_device(sizes, gridRange=(6, 6, 6), seed=seed, first_index=first, device=device)


def strong_shard(total, rank, world):
    """Contiguous shard of a dataset of `total` globally indexed units: (first, count) of `rank`.  The units are i.i.d.
    draws, so contiguous shards are balanced in cost as well as in count (SURVEY 8e asks for a size-balanced deal
    where the host knows the sizes: `shard.py` does that for packed batches)."""
    share = (int(total) + world - 1) // world
    first = min(int(total), rank * share)
    return first, max(0, min(share, int(total) - first))


def config3_strong_leg(args, device, torch, batch, barrier, reduce_max, gather, rank, world):
    """north_star's scaling sentence for config 3, STRONG-scaled: `--cube-total` (65 536) cube trusses in all, rank r
    solves the global indices of `strong_shard`; K steps between barriers, max over the ranks."""
    import numpy as np
    first, count = strong_shard(args.cube_total, rank, world)
    packed, tensors = cube_workload(count, rank, device=device, first=first)
    solver = batch.RaggedSolver(packed, reorder=True, tensors=tensors)
    for _ in range(max(1, args.cube_warmup)):
        solver.step()
    torch.cuda.synchronize(device)
    solver.adopt_launch_hints()
    solver.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.cube_steps):
        solver.step()
    barrier()
    local = time.perf_counter() - t0
    lo, hi = reduce_max(local, "min"), reduce_max(local)
    bad = int((solver.info != 0).sum().item())
    shards = gather({"rank": rank, "first": first, "count": count, "members": int(packed.nM.astype(np.int64).sum()),
                     "joints": int(packed.nJ.astype(np.int64).sum()), "info_nonzero": bad})
    del solver
    batch.release_workspaces()
    if rank != 0:
        return None
    return {"config": "BASELINE config 3: mixed cube trusses (grid 6x6x6, 8..190 cubes), one ragged batch",
            "scaling": "strong", "total_trusses": int(args.cube_total), "steps": args.cube_steps,
            "value": args.cube_total * args.cube_steps / hi, "unit": "solves/s",
            "ms_per_step": hi / args.cube_steps * 1e3,
            "rank_ms_per_step": {"min": lo / args.cube_steps * 1e3, "max": hi / args.cube_steps * 1e3},
            "shards": shards, "info_nonzero": sum(sh["info_nonzero"] for sh in shards),
            "note": "rank r holds the trusses with the global indices [first, first + count) resident and solves them "
                    "every step (joint order inside the step); no exchange between the ranks"}


def config5_strong_leg(args, device, torch, barrier, reduce_max, gather, rank, world):
    """north_star's scaling sentence for config 5, STRONG-scaled and END TO END: a dataset of `--dataset-total` (1e6)
    cube trusses, two solves per sample, every sample delivered in host memory as the packed HeteroData tensors
    (`data.dataset_stream`); rank r owns the chunks r, r + world, ... of the globally indexed dataset."""
    from python_stable_3d_truss_analysis_amd import MemberType, TaskType
    from python_stable_3d_truss_analysis_amd import data as gdata
    total = int(args.dataset_total)
    per_rank = (total + world - 1) // world
    chunks_per_rank = max(1, (per_rank + 32767) // 32768)
    chunk = max(1, (per_rank + chunks_per_rank - 1) // chunks_per_rank)   # equal chunks of at most 32 768, the same number per rank
    kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
              taskType=TaskType.REGRESSION, device=device, forceScale=1e3, displaceScale=0.1, positionScale=100.)
    for _ in gdata.dataset_stream(min(total, 2 * chunk), rank=0, world=1, chunk=chunk, **kw):   # warm: rings, allocator
        pass
    barrier()
    t0 = time.perf_counter()
    got = nbytes = bad = joints = 0
    firsts = []
    for graphs in gdata.dataset_stream(total, rank=rank, world=world, chunk=chunk, **kw):
        got += len(graphs)
        nbytes += graphs.nbytes
        bad += int(graphs.tensors["info"].ne(0).sum().item())
        joints += int(graphs.nJ.sum())
        firsts.append([int(graphs.first), len(graphs)])
    barrier()
    local = time.perf_counter() - t0
    lo, hi = reduce_max(local, "min"), reduce_max(local)
    shards = gather({"rank": rank, "chunks": firsts, "samples": got, "joints": joints, "bytes": nbytes, "info_nonzero": bad})
    gdata.release_stream_buffers()
    if rank != 0:
        return None
    return {"config": "BASELINE config 5: cube-truss dataset generation, two solves per sample, streamed to the host "
                      "as packed HeteroData tensors",
            "scaling": "strong", "total_samples": total, "chunk": chunk, "solves_per_sample": 2,
            "value": total / hi, "unit": "samples/s", "seconds": hi,
            "rank_seconds": {"min": lo, "max": hi},
            "bytes_per_sample": sum(sh["bytes"] for sh in shards) / max(1, sum(sh["samples"] for sh in shards)),
            "shards": shards, "info_nonzero": sum(sh["info_nonzero"] for sh in shards),
            "note": "data.dataset_stream on every rank: generation, joint order, two solves, packed feature kernel on "
                    "the device; DMA into page-locked host memory overlapped with the next chunk; no exchange "
                    "between the ranks"}


def cube_batch_leg(args, device, torch, batch, barrier, reduce_max, rank, world, cpu=None):
    """The ragged workload north_star names (BASELINE config 3): `--cube-batch` random cube trusses per GPU,
    resident in HBM in the GENERATOR's joint numbering; one step = joint order on the device + bucketed
    assembly / factorisation / substitution / recovery + results back in the caller's order and numbering
    (`batch.RaggedSolver`).  Timed like the headline: barrier, K steps, barrier, max over ranks."""
    import numpy as np
    cube_workload(256, rank, device=device)   # warm
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    packed, tensors = cube_workload(args.cube_batch, rank, device=device)
    torch.cuda.synchronize(device)
    t_gen = time.perf_counter() - t0
    solver = batch.RaggedSolver(packed, reorder=True, tensors=tensors)
    for _ in range(max(1, args.cube_warmup)):
        solver.step()
    torch.cuda.synchronize(device)
    hinted = solver.adopt_launch_hints()
    solver.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.cube_steps):
        solver.step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed_min = reduce_max(elapsed_local, "min")
    elapsed = reduce_max(elapsed_local)
    records = []
    for _ in range(args.cube_steps):
        solver.step(record=records)
    torch.cuda.synchronize(device)
    stage_ms = {}
    for name, e0, e1 in records:
        stage_ms[name] = stage_ms.get(name, 0.0) + e0.elapsed_time(e1) / args.cube_steps
    info_bad = int((solver.info != 0).sum().item())
    if rank != 0:
        return None
    # the oracle's sample of this batch (solved on the host before the GPU was touched): the leg's own check
    check = None
    if cpu is not None:
        refs = cpu[1]
        idx = torch.tensor(sorted(refs), device=device)
        got_u, got_N = solver.u[idx].cpu().numpy(), solver.N[idx].cpu().numpy()
        eu = en = 0.0
        for k, i in enumerate(sorted(refs)):
            ru, rn = refs[i]["u"], refs[i]["N"]
            eu = max(eu, float(np.abs(got_u[k, :len(ru)] - ru).max() / np.abs(ru).max()))
            en = max(en, float(np.abs(got_N[k, :len(rn)] - rn).max() / np.abs(rn).max()))
        check = {"u": eu, "N": en, "trusses_checked": len(refs)}
    # executed matrix-core work and algorithmic bytes of this rank's batch, from the envelope metadata of every
    # bucket (the buckets share one workspace: re-assemble bucket by bucket to read it)
    flops = bytes_alg = tiles = ktiles = 0.0
    n_wide = 0
    per_stage = {"order": 0.0, "assemble": 0.0, "potrf": 0.0, "recover": 0.0}
    for bk in solver.buckets:
        db, idx = bk["dev"], bk["idx"]
        nJ, nM, nf = packed.nJ[idx], packed.nM[idx], packed.n_free[idx]
        if bk["renumbered"]:
            per_stage["order"] += float((2 * (49 * nJ.astype(np.int64) + 8 * nM) + 4 * nJ).sum())
        if db.small:   # the fused small-system kernel: inputs once, results once
            per_stage["assemble"] += float((24 * nM.astype(np.int64) + 49 * nJ).sum())
            per_stage["recover"] += float((8 * nM.astype(np.int64) + 48 * nJ).sum())
            continue
        # (the buckets share one workspace, which holds the LAST bucket's metadata: redo this bucket's first two
        # stages on its own gathered inputs, which persist)
        db.dofmap(); db.assemble()
        torch.cuda.synchronize(device)
        env = db.env.cpu().numpy()
        nchm, npan = db.rows // 16, db.rows // 64
        narrow = (env[:, nchm + npan] & 0xff) == 1
        n_wide += int((~narrow).sum())
        c = envelope_counts_batch(nf, nJ, nM, env[:, :nchm], env[:, nchm:nchm + npan],
                                  env[:, nchm + npan + 8: nchm + npan + 8 + nchm], narrow,
                                  kmask=env[:, 2 * nchm + npan + 8: 3 * nchm + npan + 8])
        ktiles += float(c["stiffness_tiles"].sum())
        flops += float(c["potrf_tile_flops"][narrow].sum())
        tiles += float(c["tiles"].sum())
        for k in ("assemble", "potrf", "recover"):
            per_stage[k] += float(c[k + "_bytes"].sum())
    bytes_alg = sum(per_stage.values())
    step_s = elapsed / args.cube_steps
    tf, gbs = flops / step_s / 1e12, bytes_alg / step_s / 1e9
    intensity = flops / max(1.0, bytes_alg)
    bound = "mfma" if intensity >= PEAK_FP64_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    return {"workload": f"{packed.B} random cube trusses per GPU (grid 6x6x6, 8..190 cubes, seed 7): "
                        f"{int(packed.nM.min())}..{int(packed.nM.max())} members, n_free {int(packed.n_free.min())}.."
                        f"{int(packed.n_free.max())} (mean {float(packed.n_free.mean()):.0f})",
            "value": world * packed.B * args.cube_steps / elapsed, "unit": "solves/s", "ms_per_step": step_s * 1e3,
            "rank_ms_per_step": {"min": elapsed_min / args.cube_steps * 1e3, "max": step_s * 1e3},
            "steps": args.cube_steps, "batch_per_gpu": packed.B, "buckets": len(solver.buckets),
            "buckets_with_launch_hints": hinted, "info_nonzero": info_bad, "wide_envelopes": n_wide,
            "joint_order": "trs_joint_order on the device, INSIDE the timed step (effort 3: every coordinate sweep; reverse "
                           "Cuthill-McKee and its reverse for trusses below 128 free joints); results in the generator's "
                           "numbering",
            "stages_ms": stage_ms, "lanes": solver.lanes,
            "stages_ms_note": "event pairs per bucket, summed; the buckets run on `lanes` streams at the same time, so the "
                              "sums exceed the step",
            "roofline": {"bound": bound, "intensity_flop_per_byte": intensity,
                         "mfma": {"achieved": tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP64_TFLOPS,
                                  "flop_model": "MFMA work of the factorisation inside the 16x16-tile envelopes, "
                                                "per step and GPU, over the WHOLE step time"},
                         "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                 "byte_model": "order (inputs read, renumbered inputs + permutation written) + assembly "
                                               "(inputs, the stored tiles that hold an entry of K_ff, load vector) + "
                                               "factorisation with fused substitution (those tiles read, every stored tile "
                                               "written as factor and read back, vectors) + recovery, over the WHOLE step time; "
                                               "the bucket gather / scatter copies are overhead, not counted",
                                 "bytes_per_step": bytes_alg, "by_stage": per_stage},
                         "stored_tiles_per_truss": tiles / max(1, packed.B),
                         "stiffness_tiles_per_truss": ktiles / max(1, packed.B)},
            "device_generate_s": t_gen,
            "cpu_baseline": cpu[0] if cpu is not None else None,
            "max_rel_err_vs_oracle": check,
            "note": "generated on the device, resident in generator order; one launch pipeline per size bucket, the "
                    "buckets dealt onto `lanes` streams (fork from / join into the caller's stream inside the step), a "
                    "workspace per lane"}


def host_fed_leg(args, device, torch, batch):
    """Informational, rank 0 at N = 1, one of the legs that drive SEVERAL streams (they run last, under the watchdog): the
    cube batch from page-locked HOST arrays to page-locked host results, one call (set-up, bucket pulls, device order +
    solves, pushes: `batch.solve_batch_streamed`)."""
    import numpy as np
    packed, tensors = cube_workload(args.cube_batch, 0, device=device)
    # the host batch in the TABLE member form (ABI 10: uint16 end joints + one type index per member, 5 instead of 24
    # bytes per member over PCIe; the generator's trusses use one member type) - what `pack_json(..., members="auto")`
    # gives for such a batch; results bit for bit those of the general form (tests/test_gpu_member_forms.py)
    host = packed.to_packed(tensors)
    pinned, pool = host.table().pinned(), batch.ResultPool(tracked=True)   # (results are only read)
    for _ in range(2):
        first = batch.solve_batch_streamed(pinned, device, reorder=True, pool=pool)
    first_u, first_N = np.array(first.displace), np.array(first.internal)   # (the pool's arrays are re-used by the next call)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        got = batch.solve_batch_streamed(pinned, device, reorder=True, pool=pool)
    dt = (time.perf_counter() - t0) / reps
    # copy kernels and solver kernels share the device in this pipeline: the calls must still repeat each other bit for bit
    repeat = bool(np.array_equal(first_u, got.displace, equal_nan=True) and np.array_equal(first_N, got.internal, equal_nan=True))
    # the same pipeline as a RESIDENT host-fed solver (set up once, stepped again: what a caller who keeps the solver pays)
    resident = None
    try:
        fields = batch.RaggedSolver.GATHER_TABLE if pinned.is_table else batch.RaggedSolver.GATHER
        host_in = {f: torch.from_numpy(getattr(pinned, f)) for f in fields}
        if pinned.is_table:
            host_in["types"] = torch.from_numpy(np.ascontiguousarray(pinned.types, dtype=np.float64))
        host_out = batch.host_result_arrays(torch, pool, packed.B, pinned.nJ_max, pinned.nM_max, device)
        solver = batch.RaggedSolver(pinned, device, reorder=True, max_slab_bytes=48 << 30, host_io=(host_in, host_out))
        solver.step(); torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.step()
        torch.cuda.synchronize(device)
        dts = (time.perf_counter() - t0) / reps
        same = bool(np.array_equal(first_u, solver.result().displace, equal_nan=True))
        resident = {"solves_per_s": packed.B / dts, "ms_per_step": dts * 1e3, "results_equal_the_calls": same}
        del solver
    except Exception as exc:   # informational
        resident = {"error": repr(exc)}
    nJ64, nM64 = packed.nJ.astype(np.int64), packed.nM.astype(np.int64)
    return {"solves_per_s": packed.B / dt, "ms_per_call": dt * 1e3, "resident_solver": resident,
            "h2d_live_bytes": int((nJ64 * 49 + nM64 * 5).sum()),
            "h2d_live_bytes_general_form": int((nJ64 * 49 + nM64 * 24).sum()), "member_form": "table",
            "d2h_live_bytes": int((nJ64 * 48 + nM64 * 8).sum()) + 4 * packed.B,
            "info_nonzero": int((got.info != 0).sum()), "calls_repeat_bitwise": repeat,
            "note": "host arrays in -> host results out per CALL of batch.solve_batch_streamed (set-up "
                    "included): buckets pulled over PCIe by the gather kernel (live bytes only), ordered "
                    "and solved on the device, results pushed into page-locked arrays; three streams, "
                    "the copy kernels on compute units of their own (CU masks, TRS_PCIE_CUS=16,8: the default "
                    "again since the joint-order race of EXPERIMENTS R5.1 is fixed); never the leg's value"}


def dataset_leg(args, device, torch, barrier, reduce_max, rank, world):
    """BASELINE config 5: this rank's share of a cube-truss dataset streamed through `data.dataset_chunks` -
    generation, joint order, TWO solves per sample (actual sections and the fixed prior, reference
    data.py:107-114) and the HeteroData feature kernel, all on the device; float32 tensors left resident per
    chunk.  `--dataset-samples` per GPU; rate = samples of all ranks / max rank time."""
    from python_stable_3d_truss_analysis_amd import MemberType, TaskType
    from python_stable_3d_truss_analysis_amd import data as gdata
    chunk = min(32768, args.dataset_samples)   # (chunks of 16 384 leave the buckets with fewer matrices than the chip holds waves: 453 against 523 K samples/s)
    total = args.dataset_samples * world
    kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
              taskType=TaskType.REGRESSION, device=device, forceScale=1e3, displaceScale=0.1, positionScale=100.)
    # warm: kernels and launch tables with a small chunk, then one chunk of the timed size, so that the timed chunks find
    # their workspaces and tensor shapes in the allocator's cache (the first 16 GB hipMalloc costs a fifth of a second)
    for _ in gdata.dataset_chunks(min(chunk, 2048), rank=0, world=1, chunk=min(chunk, 2048), **kw):
        pass
    for _ in gdata.dataset_chunks(chunk, rank=0, world=1, chunk=chunk, **kw):
        pass
    barrier()
    t0 = time.perf_counter()
    seen = bad = 0
    for first, packed, tensors in gdata.dataset_chunks(total, rank=rank, world=world, chunk=chunk, **kw):
        seen += packed.B
        bad += int(tensors["info"].ne(0).sum().item())
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    if rank != 0:
        return None
    return {"value": total / elapsed, "unit": "samples/s", "samples_per_gpu": args.dataset_samples, "chunk": chunk,
            "seconds": elapsed, "solves_per_sample": 2, "info_nonzero_rank0": bad, "rank0_samples": seen,
            "note": "`value` = the compute side of config 5 (data.dataset_chunks: generation, joint order, two solves "
                    "and graph features per sample, all on the device, padded tensors left resident per chunk); "
                    "`streamed` (N = 1) = config 5 end to end, the samples delivered in host memory; mixed cube "
                    "trusses of 8..190 cubes"}


def dataset_streamed_leg(args, device, resident_rate):
    """Informational, rank 0 at N = 1, a leg that drives two streams (it runs last, under the watchdog): the dataset of
    `dataset_leg` DELIVERED - packed feature rows + edge indices in page-locked host memory (data.dataset_stream, the sink
    of config 5: "results streamed to PyG HeteroData"), the copies overlapped with the next chunk."""
    from python_stable_3d_truss_analysis_amd import MemberType, TaskType
    from python_stable_3d_truss_analysis_amd import data as gdata
    chunk = min(32768, args.dataset_samples)
    total = args.dataset_samples
    kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
              taskType=TaskType.REGRESSION, device=device, forceScale=1e3, displaceScale=0.1, positionScale=100.)
    for _ in gdata.dataset_stream(2 * chunk, rank=0, world=1, chunk=chunk, **kw):   # warm: page-locks both slots of the ring
        pass
    t0 = time.perf_counter()
    got = nbytes = sbad = 0
    for graphs in gdata.dataset_stream(total, rank=0, world=1, chunk=chunk, **kw):
        got += len(graphs)
        nbytes += graphs.nbytes
        sbad += int(graphs.tensors["info"].ne(0).sum().item())
        g = graphs[len(graphs) // 2]     # a consumer's view: one sample of the chunk as a graph object
        assert g["joint"].x.shape[0] == int(graphs.nJ[len(graphs) // 2])
    s_elapsed = time.perf_counter() - t0
    return {"value": total / s_elapsed, "unit": "samples/s", "seconds": s_elapsed,
            "bytes_per_sample": nbytes / max(1, got), "d2h_GBps_rank0": nbytes / s_elapsed / 1e9,
            "info_nonzero_rank0": sbad, "rank0_samples": got, "vs_resident": (total / s_elapsed) / resident_rate,
            "note": "data.dataset_stream: as `value`, then the un-padded float32 feature rows, targets and the "
                    "members' end joints (row 0 of every sample's j2m edge index; int32) of every chunk copied "
                    "by DMA into page-locked host memory on a second stream while the device works on the "
                    "next chunk; per chunk one sample materialised as a graph object (HeteroData where "
                    "torch_geometric is installed, else the dict-of-stores stand-in)"}


def large_truss_leg(args, device, torch, batch):
    """Informational, rank 0 at N = 1: the route LARGE trusses take (the reference has no size cap, truss.py:307-316) -
    `--large-trusses` (512) cube trusses of 300..400 cubes on an 8 x 8 x 8 grid (n_free ~ 1 500..2 000, ~4 000
    members): joint order on the device, then the staged pipeline - their envelopes reach further below the diagonal
    blocks than the wave-per-matrix kernel takes, so the factorisation is `trs_potrf_kernel` (a work-group per matrix,
    LDS hand-offs) and the substitution `trs_potrs_kernel`.  Stage times from events, algorithmic bytes and executed
    MFMA FLOP from the envelope metadata of every truss."""
    import numpy as np
    from python_stable_3d_truss_analysis_amd import generate as gen
    B = int(args.large_trusses)
    cubes = np.random.default_rng(21).integers(300, 401, size=B)
    sizes, tensors = gen.generate_cube_batch_device(cubes, gridRange=(8, 8, 8), seed=21, device=device)
    ordered = batch.joint_order_device(torch, tensors, effort=3)
    tens = dict(tensors)
    tens.update({k: ordered[k] for k in ("xyz", "conn", "cbits", "loads")})
    dev = batch.DeviceBatch.from_device(tens, sizes.n_max, joint_out=ordered["perm"])
    find_order = lambda: batch.joint_order_device(torch, tensors, effort=3, out=ordered)
    calls = (("order", find_order), ("dofmap", dev.dofmap), ("assemble", dev.assemble), ("potrf", dev.potrf),
             ("potrs", dev.potrs), ("recover", dev.recover))
    reps = 5

    def measure():
        for _ in range(2):
            for _, call in calls:
                call()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            for _, call in calls:
                call()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in calls] for _ in range(reps)]
        for r in range(reps):
            for (name, call), (e0, e1) in zip(calls, ev[r]):
                e0.record(); call(); e1.record()
        torch.cuda.synchronize(device)
        return dt, {name: float(np.mean([ev[r][i][0].elapsed_time(ev[r][i][1]) for r in range(reps)]))
                    for i, (name, _) in enumerate(calls)}

    # the other route first: every matrix on the work-group kernels (four waves per matrix: TRS_ASM_ALL_WIDE)
    dev.options["all_wide"] = True
    dt_wide, stage_wide = measure()
    u_wide = dev.u.clone()
    dev.options["all_wide"] = False
    dt, stage_ms = measure()
    scale = dev.u.abs().amax(dim=(1, 2), keepdim=True)
    routes_agree = float(((u_wide - dev.u).abs() / scale).max().item())
    env = dev.env.cpu().numpy()
    nchm, npan = dev.rows // 16, dev.rows // 64
    narrow = (env[:, nchm + npan] & 0xff) == 1
    flops = asm_b = potrf_b = potrs_b = rec_b = dense_flops = 0.0
    for b in range(B):
        n, nJ, nM = int(sizes.n_free[b]), int(sizes.nJ[b]), int(sizes.nM[b])
        ft, last = env[b, :nchm], env[b, nchm:nchm + npan]
        cend = env[b, nchm + npan + 8: nchm + npan + 8 + nchm]
        kmask = env[b, 2 * nchm + npan + 8: 3 * nchm + npan + 8]
        c = algorithmic_counts(n, nJ, nM, cend, bool(narrow[b]), kmask)
        flops += potrf_tile_flops(n, ft, last, cend, bool(narrow[b]))
        dense_flops += c["potrf_flops"]
        asm_b += c["assemble_bytes"]; potrf_b += c["potrf_bytes"]; potrs_b += c["potrs_bytes"]; rec_b += c["recover_bytes"]
    res_info = int((dev.info != 0).sum().item())
    gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9
    potrf_tf = flops / (stage_ms["potrf"] * 1e-3) / 1e12
    return {"workload": f"{B} cube trusses of 300..400 cubes on an 8x8x8 grid: {int(sizes.nM.min())}..{int(sizes.nM.max())} "
                        f"members, n_free {int(sizes.n_free.min())}..{int(sizes.n_free.max())}; joint order inside the step",
            "solves_per_s": B / dt, "ms_per_step": dt * 1e3, "stages_ms": stage_ms, "info_nonzero": res_info,
            "wide_envelopes": int((~narrow).sum()), "slab_rows": int(dev.rows),
            "potrf": {"kernel": "trs_potrf_kernel" if not narrow.all() else "trs_potrf_narrow_kernel<false, 2>",
                      "mfma_tflops": potrf_tf, "frac_of_fp64_peak": potrf_tf / PEAK_FP64_TFLOPS,
                      "dense_equivalent_tflops": dense_flops / (stage_ms["potrf"] * 1e-3) / 1e12,
                      "algorithmic_GBps": gbs(potrf_b, stage_ms["potrf"]), "frac_of_hbm_peak": gbs(potrf_b, stage_ms["potrf"]) / PEAK_HBM_GBS,
                      "intensity_flop_per_byte": flops / max(1.0, potrf_b)},
            "assemble": {"algorithmic_GBps": gbs(asm_b, stage_ms["assemble"]), "frac_of_hbm_peak": gbs(asm_b, stage_ms["assemble"]) / PEAK_HBM_GBS},
            "potrs": {"algorithmic_GBps": gbs(potrs_b, stage_ms["potrs"]), "frac_of_hbm_peak": gbs(potrs_b, stage_ms["potrs"]) / PEAK_HBM_GBS},
            "recover": {"algorithmic_GBps": gbs(rec_b, stage_ms["recover"]), "frac_of_hbm_peak": gbs(rec_b, stage_ms["recover"]) / PEAK_HBM_GBS},
            "work_group_route": {"solves_per_s": B / dt_wide, "ms_per_step": dt_wide * 1e3, "stages_ms": stage_wide,
                                 "max_rel_diff_u_vs_default_route": routes_agree,
                                 "note": "options={'all_wide': True}: every matrix on trs_potrf_kernel / trs_potrs_kernel "
                                         "(a work-group of four waves per matrix) whatever its envelope"},
            "note": "informational: FLOP = the MFMA work inside the 16x16-tile envelopes (what the kernel executes), bytes = "
                    "stored tiles once per stage + vectors + inputs (`algorithmic_counts`); profile: profiles/r06_large_*"}


def ga_leg(device, torch):
    """BASELINE config 4 (informational, rank 0 at N = 1): one GA generation on bar-120 with nPop = 1024 - every
    fitness evaluation of the generation in ONE batched solve (the fused small-system kernel with the constraint
    reductions, reference ga.py:139-160), wall time per generation evaluation."""
    import random
    from python_stable_3d_truss_analysis_amd import MemberType, Truss
    from python_stable_3d_truss_analysis_amd.ga import GA
    truss = Truss(3).LoadFromJSON(data=load_case("bar-120_input_0"))
    state = random.getstate()
    random.seed(0)
    types = [MemberType(i, random.uniform(1e7, 3e7), random.uniform(0.1, 1.0)) for i in range(1, 21)]   # example.py:186
    ga = GA(truss, types, nIteration=3, nPop=1024, nElite=256)
    pop = ga.Initialize()
    random.setstate(state)
    ga.GetFitnessBatch(pop)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fit = ga.GetFitnessBatch(pop)
    dt = (time.perf_counter() - t0) / reps
    # the WHOLE loop of config 4 (ga.GA.Evolve, reference ga.py:192-237): selection, offspring and the batched fitness
    # of every generation, under one seed, by the native generation loop and by the reference-shaped Python loop
    evolve = {}
    gens = 30
    for name, native in (("native_loop", True), ("python_loop", False)):
        try:
            state = random.getstate()
            random.seed(1)
            run = GA(truss, types, nIteration=gens, nPatience=10 ** 6, nPop=1024, nElite=256)
            run._adopt_population(ga)    # (the resident population batch, built above)
            t0 = time.perf_counter()
            _, info, _, history = run.Evolve(isPrintMessage=False, native=native)
            wall = time.perf_counter() - t0
            random.setstate(state)
            evolve[name] = {"ms_per_generation": wall / gens * 1e3, "generations": gens,
                            "best_fitness": history[-1], "x_fitness_call": wall / gens / dt}
        except Exception as exc:
            evolve[name] = {"error": repr(exc)}
    if all("best_fitness" in v for v in evolve.values()):
        evolve["same_trajectory"] = evolve["native_loop"]["best_fitness"] == evolve["python_loop"]["best_fitness"]
    return {"nPop": 1024, "truss": "bar-120 (120 members, 111 free DOFs), 20 member types", "ms_per_generation_eval": dt * 1e3,
            "fitness_evals_per_s": 1024 / dt, "feasible_in_population": int(sum(1 for f in fit if f[1] and f[2])),
            "evolve": evolve,
            "note": "wall time of GA.GetFitnessBatch (gene matrix -> sections on the device -> trs_solve_small with "
                    "the fitness reductions -> one download); informational"}


def reference_protocol_leg():
    """The reference's OWN published benchmark (README.md:86-94, example.py:1-25): load a
    truss once, call `Truss.Solve()` 30 times, mean wall time per call - through this package's drop-in `Truss`
    (every call: pack, upload, kernels, download, sparse result dicts), for the seven bundled cases.  Informational,
    rank 0 at N = 1: one truss per call is not what the device is for, but it is the number the reference quotes."""
    from python_stable_3d_truss_analysis_amd.truss import Truss
    published = {"bar-6": 0.00037, "bar-10": 0.00050, "bar-25": 0.00126, "bar-47": 0.00253, "bar-72": 0.00323,
                 "bar-120": 0.00557, "bar-942": 0.05253}     # seconds, Intel i7-10750H (README.md:88-94)
    out = {}
    for case, ref_s in published.items():
        truss = Truss(dim=2 if case in ("bar-10", "bar-47") else 3).LoadFromJSON(data=load_case(f"{case}_input_0"))
        truss.Solve()
        ts = []
        for _ in range(30):
            t0 = time.time()
            truss.Solve()
            ts.append(time.time() - t0)
        mean = sum(ts) / len(ts)
        out[case] = {"mean_ms": mean * 1e3, "reference_published_ms": ref_s * 1e3, "ratio": ref_s / mean}
    out["note"] = ("mean wall time of 30 Truss.Solve() calls per case through the drop-in API, against the reference's "
                   "published single-truss times (its README, other hardware: a laptop CPU); informational")
    return out


def repeat_stats(seconds, potrf_ms, steps, trusses_per_step):
    """min / median / max over the repeats of the timed K-step region (each repeat bracketed like the contract's
    region: barrier + synchronise on both sides, max over the ranks): ms per step, the whole-job rate, and the
    dominant kernel's average launch of each repeat (events on the launch stream)."""
    if not seconds:
        return {"count": 0}
    ms = sorted(s / steps * 1e3 for s in seconds)
    med = lambda v: float(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]))
    pk = sorted(potrf_ms)
    return {"count": len(ms), "steps_per_repeat": steps,
            "ms_per_step": {"min": ms[0], "median": med(ms), "max": ms[-1]},
            "value": {"min": trusses_per_step / (ms[-1] * 1e-3), "median": trusses_per_step / (med(ms) * 1e-3),
                      "max": trusses_per_step / (ms[0] * 1e-3), "unit": "solves/s"},
            "roofline_avg_launch_ms": {"min": pk[0], "median": med(pk), "max": pk[-1]},
            "note": "further repeats of the same timed region right behind the first one (`value` / `ms_per_step` of "
                    "the line are the FIRST region, as the contract asks)"}


class LateLegWatchdog:
    """Armed around one late (multi-stream, informational) leg: if the leg has not finished after `seconds`, the line
    assembled so far is printed - with a note naming the leg - and the process exits with status
    `LateLegWatchdog.EXIT_STATUS` (3): a device call that never returns is a fault of the run, whatever the line
    holds, and the caller must see it in the exit code.  A stalled device call never returns to Python, so nothing
    softer than `os._exit` can end it; everything the contract asks for is in the line by then.
    `put(key_setter)` is how the main thread stores the leg's result: it takes the same lock as the watchdog's dump,
    so the line is never serialised while it is being changed, and never printed twice."""

    EXIT_STATUS = 3

    def __init__(self, line, name, seconds):
        import threading
        self._lock = threading.Lock()
        self._finished = False
        self._wake = threading.Event()
        self._thread = threading.Thread(target=self._watch, args=(line, name, float(seconds)), daemon=True)
        self._thread.start()

    def _watch(self, line, name, seconds):
        if self._wake.wait(seconds):
            return
        with self._lock:
            if self._finished:      # the leg came back at the deadline: the main thread prints the line
                return
            out = json.loads(json.dumps(line))   # (a copy: nothing below touches the caller's dict)
            out["watchdog"] = (f"the informational leg `{name}` did not finish within {seconds:.0f} s (multi-stream "
                               f"work stalled on the device); the line was printed without it and the process ended "
                               f"with status {self.EXIT_STATUS}")
            print(json.dumps(out), flush=True)
            os._exit(self.EXIT_STATUS)

    def put(self, store):
        """Run `store()` (which writes the leg's result into the line) and disarm the watchdog, atomically with
        respect to its dump."""
        with self._lock:
            store()
            self._finished = True
        self._wake.set()

    def done(self):
        with self._lock:
            self._finished = True
        self._wake.set()


def choose_timing_group(dist, torch, device, rank, world, ndev):
    """The process group that carries the barrier around the timed region and the max-over-ranks of the elapsed time
    (there is no collective on the data path), chosen COLLECTIVELY: gloo comes up first and is the control plane;
    RCCL (backend "nccl") is tried only when every rank says over gloo that it is willing - it owns a GPU of its own
    (RCCL refuses duplicate devices) and `TRS_BENCH_NO_RCCL_RANK` does not name it (the test hook that makes one rank
    decline) -, and is used only when every rank says over gloo that its probe all-reduce went through.  Every rank
    therefore takes the same branch: a rank whose RCCL set-up fails can no longer leave the others waiting in a
    collective it never joins.  Returns (name, group or None = the default gloo group)."""
    def agree(flag):   # minimum over the ranks, on the gloo control plane
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    declines = os.environ.get("TRS_BENCH_NO_RCCL_RANK", "")
    willing = world <= ndev and str(rank) not in [r for r in declines.split(",") if r]
    if not agree(willing):
        return "gloo", None
    ok, group = True, None
    try:
        group = dist.new_group(backend="nccl")
        probe = torch.zeros([1], device=device)
        dist.all_reduce(probe, group=group)
        torch.cuda.synchronize(device)
    except Exception as exc:  # pragma: no cover - needs a broken RCCL set-up
        print(f"[bench rank {rank}] RCCL group failed ({exc!r})", file=sys.stderr)
        ok = False
    if agree(ok):
        return "nccl", group
    if rank == 0:
        print("[bench] RCCL did not come up on every rank; timing over gloo", file=sys.stderr)
    return "gloo", None


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) with
    torch.distributed.run as a CHILD process and pass its output and exit code through.  This process
    has made no GPU call at this point (and never replaces itself with exec)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # (a timed region of ~75 ms; the 25 repeats of it add ~2 s)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="trusses per GPU")
    ap.add_argument("--case", default="bar-942_input_0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pool-seconds", type=float, default=8.0,
                    help="sample length of the all-host-cores oracle baseline (0 = skip)")
    ap.add_argument("--cube-cpu-seconds", type=float, default=12.0,
                    help="sample length of the oracle baseline on the cube-truss distribution (0 = skip)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the informational PCIe-inclusive pass")
    ap.add_argument("--cube-batch", type=int, default=65536,
                    help="trusses per GPU of the mixed cube-truss leg, BASELINE config 3 (0 = skip)")
    ap.add_argument("--dataset-samples", type=int, default=131072,
                    help="samples per GPU of the dataset leg, BASELINE config 5 (0 = skip)")
    ap.add_argument("--cube-total", type=int, default=65536,
                    help="trusses IN ALL of the strong-scaled config-3 entry of `north_star_scaling` (0 = skip)")
    ap.add_argument("--dataset-total", type=int, default=1000000,
                    help="samples IN ALL of the strong-scaled, streamed config-5 entry of `north_star_scaling` (0 = skip)")
    ap.add_argument("--large-trusses", type=int, default=512,
                    help="trusses of the informational large-truss leg (8x8x8 grid, 300..400 cubes; 0 = skip)")
    ap.add_argument("--cube-steps", type=int, default=5)
    ap.add_argument("--cube-warmup", type=int, default=1)
    ap.add_argument("--dense", action="store_true",
                    help="treat every stiffness matrix as dense (no envelope tile skipping)")
    ap.add_argument("--no-dense-ref", action="store_true",
                    help="skip the dense-mode reference measurement of the factorisation (profiling runs)")
    ap.add_argument("--compact", action="store_true",
                    help="A/B: the stiffness matrix leaves the assembly as compact entry lists and the fused "
                         "factorisation forms the tiles from them (K_ff never dense in HBM) instead of the slab")
    ap.add_argument("--joint-order", default="profile", choices=("profile", "rcm", "given"),
                    help="numbering of the joints in the resident batch: 'profile' (default) = the cheapest of "
                         "RCM and the coordinate sweeps (trs_joint_order, csrc/order.hip, on the device, once per "
                         "topology, outside the timed region like the upload - the line also carries the rate with "
                         "the order INSIDE the step, `order_in_step`); results are delivered in the GIVEN numbering "
                         "either way (trs_recover's joint_out) and checked against the oracle in it")
    ap.add_argument("--late-leg-seconds", type=float, default=180.0,
                    help="watchdog of each informational multi-stream leg (dataset.streamed, cube_batch.host_fed): after "
                         "this long the line is printed without the leg")
    ap.add_argument("--repeats", type=int, default=25,
                    help="further repeats of the timed K-step region, right behind it, for the spread of the headline "
                         "(`repeats`: min / median / max; the contract's `value` stays the FIRST region)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible GPUs (ranks share devices round-robin; for testing "
                         "the multi-rank path on a 1-GPU box - the line then reports the devices really used)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    from python_stable_3d_truss_analysis_amd import batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # the native host helpers of this rank (generator, JSON reader, host joint order) get its share of the CPUs
    from python_stable_3d_truss_analysis_amd import generate as _gen
    _gen.set_host_thread_share(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    host_threads = _gen.host_threads()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    data = load_case(args.case)
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline and args.cpu_pool_seconds > 0:
        try:  # forked workers: only safe while this process has not initialised the GPU
            cpu_all = cpu_baseline_all_cores(data, args.cpu_pool_seconds)
        except Exception as exc:  # informational only
            cpu_all = {"error": repr(exc)}
    cube_cpu = None
    if world == 1 and not args.no_cpu_baseline and args.cube_batch > 0 and args.cube_cpu_seconds > 0:
        try:  # (its all-cores part forks: before the GPU is initialised)
            cube_cpu = cube_cpu_baseline(args.cube_batch, args.cube_cpu_seconds, pool_too=args.cpu_pool_seconds > 0)
        except Exception as exc:
            print(f"[bench] cube CPU baseline failed: {exc!r}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.oversubscribe:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} visible GPU(s) (one process per GPU; "
                         "--oversubscribe shares devices for testing)")
    dev_index = local_rank % ndev
    n_devices_used = min(world, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    timing_group, tgroup = None, None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        timing_group, tgroup = choose_timing_group(dist, torch, device, rank, world, ndev)

    def barrier():
        if distributed:
            if timing_group == "nccl":
                dist.barrier(group=tgroup, device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize(device)

    def reduce_max(seconds, op="max"):
        if not distributed:
            return seconds
        tmax = torch.tensor([seconds], dtype=torch.float64, device=device if timing_group == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN, group=tgroup)
        return float(tmax.item())

    def gather(obj):
        """`obj` of every rank on rank 0, in rank order (the gloo control plane; a list of one at N = 1)."""
        if not distributed:
            return [obj]
        out = [None] * world if rank == 0 else None
        dist.gather_object(obj, out, dst=0)
        return out

    # The legs of the other configurations run FIRST (each with its own barrier-bracketed region): they are seconds of
    # device work, after which the chip sits at its steady clocks - the headline's timed region, 27 ms of work behind a
    # handful of warm-up steps, otherwise starts on a device that is still ramping up (first region 2.4 % slower than
    # the median of its own 25 repeats; `leg_order` in the line says which way round it was).
    # the ragged workload of north_star (BASELINE config 3), on EVERY rank, with its own barrier-bracketed region
    cube = None
    if args.cube_batch > 0:
        torch.cuda.empty_cache()
        try:
            cube = cube_batch_leg(args, device, torch, batch, barrier, reduce_max, rank, world, cpu=cube_cpu)
        except Exception as exc:  # never lose the headline line over it
            cube = {"error": repr(exc)}
            if distributed:
                raise
    cube_strong = None
    if args.cube_total > 0 and not (world == 1 and args.cube_total == args.cube_batch and isinstance(cube, dict)
                                    and "value" in cube):
        cube_strong = config3_strong_leg(args, device, torch, batch, barrier, reduce_max, gather, rank, world)
    dataset = None
    if args.dataset_samples > 0:
        try:
            dataset = dataset_leg(args, device, torch, barrier, reduce_max, rank, world)
        except Exception as exc:
            dataset = {"error": repr(exc)}
            if distributed:
                raise
    data_strong = None
    if args.dataset_total > 0 and world > 1:   # (N = 1: a late leg under the watchdog, like the other multi-stream legs)
        data_strong = config5_strong_leg(args, device, torch, barrier, reduce_max, gather, rank, world)
    batch.release_workspaces()   # (the legs' slabs go back to the driver: the headline batch allocates on a clean device)

    packed = batch.pack_json([data]).replicate(args.batch)
    order = False if args.joint_order == "given" or args.dense else args.joint_order
    # The headline treats the copies as INDEPENDENT problems ("no dedup of identical inputs"): nothing that
    # a step learns about one truss is carried over to another truss or to a later step.  With the default joint order
    # ("profile") the batch is resident in the GIVEN numbering and every timed step finds, applies and prices the
    # order of every truss again (trs_joint_order, all sweeps) before the five stages, forms the tile masks again
    # (no `adopt_tile_hint`) and launches without launch hints (`all_narrow`: the kernels of wide envelopes are
    # launched and find nothing).  Rounds 1-5 reported the form with the order found once before the timed region and
    # the masks switched off after the first warm-up step: that figure is `extra.profile_order_hoisted` now.
    order_in_step = order == "profile"
    if order_in_step:
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        raw = {f: up(getattr(packed, f)) for f in batch.DeviceBatch.INPUT_FIELDS}
        ordered = batch.joint_order_device(torch, raw, effort=3)   # (allocates the order's outputs; what `reorder=True` does)
        tens = dict(raw)
        tens.update({k: ordered[k] for k in ("xyz", "conn", "cbits", "loads")})
        dev = batch.DeviceBatch.from_device(tens, packed.n_max, joint_out=ordered["perm"])
        if args.compact:
            dev.options["compact"] = True
        find_order = lambda: batch.joint_order_device(torch, raw, effort=3, out=ordered)
    else:
        dev = batch.DeviceBatch(packed, device, use_envelope=not args.dense, reorder=order,
                                options={"compact": True} if args.compact else None)
        find_order = lambda: None
    stages = (("order",) if order_in_step else ()) + STAGES
    n, nJ, nM = int(packed.n_free[0]), int(packed.nJ[0]), int(packed.nM[0])

    def step(events=None, potrf_events=None):
        calls = ((find_order,) if order_in_step else ()) + (dev.dofmap, dev.assemble, dev.potrf, dev.potrs, dev.recover)
        if potrf_events is not None:   # the timed step: the stages one by one, events around the dominant kernel
            find_order()
            dev.dofmap(); dev.assemble()
            potrf_events[0].record(); dev.potrf(); potrf_events[1].record()
            dev.potrs(); dev.recover()
            return
        if events is None:
            find_order()
            dev.solve()
            return
        for call, ev in zip(calls, events):
            ev[0].record()
            call()
            ev[1].record()

    warm_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for k in range(args.warmup):   # (the form the timed steps have: the stages one by one, events around the factorisation)
        step(potrf_events=warm_events)
        if k == 0 and not order_in_step:
            dev.adopt_tile_hint()   # (a hoisted order: bar-942 has too few envelope tiles without an entry of K_ff - the masks are not formed again)
    # Events are recorded on torch's current stream, which is the stream the C ABI launches on.  The TIMED
    # steps carry two events each, around the factorisation (roofline.avg_launch_ms); an event pair around
    # every stage costs 2.5 % of the step (tools/event_overhead.py), so the per-stage breakdown comes from an
    # equal number of instrumented steps right after the timed region.
    new_event = lambda: torch.cuda.Event(enable_timing=True)
    potrf_events = [(new_event(), new_event()) for _ in range(args.steps)]
    for e0, e1 in potrf_events:   # (an event object is created by the runtime at its first record(): not inside the timed
        e0.record()               # region - 2 x K creations were 1-2 % of a 27 ms region, which the repeats did not pay)
        e1.record()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(potrf_events=potrf_events[k])
    barrier()
    elapsed = time.perf_counter() - t0
    potrf_ms_timed = float(np.mean([e0.elapsed_time(e1) for e0, e1 in potrf_events]))
    # the spread of the headline: the same K-step region again and again (same calls, same brackets; the events of
    # the first region are recorded over - their times are taken already), max over the ranks per repeat
    repeat_s, repeat_potrf_ms = [], []
    for _ in range(max(0, args.repeats)):
        barrier()
        t0r = time.perf_counter()
        for k in range(args.steps):
            step(potrf_events=potrf_events[k])
        barrier()
        repeat_s.append(reduce_max(time.perf_counter() - t0r))
        repeat_potrf_ms.append(float(np.mean([e0.elapsed_time(e1) for e0, e1 in potrf_events])))
    all_events = [[(new_event(), new_event()) for _ in stages] for _ in range(args.steps)]
    for k in range(args.steps):
        step(all_events[k])
    torch.cuda.synchronize(device)

    elapsed_min = reduce_max(elapsed, "min")   # (the fastest rank: imbalance between the ranks shows in the line)
    elapsed = reduce_max(elapsed)

    stage_ms = {s: float(np.mean([all_events[k][i][0].elapsed_time(all_events[k][i][1])
                                  for k in range(args.steps)])) for i, s in enumerate(stages)}
    res = dev.result()

    # reference point outside the timed region (rank 0): the same kernels with the envelope switched
    # off, i.e. the dense factorisation that SURVEY section 8d's FLOP figure describes
    dense_ms = None
    # (the informational legs below run at N = 1 only: at N > 1 they would only keep the other ranks waiting)
    if world == 1 and not args.dense and not args.no_dense_ref:
        dense = batch.DeviceBatch(packed, device, use_envelope=False)
        dense.solve(); torch.cuda.synchronize(device)
        dense.dofmap(); dense.assemble()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dense.potrf(); e1.record(); torch.cuda.synchronize(device)
        dense_ms = e0.elapsed_time(e1)
        del dense

    # informational: the same timed loop with the joints in the numbering the data came in (rank 0, N = 1)
    given = None
    if world == 1 and order and not args.no_dense_ref:
        plain = batch.DeviceBatch(packed, device, use_envelope=True)
        for _ in range(args.warmup):
            plain.solve()
        plain.adopt_tile_hint()
        torch.cuda.synchronize(device)
        t0g = time.perf_counter()
        for _ in range(args.steps):
            plain.solve()
        torch.cuda.synchronize(device)
        dtg = time.perf_counter() - t0g
        given = {"value": args.batch * args.steps / dtg, "unit": "solves/s", "ms_per_step": dtg / args.steps * 1e3,
                 "note": "the same batch resident in the given joint numbering (--joint-order given as the headline)"}
        del plain

    # informational: what rounds 1-5 reported as the headline - the order found ONCE per topology before the timed
    # region, the tile masks switched off after the first warm-up step, launch hints adopted (rank 0, N = 1)
    hoisted = None
    if world == 1 and order_in_step and not args.no_dense_ref:
        hd = batch.DeviceBatch(packed, device, use_envelope=True, reorder="profile",
                               options={"compact": True} if args.compact else None)
        for k in range(max(1, args.warmup)):
            hd.solve()
            if k == 0:
                hd.adopt_tile_hint()
        torch.cuda.synchronize(device)
        t0h = time.perf_counter()
        for _ in range(args.steps):
            hd.solve()
        torch.cuda.synchronize(device)
        dth = time.perf_counter() - t0h
        same = bool(torch.equal(hd.u, dev.u) and torch.equal(hd.N, dev.N) and torch.equal(hd.f_ext, dev.f_ext))
        hoisted = {"value": args.batch * args.steps / dth, "unit": "solves/s", "ms_per_step": dth / args.steps * 1e3,
                   "results_bitwise_equal_to_headline": same,
                   "note": "the joint order found once per topology BEFORE the timed region, tile masks switched off "
                           "after the first warm-up step (adopt_tile_hint), launch hints adopted: the `value` of rounds "
                           "1-5; it shares work between the identical copies, which SURVEY 8d rules out for `value`"}
        del hd

    # informational: the same step fed from / drained to page-locked host memory over PCIe, upload of the
    # next batch, solve and download of the previous one overlapped on three streams (batch.StreamedSolver)
    def pcie_leg():
        pipe = batch.StreamedSolver(packed, device, slots=3, use_envelope=not args.dense, same_topology=True)
        src = pipe.host_in[0]
        for _ in range(3):          # warm-up: first use of the page-locked buffers, allocator, clocks
            pipe.submit(src)
        pipe.drain()
        torch.cuda.synchronize(device)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.submit(src)
        last = pipe.drain()[-1]
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps
        in_bytes = sum(v.numel() * v.element_size() for v in src.values())
        out_bytes = sum(v.numel() * v.element_size() for v in pipe.host_out[0].values())
        return {"solves_per_s": args.batch / dt, "ms_per_step": dt * 1e3,
                "h2d_bytes_per_truss": in_bytes // args.batch, "d2h_bytes_per_truss": out_bytes // args.batch,
                "h2d_plus_d2h_GBps": (in_bytes + out_bytes) / dt / 1e9,
                "info_nonzero": int((last.info != 0).sum()),
                "note": "steady state of batch.StreamedSolver: upload of all inputs, solve and download of u, "
                        "f_ext, N, info through page-locked host buffers on three streams (one DMA per direction "
                        "and batch), three resident batches; never the headline value"}


    pcie = None
    if world == 1 and not args.no_pcie:
        try:
            pcie = pcie_leg()
        except Exception as exc:
            pcie = {"error": repr(exc)}

    if rank == 0:
        total_trusses = world * args.batch * args.steps
        potrf_s = potrf_ms_timed * 1e-3   # the dominant kernel, measured inside the timed region
        potrf_kernel = "trs_potrf_kernel"
        if dev.env is not None:
            env = dev.env[0].cpu().numpy()
            nchm = dev.rows // 16
            env_ft, env_last = env[:nchm], env[nchm:nchm + dev.rows // 64]
            slack = int(env[nchm + dev.rows // 64])
            narrow = (slack & 0xff) == 1   # the kernel choice recorded by trs_assemble (csrc/trs_common.h)
            compact = bool(slack & 0x100)  # K_ff as compact entry lists, tiles formed in the factorisation
            rs = 4 if slack & 0x200 else 2  # items of four chunks for the wider ones of the narrow envelopes
            substituted = bool(slack & 0x400)  # the factorising wave substituted as well (trs_common.h)
            potrf_kernel = ("trs_potrf_narrow_kernel<true, 2>" if compact else f"trs_potrf_narrow_kernel<false, {rs}>") \
                if narrow else "trs_potrf_kernel"
            env_cend = env[nchm + dev.rows // 64 + 8: nchm + dev.rows // 64 + 8 + nchm]
            env_kmask = env[2 * nchm + dev.rows // 64 + 8: 3 * nchm + dev.rows // 64 + 8]
            tile_flops = potrf_tile_flops(n, env_ft, env_last, env_cend, narrow)
            counts = algorithmic_counts(n, nJ, nM, env_cend, narrow, env_kmask)
            if compact:  # the factorisation reads the entry lists instead of the stiffness tiles
                meta = env[nchm + dev.rows // 64: nchm + dev.rows // 64 + 8]
                work = dev.work.view(-1)
                tb_off, td_off = int(meta[2]) * 16, int(meta[1]) * 16
                ntile = int(work[tb_off + 4 * (npad_of(n) // 16): tb_off + 4 * (npad_of(n) // 16) + 4]
                            .cpu().numpy().view(np.int32)[0])
                last = work[td_off + 8 * (ntile - 1): td_off + 8 * ntile].cpu().numpy().view(np.int32)
                entries = int(last[0] + last[1])
                lists = 10 * entries + 8 * ntile + 4 * (npad_of(n) // 16 + 1)
                counts["compact_list_bytes"] = lists
                counts["assemble_bytes"] += lists - counts["slab_tile_bytes"]
                counts["potrf_bytes"] += lists - counts["slab_tile_bytes"]
            if substituted:  # one kernel does both stages: its bytes are those of both
                counts["potrf_bytes"] += counts["potrs_bytes"]
        else:
            substituted = False
            tile_flops = potrf_tile_flops(n)
            counts = algorithmic_counts(n, nJ, nM)
        dense_flops = algorithmic_counts(n, nJ, nM)["potrf_flops"]
        achieved_tflops = tile_flops * args.batch / potrf_s / 1e12
        asm_gbs = counts["assemble_bytes"] * args.batch / (stage_ms["assemble"] * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "potrf_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                rec = json.load(fh)
            # the PMC pass is valid for the configuration AND the kernel sources it was taken on only
            if rec.get("envelope") == (not args.dense) and rec.get("batch") == args.batch and \
                    rec.get("joint_order", "given") == (order or "given") and \
                    rec.get("source_sha") == kernel_source_sha() and rec.get("kernel") == potrf_kernel:
                traffic = rec.get("hbm_bytes_per_launch")
        # Which roof bounds the factorisation: its arithmetic intensity (executed FLOP per byte of the
        # stored slab part read once + written once) against the machine balance peak_flops / peak_bw.
        potrf_bytes = counts["potrf_bytes"]
        potrf_gbs = potrf_bytes * args.batch / potrf_s / 1e9
        intensity = tile_flops / potrf_bytes
        balance = PEAK_FP64_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        mfma_frac, hbm_frac = achieved_tflops / PEAK_FP64_TFLOPS, potrf_gbs / PEAK_HBM_GBS
        roofline = {"kernel": potrf_kernel, "traffic": traffic, "avg_launch_ms": potrf_ms_timed,
                    "flop_per_truss": tile_flops, "bytes_per_truss": potrf_bytes,
                    "intensity_flop_per_byte": intensity, "machine_balance_flop_per_byte": balance,
                    "mfma": {"achieved": achieved_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                             "frac": mfma_frac},
                    "hbm": {"achieved": potrf_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac},
                    "flop_model": "MFMA work inside the 16x16-tile envelope of K_ff (what the kernel "
                                  "executes; equals the dense tile count with --dense)",
                    "byte_model": "stiffness tiles inside the envelope that hold an entry of K_ff (or their compact "
                                  "entry lists) read once, factor tiles written once, load vector in / out (uf)" +
                                  ("; the same wave then substitutes: factor tiles read once more, y in, u out "
                                   "(the trs_potrs launch finds nothing left to do)" if substituted else ""),
                    "fused_substitution": substituted,
                    # SURVEY 8d's stricter count for the factorisation alone - stiffness tiles read once + factor tiles
                    # written once, the substitution's read-back of the factor NOT counted - over the same launch (which
                    # does substitute): the lower bound of what one may call "algorithmic" for this kernel
                    "hbm_read_k_write_l": {"bytes_per_truss": int(counts["stiffness_tile_bytes"] + counts["slab_tile_bytes"]),
                                           "frac": (counts["stiffness_tile_bytes"] + counts["slab_tile_bytes"]) * args.batch
                                           / potrf_s / 1e9 / PEAK_HBM_GBS},
                    "dense_equivalent_tflops": dense_flops * args.batch / potrf_s / 1e12,
                    "dense_flop_per_truss": dense_flops}
        if intensity >= balance:
            roofline.update(bound="mfma", achieved=achieved_tflops, peak=PEAK_FP64_TFLOPS,
                            unit="TFLOP/s", frac=mfma_frac)
        else:
            roofline.update(bound="hbm", achieved=potrf_gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=hbm_frac)
        def hbm_stage(stage, bytes_per_truss):  # algorithmic bytes of a stage against the HBM roof
            gbs = bytes_per_truss * args.batch / (stage_ms[stage] * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": gbs / PEAK_HBM_GBS, "bytes_per_truss": bytes_per_truss}

        line = {
            "metric": "truss solves/sec (batched Solve)",
            "value": total_trusses / elapsed,
            "unit": "solves/s",
            "n_gpus": n_devices_used,
            "ranks": world,
            "timing_group": timing_group,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "rank_ms_per_step": {"min": elapsed_min / args.steps * 1e3, "max": elapsed / args.steps * 1e3},
            "repeats": repeat_stats(repeat_s, repeat_potrf_ms, args.steps, world * args.batch),
            "host_threads_per_rank": host_threads,
            "leg_order": "cube_batch, dataset, then the headline (config 2) and the informational legs",
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),   # (set to 8 by the package unless the caller chose)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic: bundled bar-942 truss replicated as independent problems",
            "config": {"workload": f"{args.case} x {args.batch} independent copies per GPU "
                                   f"(nJ {nJ}, nM {nM}, n_free {n})",
                       "batch_per_gpu": args.batch, "parallelism": f"batch-sharded x{world}, no collective",
                       "joint_order": ("profile: trs_joint_order on the device INSIDE the timed step, for every truss "
                                       "of every step (all coordinate sweeps, priced by the tile envelope); tile masks "
                                       "formed every step, no launch hints; results in the given numbering"
                                       if order_in_step else
                                       (order or "given") + (" (found once per topology, before the timed region; results "
                                                             "in the given numbering)" if order else ""))},
            "roofline": roofline,
            "stages_ms": stage_ms,
            "stages_ms_note": "event pairs around every stage, measured over an equal number of instrumented steps "
                              "right after the timed region (the timed steps carry events around the factorisation "
                              "only: roofline.avg_launch_ms)",
            "assemble_roofline": {"bound": "hbm", "achieved": asm_gbs, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": asm_gbs / PEAK_HBM_GBS,
                                  "bytes_per_truss": counts["assemble_bytes"],
                                  "note": "bytes of the slab part that is stored (upper 16-row tiles inside the "
                                          "envelope + load column) + inputs; the full symmetric dense "
                                          f"figure of SURVEY 8d would be {counts['assemble_bytes_full_contract']} B"},
            "potrs_roofline": ({"note": "substituted by the factorising wave: see roofline (its bytes are counted "
                                        "there); the stage time is the launch that finds no work",
                                "avg_launch_ms": stage_ms["potrs"]} if substituted
                               else hbm_stage("potrs", counts["potrs_bytes"])),
            "recover_roofline": hbm_stage("recover", counts["recover_bytes"]),
            "info_nonzero": int((res.info != 0).sum()),
            "envelope": not args.dense,
        }
        # north_star's last sentence in ONE record: config 2 weak-scaled (the headline), config 3 and config 5
        # strong-scaled, each with the spread over the ranks - what a driver that runs this command at N = 1, 2, 4, 8
        # tabulates
        scaling = {"n_gpus": n_devices_used, "ranks": world,
                   "config2_weak": {"config": f"BASELINE config 2: {args.case} x {args.batch} independent copies per GPU",
                                    "scaling": "weak", "value": line["value"], "unit": "solves/s",
                                    "ms_per_step": line["ms_per_step"], "rank_ms_per_step": line["rank_ms_per_step"]}}
        if cube_strong is not None:
            scaling["config3_strong"] = cube_strong
        elif args.cube_total > 0 and isinstance(cube, dict) and "value" in cube:
            scaling["config3_strong"] = {
                "config": "BASELINE config 3: mixed cube trusses (grid 6x6x6, 8..190 cubes), one ragged batch",
                "scaling": "strong", "total_trusses": int(args.cube_total), "steps": args.cube_steps,
                "value": cube["value"], "unit": "solves/s", "ms_per_step": cube["ms_per_step"],
                "rank_ms_per_step": cube["rank_ms_per_step"],
                "shards": [{"rank": 0, "first": 0, "count": int(args.cube_total), "info_nonzero": cube["info_nonzero"]}],
                "info_nonzero": cube["info_nonzero"],
                "note": "at N = 1 the strong-scaled entry IS the `cube_batch` leg (same trusses, same steps)"}
        if data_strong is not None:
            scaling["config5_strong"] = data_strong
        line["north_star_scaling"] = scaling
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if cube is not None:
            line["cube_batch"] = cube
        if dataset is not None:
            line["dataset"] = dataset
        if world == 1 and not args.no_pcie:   # (the informational legs' switch)
            try:
                line["ga_generation"] = ga_leg(device, torch)
            except Exception as exc:
                line["ga_generation"] = {"error": repr(exc)}
            if args.large_trusses > 0:
                try:
                    line["large_truss"] = large_truss_leg(args, device, torch, batch)
                except Exception as exc:
                    line["large_truss"] = {"error": repr(exc)}
                batch.release_workspaces()
            try:
                line["reference_protocol"] = reference_protocol_leg()
            except Exception as exc:
                line["reference_protocol"] = {"error": repr(exc)}
        if given is not None:
            line["given_joint_order"] = given
        if hoisted is not None:
            line["extra"] = {"profile_order_hoisted": hoisted}
        if dense_ms is not None:
            line["dense_mode_potrf"] = {
                "avg_launch_ms": dense_ms,
                "achieved_tflops": dense_flops * args.batch / (dense_ms * 1e-3) / 1e12,
                "frac_of_peak": dense_flops * args.batch / (dense_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                "note": "same kernel with the envelope off: n^3/3 + 2 n^2 FLOP per truss (SURVEY 8d)"}
            line["roofline"]["frac_dense_8d"] = line["dense_mode_potrf"]["frac_of_peak"]
            line["roofline"]["frac_dense_8d_note"] = (
                "SURVEY 8d's dense figure (n^3/3 + 2 n^2 FLOP per truss) over the DENSE-mode launch of the same "
                "factorisation kernel (envelope off), as a fraction of the FP64 MFMA peak; the headline kernel "
                "skips the structural zeros of K_ff and is priced by its own bytes (frac)")
        if not args.no_cpu_baseline and world == 1:  # the oracle leg (rank 0 at N = 1 only): CPU baseline + check
            line["cpu_baseline"], ref = cpu_baseline(data, args.cpu_seconds)
            if cpu_all is not None:
                line["cpu_baseline_all_cores"] = cpu_all
            line["max_rel_err_vs_oracle"] = {   # every truss of the batch (independent copies of one problem)
                "u": float(np.abs(res.displace[:, :nJ] - ref["u"][None]).max() / np.abs(ref["u"]).max()),
                "N": float(np.abs(res.internal[:, :nM] - ref["N"][None]).max() / np.abs(ref["N"]).max()),
                "trusses_checked": int(res.displace.shape[0])}
        # The legs that drive SEVERAL streams (informational, N = 1) run last, each under a watchdog: this runtime has
        # stopped multi-stream work on the device without an error (EXPERIMENTS R4.9) - if one of them does not come
        # back, the line is printed without it and the process ends instead of hanging the caller.
        if world == 1:
            late = []
            if args.dataset_total > 0:   # (first: the one late leg the scaling record needs)
                late.append(("north_star_scaling.config5_strong",
                             lambda: config5_strong_leg(args, device, torch, barrier, reduce_max, gather, rank, world),
                             lambda v: scaling.__setitem__("config5_strong", v)))
            if args.no_pcie:   # (the informational legs' switch)
                pass
            elif isinstance(dataset, dict) and "value" in dataset:
                late.append(("dataset.streamed", lambda: dataset_streamed_leg(args, device, dataset["value"]),
                             lambda v: dataset.__setitem__("streamed", v)))
            if not args.no_pcie and isinstance(cube, dict) and "value" in cube:
                late.append(("cube_batch.host_fed", lambda: host_fed_leg(args, device, torch, batch),
                             lambda v: cube.__setitem__("host_fed", v)))
            for name, run, put in late:
                guard = LateLegWatchdog(line, name, args.late_leg_seconds)
                try:
                    value = run()
                except Exception as exc:   # (an informational leg must not take the line down)
                    value = {"error": repr(exc)}
                guard.put(lambda: put(value))
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
