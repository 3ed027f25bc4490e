#!/usr/bin/env python3
"""Speed calibration of the CPU oracle against the REAL reference (build container only: imports /root/reference
with the shims of tests/golden/make_golden.py; nothing of the reference is written anywhere).

`bench.py`'s `cpu_baseline` is the oracle (`oracle/truss_oracle.py`, kind "port") because the reference's Python
cannot travel to the GPU box.  BASELINE.md section 4(b) asks how its speed relates to the reference's own
`Truss.Solve()`: this script times both on the same inputs, same process, one BLAS thread, with the reference's
published protocol (`example.py:1-25`: load once, 30 x `Solve()`, mean) and writes

    tests/golden/oracle_speed.json   {case: {reference_ms, oracle_ms, ratio = oracle_ms / reference_ms}, ...}

`bench.py` quotes the ratios as `cpu_baseline.reference_speed_ratio` (a ratio > 1 means the reported baseline
UNDER-states the reference by that factor).

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_oracle.py
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np

RUNS = 30   # reference example.py:1-25


def mean_ms_pair(fn_a, fn_b, runs=RUNS):
    """Means of `runs` calls of each, INTERLEAVED (a, b, a, b, ...) so that clock drift of a shared host hits both."""
    fn_a(); fn_b()
    ta = tb = 0.0
    for _ in range(runs):
        t0 = time.time(); fn_a(); t1 = time.time(); fn_b(); t2 = time.time()
        ta += t1 - t0
        tb += t2 - t1
    return 1e3 * ta / runs, 1e3 * tb / runs


def main():
    from make_golden import import_reference
    rt, rty, rg, rga = import_reference()
    from oracle import truss_oracle as orc
    from python_stable_3d_truss_analysis_amd import generate as gen
    from python_stable_3d_truss_analysis_amd.data import dataset_sizes
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    out = {}
    data_dir = os.path.join(ROOT, "tests", "golden", "data")
    cases = [("bar-6", 3), ("bar-10", 2), ("bar-25", 3), ("bar-47", 2), ("bar-72", 3), ("bar-120", 3), ("bar-942", 3)]
    inputs = []
    for case, dim in cases:
        with open(os.path.join(data_dir, f"{case}_input_0.json")) as fh:
            inputs.append((case, dim, json.load(fh), RUNS))
    # the cube-truss distribution of bench.py's cube leg (config 3): the first trusses of its batch whose sizes fall
    # into eight size classes, solved by both
    sizes = dataset_sizes(7, 0, 4096, (8, 190))
    picks = [int(np.nonzero((sizes >= lo) & (sizes < lo + 23))[0][0]) for lo in range(8, 190, 23)]
    for i in picks:
        pk = gen.generate_cube_batch(sizes[i:i + 1], gridRange=(6, 6, 6), seed=7, first_index=i)
        inputs.append((f"cube[{i}] ({int(sizes[i])} cubes, {int(pk.nM[0])} members, n {int(pk.n_free[0])})", 3,
                       gen.packed_to_json(pk, 0), 5))
    for case, dim, data, runs in inputs:
        truss = rt.Truss(dim)
        truss.LoadFromJSON(data=data)
        prepared = orc.prepare(data)     # the truss as the reference holds it once loaded (lengths cached)
        ref_ms, orc_ms = mean_ms_pair(truss.Solve, lambda: orc.solve(prepared), runs)
        out[case] = {"reference_ms": ref_ms, "oracle_ms": orc_ms, "ratio": orc_ms / ref_ms, "runs": runs}
        print(f"{case:60s} reference {ref_ms:9.3f} ms   oracle {orc_ms:9.3f} ms   ratio {orc_ms / ref_ms:5.2f}")
    cube = [v["ratio"] for k, v in out.items() if k.startswith("cube")]
    out["_summary"] = {
        "bar-942": out["bar-942"]["ratio"], "cube_mean": float(np.mean(cube)),
        "note": "ratio = oracle time / reference time, same inputs, same process, 1 BLAS thread, mean of `runs` "
                "Solve() calls after one warm-up (reference protocol example.py:1-25); measured in the build "
                "container (8 vCPU Xeon 2.1 GHz), numpy " + np.__version__}
    path = os.path.join(ROOT, "tests", "golden", "oracle_speed.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
