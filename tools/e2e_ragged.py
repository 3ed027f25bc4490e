#!/usr/bin/env python3
"""End-to-end wall time of solve_batch on a mixed cube-truss batch held in host arrays (generate ->
[RCM] -> bucket -> upload -> solve -> download -> scatter back), against the resident-batch GPU time."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch, generate as gen

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
packed = gen.generate_cube_batch(rng.integers(8, 191, size=B), gridRange=(6, 6, 6), seed=7)
batch.solve_batch(packed.take(np.arange(64)))            # warm: library load, first launches
torch.cuda.synchronize()
for reorder in (False, True):
    for attempt in ("first call (allocator warm-up)", "second call"):
        t0 = time.perf_counter()
        res = batch.solve_batch(packed, reorder=reorder)
        dt = time.perf_counter() - t0
        print(f"reorder={reorder}, {attempt}: {dt:.3f} s end to end = {B / dt:.0f} solves/s, "
              f"info_nonzero={int((res.info != 0).sum())}")
pinned, pool = packed.pinned(), batch.ResultPool(tracked=True)
for reorder in (True, "fast", "rcm"):
    for attempt in ("first call (pool allocation)", "second call", "third call"):
        t0 = time.perf_counter()
        res = batch.solve_batch(pinned, reorder=reorder, pool=pool)
        dt = time.perf_counter() - t0
        print(f"pinned inputs + result pool, reorder={reorder}, {attempt}: {dt:.3f} s end to end = {B / dt:.0f} solves/s, "
              f"info_nonzero={int((res.info != 0).sum())}")
cProfile.run("batch.solve_batch(pinned, reorder=True, pool=pool)", "/tmp/e2e.prof")
pstats.Stats("/tmp/e2e.prof").sort_stats("cumtime").print_stats(14)
