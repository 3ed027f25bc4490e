#!/usr/bin/env python3
"""The ragged cube step by (lanes, slab budget): every configuration measured several times in alternation inside one
process (box-to-box and minute-to-minute drifts are larger than the differences looked for), results compared bit for
bit with the first.    python tools/lanes_matrix.py [rounds] [steps]        VARIANT=tag runs a variant build"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi
if os.environ.get("VARIANT"):
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{os.environ['VARIANT']}.so")
from python_stable_3d_truss_analysis_amd import batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
configs = [tuple(int(v) for v in c.split("x")) for c in os.environ.get(
    "CONFIGS", "1x48 1x144 1x200 2x144 3x96 3x144 4x48 4x144 4x200 6x144").split()]
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
ref, times, nb = None, {c: [] for c in configs}, {}
for r in range(rounds):
    for cfg in configs:
        lanes, gib = cfg
        solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=lanes, max_slab_bytes=gib << 30)
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        torch.cuda.synchronize()
        times[cfg].append((time.perf_counter() - t0) / steps * 1e3)
        nb[cfg] = len(solver.buckets)
        if ref is None:
            ref = (solver.u.clone(), solver.N.clone())
        assert torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]), cfg
        del solver
        batch.release_workspaces()
for cfg in configs:
    t = times[cfg]
    print(f"{cfg[0]} lanes, {cfg[1]:3d} GiB, {nb[cfg]:2d} buckets: " + " ".join(f"{v:.2f}" for v in t) +
          f"   median {np.median(t):.2f} ms", flush=True)
print("all configurations bitwise equal")
