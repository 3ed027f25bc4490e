#!/usr/bin/env python3
"""The ragged cube step with its buckets dealt onto several streams (`RaggedSolver(lanes=)`, opt-in): time per step by
workspace budget and number of lanes, results compared bit for bit with one lane (EXPERIMENTS R4.9).
    python tools/two_stream_step.py [budget_GiB ...]      (default: 48 144)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

budgets = [int(a) for a in sys.argv[1:]] or [48, 144]
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
ref = None
for gib in budgets:
    for lanes in (1, 2, 3, 4):
        solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=lanes, max_slab_bytes=gib << 30)
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            solver.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None:
            ref = (solver.u.clone(), solver.N.clone())
        same = bool(torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]))
        print(f"budget {gib:3d} GiB, {solver.lanes} lanes, {len(solver.buckets):2d} buckets: {ms:.2f} ms per step (bitwise equal to the "
              f"first: {same})", flush=True)
        del solver
        batch.release_workspaces()
