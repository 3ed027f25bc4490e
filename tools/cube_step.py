#!/usr/bin/env python3
"""Steps of the ragged cube-truss workload (BASELINE config 3: batch.RaggedSolver on 65 536 generated trusses) for
profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/cube_step.py [--cubes 65536] [--steps 3]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

ap = argparse.ArgumentParser()
ap.add_argument("--cubes", type=int, default=65536)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
sizes, tensors = bench.cube_workload(args.cubes, 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
for _ in range(args.steps):
    solver.step()
torch.cuda.synchronize()
print("info_nonzero", int((solver.info != 0).sum().item()))
# per-stage times of one step (events on the stream), summed over the buckets
import time
rec = []
solver.step(record=rec); torch.cuda.synchronize()
tot = {}
for name, e0, e1 in rec:
    tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
t0 = time.perf_counter()
for _ in range(args.steps):
    solver.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f"options {solver.buckets[-1]['dev'].options}: step {dt * 1e3:.2f} ms = {args.cubes / dt / 1e6:.3f} M solves/s; "
      + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()))
