#!/usr/bin/env python3
"""Steps of the ragged cube-truss workload (BASELINE config 3: batch.RaggedSolver on 65 536 generated trusses) for
profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/cube_step.py [--cubes 65536] [--steps 3]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi
if os.environ.get("VARIANT"):   # a variant build (tools/build_variants.sh) instead of the product library
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{os.environ['VARIANT']}.so")
from python_stable_3d_truss_analysis_amd import batch

ap = argparse.ArgumentParser()
ap.add_argument("--cubes", type=int, default=65536)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--slab-gb", type=int, default=0, help="max_slab_bytes of the solver, GiB (all lanes together; 0 = the solver's default)")
ap.add_argument("--lanes", type=int, default=None, help="streams the buckets are dealt onto (default: the solver's)")
args = ap.parse_args()
sizes, tensors = bench.cube_workload(args.cubes, 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=args.lanes, max_slab_bytes=(args.slab_gb << 30) or None)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
for _ in range(args.steps):
    solver.step()
torch.cuda.synchronize()
print("info_nonzero", int((solver.info != 0).sum().item()))
# per-stage times of one step (events on the buckets' streams), summed over the buckets (lanes overlap: the sums exceed the step)
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
print(f"options {solver.buckets[-1]['dev'].options}, {solver.lanes} lanes, {len(solver.buckets)} buckets: step {dt * 1e3:.2f} ms = {args.cubes / dt / 1e6:.3f} M solves/s; "
      + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()))
