#!/usr/bin/env python3
"""What the joint-order kernel costs the ragged cube STEP: the step as shipped (per bucket
trs_joint_order_rows: search + renumbering, inside the step) against the same step with the SAME permutations handed in
(`reorder=<array>`: the renumbered inputs are resident, every bucket gathers its rows with a copy launch instead of
searching) - i.e. with the search knocked out and only its data movement left.  Alternated in one process, results
compared bit for bit.      python tools/order_knockout.py [rounds] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
first = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
first.step(); torch.cuda.synchronize()
perm = np.tile(np.arange(sizes.nJ_max, dtype=np.int32), (sizes.B, 1))
for bk in first.buckets:
    if bk["order_on_device"]:
        p = bk["ordered"]["perm"].cpu().numpy()
        perm[np.asarray(bk["idx"])[:, None], np.arange(p.shape[1])[None, :]] = p
ref = (first.u.clone(), first.N.clone())
del first
batch.release_workspaces()
configs = {"search inside the step": True, "permutations given": perm}
times = {k: [] for k in configs}
stages = {}
for r in range(rounds):
    for name, reorder in configs.items():
        solver = batch.RaggedSolver(sizes, reorder=reorder, tensors=tensors)
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / steps * 1e3)
        assert torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]), name
        rec = []
        solver.step(record=rec); torch.cuda.synchronize()
        tot = {}
        for stage, e0, e1 in rec:
            tot[stage] = tot.get(stage, 0.0) + e0.elapsed_time(e1)
        stages[name] = tot
        del solver
        batch.release_workspaces()
for name in configs:
    t = times[name]
    print(f"{name:26s} " + " ".join(f"{v:.2f}" for v in t) + f"   median {np.median(t):.2f} ms per step; stage sums over the "
          "buckets (lanes overlap): " + ", ".join(f"{k} {v:.2f}" for k, v in stages[name].items()), flush=True)
print("results bitwise equal")
