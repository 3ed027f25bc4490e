#!/usr/bin/env python3
"""The compact form bucket by bucket: every staged bucket of the 65 536-truss cube batch solved alone on the chip
(one lane) in slab form and in compact form, ms per bucket from events, in alternation; results compared bit for bit.
    python tools/compact_buckets.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=1)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
solver.step(); torch.cuda.synchronize()
nJ_full, nM_full = int(solver.u.shape[1]), int(solver.N.shape[1])
ref_u, ref_N = solver.u.clone(), solver.N.clone()
print(f"{'rows':>5s} {'trusses':>8s} {'slab ms':>9s} {'compact ms':>11s} {'ratio':>6s}")
tot = [0.0, 0.0]
for bk in solver.buckets:
    db = bk["dev"]
    if db.small or not bk["fused_io"]:
        continue
    ms = {False: [], True: []}
    for r in range(reps):
        for compact in (False, True):
            db.options["compact"] = compact
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            db.solve_rows(bk["rows"], solver.outs[0], nJ_full, nM_full)
            e1.record()
            torch.cuda.synchronize()
            ms[compact].append(e0.elapsed_time(e1))
    db.options["compact"] = False
    a, b = float(np.median(ms[False])), float(np.median(ms[True]))
    tot[0] += a; tot[1] += b
    print(f"{db.rows:5d} {bk['count']:8d} {a:9.3f} {b:11.3f} {b / a:6.3f}", flush=True)
print(f"{'sum':>5s} {'':8s} {tot[0]:9.3f} {tot[1]:11.3f} {tot[1] / tot[0]:6.3f}")
print("bitwise equal to the slab step:", bool(torch.equal(solver.u, ref_u) and torch.equal(solver.N, ref_N)))
