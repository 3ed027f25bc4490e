#!/usr/bin/env python3
"""Cost of the per-stage events and separate C calls of bench.py's step against one trs_solve call."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from python_stable_3d_truss_analysis_amd import batch
data = json.load(open("tests/golden/data/bar-942_input_0.json"))
dev = batch.DeviceBatch(batch.pack_json([data]).replicate(4096))
calls = (dev.dofmap, dev.assemble, dev.potrf, dev.potrs, dev.recover)
def run(mode, steps=20):
    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in calls] for _ in range(steps)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        if mode == "solve": dev.solve()
        elif mode == "calls":
            for c in calls: c()
        elif mode == "potrf_ev":
            dev.dofmap(); dev.assemble(); evs[k][2][0].record(); dev.potrf(); evs[k][2][1].record(); dev.potrs(); dev.recover()
        else:
            for c, e in zip(calls, evs[k]):
                e[0].record(); c(); e[1].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return 4096 * steps / dt
for m in ("solve", "calls", "potrf_ev", "all_ev", "solve"):
    run(m, 3)
    print(m, round(run(m)))
