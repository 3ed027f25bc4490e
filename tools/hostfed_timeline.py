#!/usr/bin/env python3
"""Timeline of one host-fed step (`RaggedSolver(host_io=...)`) of the 65 536 cube trusses in the table member form:
set-up time, then per bucket when its pull, its device work and its push end (ms from the start of the step).
    python tools/hostfed_timeline.py [lanes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
host = sizes.to_packed(tensors)
del tensors
torch.cuda.empty_cache()
pinned = (host.table() if os.environ.get("FORM", "table") == "table" else host).pinned()
pool = batch.ResultPool(tracked=True)
dev = torch.device("cuda:0")
fields = batch.RaggedSolver.GATHER_TABLE if pinned.is_table else batch.RaggedSolver.GATHER
host_in = {f: torch.from_numpy(getattr(pinned, f)) for f in fields}
if pinned.is_table:
    host_in["types"] = torch.from_numpy(pinned.types)
host_out = batch.host_result_arrays(torch, pool, sizes.B, pinned.nJ_max, pinned.nM_max, dev)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    solver = batch.RaggedSolver(pinned, dev, reorder=True, max_slab_bytes=48 << 30, host_io=(host_in, host_out), lanes=lanes)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    solver.step(); torch.cuda.synchronize(); t2 = time.perf_counter()
    solver.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"set-up {1e3 * (t1 - t0):.1f} ms, first step {1e3 * (t2 - t1):.1f} ms, second step {1e3 * (t3 - t2):.1f} ms "
          f"({sizes.B / (t3 - t2) / 1e6:.3f} M solves/s as a resident host-fed solver), {solver.lanes} run lane(s)")
rec = []
solver.step(record=rec); torch.cuda.synchronize()
rows = {}
for name, e0, e1 in rec:
    k = int(name.split()[1])
    rows.setdefault(k, {})[" ".join(name.split()[2:])] = e0.elapsed_time(e1)
nJ, nM = sizes.nJ.astype(np.int64), sizes.nM.astype(np.int64)
per_m = 5 if pinned.is_table else 24
print(f"{'bucket':>6s} {'trusses':>8s} {'rows':>5s} {'up MB':>7s} {'down MB':>8s} | {'pull':>13s} | {'run':>13s} | {'push':>13s}   (begin - end, ms)")
for k, bk in enumerate(solver.buckets):
    idx = bk["idx"]
    up = (49 * nJ[idx] + per_m * nM[idx]).sum() / 1e6
    down = (48 * nJ[idx] + 8 * nM[idx]).sum() / 1e6
    r = rows[k]
    print(f"{k:6d} {bk['count']:8d} {bk['dev'].rows:5d} {up:7.1f} {down:8.1f} | {r['pull begins']:6.1f}-{r['pulled']:6.1f} | "
          f"{r['run begins']:6.1f}-{r['solved']:6.1f} | {r['push begins']:6.1f}-{r['pushed']:6.1f}")
