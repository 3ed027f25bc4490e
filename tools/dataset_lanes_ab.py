#!/usr/bin/env python3
"""Config 5 by the number of lanes of the chunk solvers: `data.dataset_chunks` (resident) and `data.dataset_stream`
(delivered in host memory) with `batch.DEFAULT_LANES` = 1 / 2 / 4, alternating in one process.
    python tools/dataset_lanes_ab.py [samples] [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from python_stable_3d_truss_analysis_amd import MemberType, TaskType, batch, data as gdata

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chunk = 32768
kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
          taskType=TaskType.REGRESSION, device=torch.device("cuda:0"), forceScale=1e3, displaceScale=0.1, positionScale=100.)
for _ in gdata.dataset_chunks(chunk, rank=0, world=1, chunk=chunk, **kw):
    pass
for _ in gdata.dataset_stream(2 * chunk, rank=0, world=1, chunk=chunk, **kw):
    pass
for r in range(rounds):
    for lanes in [int(v) for v in os.environ.get("LANES", "1 2 4").split()]:
        batch.DEFAULT_LANES = lanes
        for _ in gdata.dataset_chunks(chunk, rank=0, world=1, chunk=chunk, **kw):   # (the lane workspaces of this setting)
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for first, packed, tensors in gdata.dataset_chunks(samples, rank=0, world=1, chunk=chunk, **kw):
            bad = int(tensors["info"].ne(0).sum().item())
        torch.cuda.synchronize()
        resident = samples / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for graphs in gdata.dataset_stream(samples, rank=0, world=1, chunk=chunk, **kw):
            bad = int(graphs.tensors["info"].ne(0).sum().item())
        streamed = samples / (time.perf_counter() - t0)
        print(f"round {r}, {lanes} lanes: resident {resident / 1e3:.1f} K samples/s, streamed {streamed / 1e3:.1f} K samples/s "
              f"({streamed / resident:.2f})", flush=True)
