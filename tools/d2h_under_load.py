#!/usr/bin/env python3
"""Device -> page-locked host copy rate (hipMemcpyAsync on a second stream) alone and WHILE the cube-truss solver
runs on the main stream; and what the copy costs the solver.  One 2.3 GB buffer, in 1 / 8 pieces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

dev = torch.device("cuda:0")
n = 2_300_000_000 // 4
src = torch.empty([n], dtype=torch.float32, device=dev).normal_()
dst = torch.empty([n], dtype=torch.float32, pin_memory=True)
side = torch.cuda.Stream(dev)

def copy(pieces):
    step = n // pieces
    with torch.cuda.stream(side):
        for p in range(pieces):
            dst[p * step:(p + 1) * step].copy_(src[p * step:(p + 1) * step], non_blocking=True)

def timed_copy(pieces):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side)
    copy(pieces)
    with torch.cuda.stream(side):
        e1.record(side)
    return e0, e1

sizes, tensors = bench.cube_workload(32768, 0, device=dev)
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
solver.step(); torch.cuda.synchronize(); solver.adopt_launch_hints()
for pieces in (1, 8):
    copy(pieces); torch.cuda.synchronize()
    e0, e1 = timed_copy(pieces); torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    t0 = time.perf_counter(); solver.step(); solver.step(); torch.cuda.synchronize(); step_alone = (time.perf_counter() - t0) / 2 * 1e3
    t0 = time.perf_counter()
    solver.step(); solver.step(); solver.step()
    e0, e1 = timed_copy(pieces)
    torch.cuda.synchronize()
    both = (time.perf_counter() - t0) * 1e3
    print(f"{pieces} piece(s): copy alone {alone:.1f} ms = {n * 4 / alone / 1e6:.1f} GB/s; beside 3 solver steps {e0.elapsed_time(e1):.1f} ms = "
          f"{n * 4 / e0.elapsed_time(e1) / 1e6:.1f} GB/s; solver step alone {step_alone:.1f} ms, 3 steps + copy {both:.1f} ms")
