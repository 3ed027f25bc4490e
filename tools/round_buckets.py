#!/usr/bin/env python3
"""Buckets cut at whole ROUNDS of the factorisation kernel (3 waves x 4 SIMDs x 256 CUs = 3072 matrices in flight) instead
of one bucket per 64-row size class: a bucket of 3176 matrices runs 104 of them in a second round of their own.
    python tools/round_buckets.py [span]      (one bucket per size class against `size_buckets(quantum=12 x CUs, span=)`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

span = int(sys.argv[1]) if len(sys.argv) > 1 else 2
orig = batch.size_buckets

def run(label):
    solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
    solver.step(); torch.cuda.synchronize(); solver.adopt_launch_hints()
    solver.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        solver.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    rec = []; solver.step(record=rec); torch.cuda.synchronize()
    tot = {}
    for name, e0, e1 in rec:
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
    counts = [bk["count"] for bk in solver.buckets]
    print(f"{label}: {len(counts)} buckets {counts}: step {ms:.2f} ms; " + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()), flush=True)
    return solver.u.clone(), solver.N.clone()

sizes, tensors = bench.cube_workload(65536, 0, device="cuda:0")
batch.size_buckets = lambda packed, max_slab_bytes=64 << 30, granularity=64, quantum=0, span=2: orig(packed, max_slab_bytes, granularity)
ref = run("one bucket per size class")
batch.size_buckets = lambda packed, max_slab_bytes=64 << 30, granularity=64, quantum=0, span=2: orig(packed, max_slab_bytes, granularity, quantum, span_arg)
span_arg = span
got = run(f"whole rounds, span {span}")
print("bitwise equal:", bool(torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])))
