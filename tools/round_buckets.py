#!/usr/bin/env python3
"""Buckets cut at whole ROUNDS of the factorisation kernel (3 waves x 4 SIMDs x 256 CUs = 3072 matrices in flight) instead
of one bucket per 64-row size class: a bucket of 3176 matrices runs 104 of them in a second round of their own.
    python tools/round_buckets.py [quantum] [span]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

quantum = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
span = int(sys.argv[2]) if len(sys.argv) > 2 else 2
orig = batch.size_buckets

def rounds(packed, max_slab_bytes=64 << 30, granularity=64):
    base = orig(packed, max_slab_bytes, granularity)
    n_pad = (packed.n_free.astype(np.int64) + 63) // 64 * 64
    small = [g for g in base if int(packed.n_free[g].max()) <= batch.SMALL_N and len(np.unique(n_pad[g])) > 1]
    taken = np.zeros(packed.B, dtype=bool)
    for g in small:
        taken[g] = True
    rest = np.flatnonzero(~taken)
    rest = rest[np.argsort(-n_pad[rest], kind="stable")]
    groups, i = list(small), 0
    while i < len(rest):
        top = int(n_pad[rest[i]])
        cap = max(1, max_slab_bytes // (top * (top + 16) * 8))
        in_span = int(np.searchsorted(-n_pad[rest[i:]], -(top - 64 * span), side="right"))
        take = min(cap, in_span)
        if take >= quantum:
            take = take // quantum * quantum
        groups.append(np.sort(rest[i:i + take]))
        i += take
    return groups

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
ref = run("size classes")
batch.size_buckets = rounds
got = run(f"whole rounds of {quantum}, span {span}")
print("bitwise equal:", bool(torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])))
