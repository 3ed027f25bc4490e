#!/usr/bin/env python3
"""Fused small-system kernel vs the staged pipeline: GPU time of one GA generation (1024 x bar-120,
BASELINE config 4), of 4096 x bar-25 / cube-7, and the latency of a drop-in Truss.Solve()."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from python_stable_3d_truss_analysis_amd import Truss, batch


def gpu_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
for name, B in (("bar-120_input_0", 1024), ("bar-25_input_0", 4096), ("cube-7_case_1", 4096), ("bar-6_input_0", 4096)):
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "data", name + ".json")))
    packed = batch.pack_json([data]).replicate(B)
    small, staged = batch.DeviceBatch(packed), batch.DeviceBatch(packed, use_small=False)
    assert small.small and not staged.small
    rec = {"B": B, "n_free": int(packed.n_free[0]), "small_ms": gpu_ms(small.solve), "staged_ms": gpu_ms(staged.solve),
           "small_with_fitness_ms": gpu_ms(lambda: small.solve_fitness(3e4, 10.0)),
           "staged_with_fitness_ms": gpu_ms(lambda: staged.solve_fitness(3e4, 10.0))}
    a, b = small.result(), staged.result()
    rec["max_rel_diff_u"] = float(np.abs(a.displace - b.displace).max() / np.abs(b.displace).max())
    out[f"{name} x {B}"] = rec
for name in ("bar-25_input_0", "bar-120_input_0"):
    t = Truss(3).LoadFromJSON(data=json.load(open(os.path.join(ROOT, "tests", "golden", "data", name + ".json"))))
    t.Solve()
    t0 = time.perf_counter()
    for _ in range(50):
        t.Solve()
    out[f"{name} Truss.Solve() latency ms"] = (time.perf_counter() - t0) / 50 * 1e3
print(json.dumps(out, indent=1))
