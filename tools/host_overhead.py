#!/usr/bin/env python3
"""Where the host time of the small-truss callers goes: cProfile of a drop-in Truss.Solve() (bar-25,
bar-120) and of one GA generation evaluation (1024 x bar-120), plus the bare kernel time of one truss."""
import cProfile
import json
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from python_stable_3d_truss_analysis_amd import MemberType, Truss, batch
from python_stable_3d_truss_analysis_amd.ga import GA


def load(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", "data", name + ".json")))


for name in ("bar-25_input_0", "bar-120_input_0"):
    t = Truss(3).LoadFromJSON(data=load(name))
    for _ in range(20):
        t.Solve()
    t0 = time.perf_counter()
    for _ in range(200):
        t.Solve()
    print(name, "Solve() latency ms", round((time.perf_counter() - t0) / 200 * 1e3, 4))
    packed = batch.pack_trusses([t])
    t0 = time.perf_counter()
    for _ in range(200):
        batch.pack_trusses([t])
    print("   pack_trusses ms", round((time.perf_counter() - t0) / 200 * 1e3, 4))
    t0 = time.perf_counter()
    for _ in range(200):
        batch.solve_batch(packed)
    print("   solve_batch(packed) ms", round((time.perf_counter() - t0) / 200 * 1e3, 4))
    dev = batch.DeviceBatch(packed)
    dev.solve(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dev.solve()
    e1.record(); torch.cuda.synchronize()
    print("   kernel (B=1, back to back) ms", round(e0.elapsed_time(e1) / 200, 4))
    cProfile.run("for _ in range(100): t.Solve()", "/tmp/prof")
    pstats.Stats("/tmp/prof").sort_stats("cumtime").print_stats(14)

truss = Truss(3).LoadFromJSON(data=load("bar-120_input_0"))
random.seed(0)
types = [MemberType(i, random.uniform(1e7, 3e7), random.uniform(0.1, 1.0)) for i in range(1, 21)]
ga = GA(truss, types, nPop=1024, nElite=256)
pop = ga.Initialize()
ga.GetFitnessBatch(pop)
t0 = time.perf_counter()
for _ in range(20):
    ga.GetFitnessBatch(pop)
print("GA generation evaluation ms", round((time.perf_counter() - t0) / 20 * 1e3, 3))
cProfile.run("for _ in range(10): ga.GetFitnessBatch(pop)", "/tmp/prof2")
pstats.Stats("/tmp/prof2").sort_stats("cumtime").print_stats(14)
