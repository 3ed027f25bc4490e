#!/usr/bin/env python3
"""Latency of one drop-in `Truss.Solve()` call (pack, upload, five kernels, download) per bundled case."""
import sys, json, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import Truss
for name in ("bar-25_input_0", "bar-120_input_0", "bar-942_input_0"):
    data = json.load(open(f"tests/golden/data/{name}.json"))
    t = Truss(3).LoadFromJSON(data=data)
    t.Solve()
    t0 = time.perf_counter()
    for _ in range(20):
        t.Solve()
    dt = (time.perf_counter() - t0) / 20
    print(name, "Solve() latency ms", round(dt * 1e3, 3))
import cProfile, pstats
cProfile.run("t.Solve()", "/tmp/prof")
pstats.Stats("/tmp/prof").sort_stats("cumtime").print_stats(18)
