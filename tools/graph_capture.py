#!/usr/bin/env python3
"""Experiment: the six launches of trs_solve captured in a HIP graph (torch.cuda.CUDAGraph on the
stream the C ABI launches on) against eager launches, for a GA-sized and for the headline batch."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from python_stable_3d_truss_analysis_amd import batch

for name, B in (("bar-120_input_0", 1024), ("bar-25_input_0", 1024), ("bar-942_input_0", 4096)):
    data = json.load(open(f"tests/golden/data/{name}.json"))
    dev = batch.DeviceBatch(batch.pack_json([data]).replicate(B))
    dev.solve(); torch.cuda.synchronize()
    ref = dev.u.clone()
    def timed(fn, n=200):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    eager = timed(dev.solve)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dev.solve()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dev.solve()
    dev.u.zero_()
    graph = timed(g.replay)
    same = bool(torch.equal(dev.u, ref))
    print(f"{name} x {B}: eager {eager:.1f} us/step, graph {graph:.1f} us/step, identical {same}")
