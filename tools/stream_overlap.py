#!/usr/bin/env python3
"""Experiment: two resident batches solved on two HIP streams at once vs one after the other
(does the chip overlap one batch's HBM-bound stages with the other's instruction-bound ones?)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from python_stable_3d_truss_analysis_amd import batch

data = json.load(open("tests/golden/data/bar-942_input_0.json"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
packed = batch.pack_json([data]).replicate(B)
devs = [batch.DeviceBatch(packed), batch.DeviceBatch(packed)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for d in devs:
    d.solve()
torch.cuda.synchronize()
steps = 10
t0 = time.perf_counter()
for _ in range(steps):
    for d in devs:
        d.solve()
torch.cuda.synchronize()
serial = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(steps):
    for d, s in zip(devs, streams):
        with torch.cuda.stream(s):
            d.solve()
torch.cuda.synchronize()
overlap = time.perf_counter() - t0
n = 2 * steps * B
print(f"one stream: {n / serial:.0f} solves/s   two streams: {n / overlap:.0f} solves/s   ratio {serial / overlap:.3f}")
