#!/usr/bin/env python3
"""StreamedSolver (upload -> solve -> download of bar-942 x 4096 batches over PCIe) with 2 .. 4 slots, and the
same traffic without the solve / without the download, to see which stream bounds the pipeline."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from python_stable_3d_truss_analysis_amd import batch
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    packed = batch.pack_json([json.load(fh)]).replicate(4096)
for slots in (2, 3, 4):
    pipe = batch.StreamedSolver(packed, "cuda:0", slots=slots, same_topology=True)
    src = pipe.host_in[0]
    for _ in range(4):
        pipe.submit(src)
    pipe.drain(); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.submit(src)
    pipe.drain(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nin = sum(v.numel() * v.element_size() for v in src.values())
    nout = sum(v.numel() * v.element_size() for v in pipe.host_out[0].values())
    print(f"slots {slots}: {dt * 1e3:.2f} ms per batch = {4096 / dt / 1e6:.2f} M solves/s, {(nin + nout) / dt / 1e9:.1f} GB/s combined "
          f"(up {nin / 1e6:.0f} MB, down {nout / 1e6:.0f} MB)")
    del pipe
