#!/usr/bin/env python3
"""Experiment: the 4096-truss step split into k sub-batches on k HIP streams (does the chip fill one
kernel's tail and latency-bound phases with another sub-batch's work?) against the one-batch step.

    python tools/stream_overlap.py [B=4096] [k ...]
"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from python_stable_3d_truss_analysis_amd import batch

data = json.load(open("tests/golden/data/bar-942_input_0.json"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
one = batch.pack_json([data])
STAGES = ("dofmap", "assemble", "potrf", "potrs", "recover")
steps = 20


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best


for k in ks:
    devs = [batch.DeviceBatch(one.replicate(B // k)) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]

    def batch_major():
        for d, s in zip(devs, streams):
            with torch.cuda.stream(s):
                for st in STAGES:
                    getattr(d, st)()

    def stage_major():
        for st in STAGES:
            for d, s in zip(devs, streams):
                with torch.cuda.stream(s):
                    getattr(d, st)()

    def skewed():   # sub-batch i runs stage j in slot i + j: neighbouring streams are one stage apart
        for slot in range(k + len(STAGES) - 1):
            for i, (d, s) in enumerate(zip(devs, streams)):
                j = slot - i
                if 0 <= j < len(STAGES):
                    with torch.cuda.stream(s):
                        getattr(d, STAGES[j])()

    n = B // k * k
    line = f"k={k}:"
    for name, fn in (("batch-major", batch_major), ("stage-major", stage_major), ("skewed", skewed)):
        t = timed(fn)
        line += f"  {name} {t * 1e3:.3f} ms = {n / t / 1e6:.2f} M/s"
    print(line, flush=True)
    del devs
    torch.cuda.empty_cache()
