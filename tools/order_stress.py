#!/usr/bin/env python3
"""Determinism stress of trs_joint_order: the same batch ordered again and again on several streams at once (and beside
a copy kernel), every permutation compared with the first."""
import os, sys, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch as gpu, generate as gen
rng = np.random.default_rng(8)
effort = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for lo, hi, count in ((150, 190, 304), (1, 190, 1500), (60, 100, 400)):
    packed = gen.generate_cube_batch(rng.integers(lo, hi + 1, size=count), gridRange=(6, 6, 6), seed=13).trimmed()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tensors = {f: up(getattr(packed, f)) for f in ("xyz", "conn", "cbits", "loads", "nJ", "nM")}
    ref = gpu.joint_order_device(torch, tensors, effort=effort); torch.cuda.synchronize()
    want = gpu.profile_permutation(packed, effort=effort)
    print(lo, hi, "device == host:", bool((ref["perm"].cpu().numpy() == want).all()), flush=True)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [gpu.joint_order_device(torch, tensors, effort=effort) for _ in streams]
    bad = 0
    for rep in range(60):
        for s, o in zip(streams, outs):
            with torch.cuda.stream(s):
                gpu.joint_order_device(torch, tensors, effort=effort, out=o)
        torch.cuda.synchronize()
        for o in outs:
            bad += int((o["perm"] != ref["perm"]).any().item()) + int((o["xyz"] != ref["xyz"]).any().item())
    print(lo, hi, "mismatching launches:", bad, flush=True)
