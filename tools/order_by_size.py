#!/usr/bin/env python3
"""trs_joint_order on batches of one size class each (cube trusses of a cube-count range, padded to the class's
own maxima - what a bucket of the ragged solver sees): time per launch and per truss, by effort."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from python_stable_3d_truss_analysis_amd import _capi
if len(sys.argv) > 1:   # a variant build (tools/build_variants.sh) instead of the product library
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{sys.argv[1]}.so")
from python_stable_3d_truss_analysis_amd import batch, generate as gen

rng = np.random.default_rng(0)
for lo, hi, count in ((8, 20, 8192), (20, 60, 8192), (60, 110, 8192), (110, 150, 8192), (150, 190, 8192)):
    packed = gen.generate_cube_batch(rng.integers(lo, hi + 1, size=count), gridRange=(6, 6, 6), seed=lo).trimmed()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tensors = {f: up(getattr(packed, f)) for f in ("xyz", "conn", "cbits", "loads", "nJ", "nM")}
    line = f"{lo:3d}..{hi:3d} cubes x {count} (nJ_max {packed.nJ_max}, nM_max {packed.nM_max}):"
    for effort, apply in ((3, True), (2, True), (1, True), (0, True)):
        out = batch.joint_order_device(torch, tensors, effort=effort, apply=apply); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); batch.joint_order_device(torch, tensors, effort=effort, apply=apply, out=out); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        line += f"  effort {effort}{'' if apply else ' (no apply)'} {best:.3f} ms"
    print(line)
