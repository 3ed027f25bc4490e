#!/usr/bin/env python3
"""Where the joint-order kernel's time goes: wave-cycle shares per phase from a -DTRS_ORDER_STAMPS build
(tools/build_variants.sh "ordst:-DTRS_ORDER_STAMPS"), on the mixed cube batch and on bar-942 x 4096."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from python_stable_3d_truss_analysis_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", "libtrs_ordst.so")
from python_stable_3d_truss_analysis_amd import batch, generate as gen

lib = _capi.load()
names = ("A0 adjacency", "A1 Cuthill-McKee", "A2 bins", "B sort", "B price x2", "B rcm/cm price + tail", "C wait", "C perm+reach", "C apply")
rng = np.random.default_rng(0)
cubes = gen.generate_cube_batch(rng.integers(8, 191, size=16384), gridRange=(6, 6, 6), seed=7)
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    bar = batch.pack_json([json.load(fh)]).replicate(4096)
big = gen.generate_cube_batch(rng.integers(150, 191, size=8192), gridRange=(6, 6, 6), seed=9)   # the largest buckets' trusses
small = gen.generate_cube_batch(rng.integers(8, 40, size=16384), gridRange=(6, 6, 6), seed=10).trimmed()
for label, packed in (("16384 cubes", cubes), ("8192 cubes of 150..190", big), ("16384 cubes of 8..39", small), ("bar-942 x 4096", bar)):
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tensors = {f: up(getattr(packed, f)) for f in ("xyz", "conn", "cbits", "loads", "nJ", "nM")}
    for effort in (3, 2, 0):
        batch.joint_order_device(torch, tensors, effort=effort); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.trs_order_debug_stamps(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); batch.joint_order_device(torch, tensors, effort=effort); e1.record(); torch.cuda.synchronize()
        lib.trs_order_debug_stamps(buf, 1)
        tot = float(sum(buf)) or 1.0
        print(f"{label}, effort {effort}: {e0.elapsed_time(e1):.3f} ms | " + ", ".join(f"{n} {buf[i] / tot:.3f}" for i, n in enumerate(names)))
