#!/usr/bin/env python3
"""Where the assembly of the big cube-truss buckets spends its time: the bucket's assembly with the product
library and with knock-out builds (tools/build_variants.sh "nophase1:-DTRS_EXP_ASM_NOPHASE1"
"nowalk:-DTRS_EXP_ASM_NOWALK"; wrong results, same instruction stream otherwise)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

def load(tag):
    path = _capi.LIB_PATH if tag == "default" else os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
    return lib

sizes, tensors = bench.cube_workload(65536, 0, device="cuda:0")
# (one lane and the 48 GiB budget of round 4 by default, so that the bucket set - and the numbers - compare with R4.1)
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=int(os.environ.get("LANES", 1)),
                            max_slab_bytes=int(os.environ.get("SLAB_GB", 48)) << 30)
solver.step(); torch.cuda.synchronize()
libs = {t: load(t) for t in ["default"] + (sys.argv[1:] or ["nophase1", "nowalk"])}
total = {t: 0.0 for t in libs}
for bk in solver.buckets:
    db = bk["dev"]
    if db.small:
        continue
    # this bucket's gathered + ordered inputs persist in db; the shared workspace is free to reuse
    line = f"bucket {bk['count']:5d} x {db.rows:4d} (nM_max {db.nM_max}):"
    for tag, lib in libs.items():
        db.lib = lib
        db.dofmap(); db.assemble(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            db.assemble()
        e1.record(); torch.cuda.synchronize()
        line += f"  {tag} {e0.elapsed_time(e1) / 3:.3f} ms"
        total[tag] += e0.elapsed_time(e1) / 3
    db.lib = libs["default"]
    print(line)
print("sum over the buckets (ms): " + "  ".join(f"{t} {v:.2f}" for t, v in total.items()))
