#!/usr/bin/env python3
"""The cube buckets' factorisation + substitution with the product library and with builds that route matrices to the
work-group kernel with the window in LDS (trs_window.h): time per bucket and the solutions compared.

    ONLY="assemble potrf potrs" tools/build_variants.sh "win:-DTRS_EXP_WINDOW -DTRS_EXP_WINDOW_ABOVE=0"
    python tools/window_probe.py win [more tags]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

def load(tag):
    path = _capi.LIB_PATH if tag == "default" else os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
    return lib

count = int(os.environ.get("CUBES", 65536))
tags = ["default"] + sys.argv[1:]
sizes, tensors = bench.cube_workload(count, 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
libs = {t: load(t) for t in tags}
total = {t: 0.0 for t in tags}
for bk in solver.buckets:
    db = bk["dev"]
    if db.small:
        continue
    line = f"bucket {bk['count']:5d} x {db.rows:4d}:"
    ref = None
    for tag, lib in libs.items():
        db.lib = lib
        ms = [0.0, 0.0]
        for rep in range(3):
            db.dofmap(); db.assemble()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(); db.potrf(); ev[1].record(); db.potrs(); ev[2].record(); torch.cuda.synchronize()
            if rep:
                ms[0] += ev[0].elapsed_time(ev[1]) / 2; ms[1] += ev[1].elapsed_time(ev[2]) / 2
        total[tag] += sum(ms)
        line += f"  {tag} {ms[0]:.3f}+{ms[1]:.3f}"
        if hasattr(lib, "trs_debug_stamps"):   # a -DTRS_POTRF_STAMPS build: shares of wave 0's and wave 1's cycles
            buf = (ctypes.c_ulonglong * 8)()
            lib.trs_debug_stamps(buf, 1)
            names = ("A", "A-wait", "ahead", "B-wait", "A", "A-wait", "update", "B-wait")
            for h in (0, 4):
                tot = float(sum(buf[h:h + 4])) or 1.0
                line += " [" + ", ".join(f"{names[h + i]} {buf[h + i] / tot:.2f}" for i in range(4)) + "]"
        uf, info = db.uf.clone(), db.info.clone()
        if ref is None:
            ref = (uf, info)
        else:
            n = db.n_free.long()
            mask = torch.arange(uf.shape[1], device=uf.device)[None, :] < n[:, None]
            scale = (ref[0].abs() * mask).amax(dim=1).clamp_min(1e-300)
            err = (((uf - ref[0]).abs() * mask).amax(dim=1) / scale)
            err = torch.nan_to_num(err, nan=float("inf"))
            routed = int(((db.env[:, db.rows // 16 + db.rows // 64] & 0x800) != 0).sum())
            line += f" (routed {routed}, err {float(err.max()):.1e}, info equal {bool((info == ref[1]).all())})"
    db.lib = libs["default"]
    print(line, flush=True)
print("sum over the buckets (ms): " + "  ".join(f"{t} {v:.2f}" for t, v in total.items()))
