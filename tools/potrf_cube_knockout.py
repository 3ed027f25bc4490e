#!/usr/bin/env python3
"""Where the factorisation of the cube-truss buckets spends its time: every bucket's `trs_potrf_batched` with the
product library and with knock-out builds of potrf.hip (wrong results, same instruction stream otherwise):

    ONLY=potrf tools/build_variants.sh "ign:-DTRS_EXP_IGNORE_PIVOT" "nifb:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_IFB" \\
        "ndl:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_DLOADS" "nkl:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_KLOADS" \\
        "niu:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_ITEMUPDATE" "nch:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_CHOL16" \\
        "nis:-DTRS_EXP_IGNORE_PIVOT -DTRS_EXP_NO_ITEMSTORE"
    python tools/potrf_cube_knockout.py [tags ...]

nifb = items without their block-side fragment loads (what staging the panel's rows on chip could save at most),
ndl = block update without its re-reads, nkl = no stiffness-tile loads, niu = items without their update loop,
nch = no 16 x 16 factorisations, nis = items' result tiles not stored."""
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

tags = ["default"] + (sys.argv[1:] or ["ign", "nifb", "ndl", "nkl", "niu", "nch", "nis"])
sizes, tensors = bench.cube_workload(65536, 0, device="cuda:0")
# (one lane and the 48 GiB budget of round 4 by default, so that the bucket set - and the numbers - compare with R4.1)
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=int(os.environ.get("LANES", 1)),
                            max_slab_bytes=int(os.environ.get("SLAB_GB", 48)) << 30)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
libs = {t: load(t[4:] if t.startswith("asm-") else t) for t in tags}
total = {t: 0.0 for t in tags}
for bk in solver.buckets:
    db = bk["dev"]
    if db.small:
        continue
    line = f"bucket {bk['count']:5d} x {db.rows:4d}:"
    for tag, lib in libs.items():
        ms = 0.0
        for rep in range(3):
            # ("asm-<tag>": the variant assembles - e.g. nokm: every envelope tile written - and the product factors)
            db.lib = lib if tag.startswith("asm-") else libs["default"]
            db.dofmap(); db.assemble()
            db.lib = libs["default"] if tag.startswith("asm-") else lib
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); db.potrf(); e1.record(); torch.cuda.synchronize()
            if rep:
                ms += e0.elapsed_time(e1) / 2
        total[tag] += ms
        line += f"  {tag} {ms:.3f}"
        if hasattr(lib, "trs_debug_stamps"):   # a -DTRS_POTRF_STAMPS build: per-phase wave-cycle shares
            buf = (ctypes.c_ulonglong * 8)()
            lib.trs_debug_stamps(buf, 1)
            tot = float(sum(buf)) or 1.0
            names = ("tile_loads", "block_update", "factorisation", "load_column+stores", "items", "fence", "substitution")
            line += " [" + ", ".join(f"{n} {buf[i] / tot:.3f}" for i, n in enumerate(names)) + "]"
    db.lib = libs["default"]
    print(line)
print("sum over the buckets (ms): " + "  ".join(f"{t} {v:.2f}" for t, v in total.items()))
