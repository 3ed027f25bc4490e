#!/usr/bin/env python3
"""Where the assembly kernel's time goes, without an instrument inside it: builds that END the kernel behind phase k
(ONLY=assemble tools/build_variants.sh "asmstop0:-DTRS_ASM_STOP_AFTER=0" ... "asmstop5:-DTRS_ASM_STOP_AFTER=5") timed
bucket by bucket of the 65 536-truss cube batch and on bar-942 x 4096: the differences are the phases.
    python tools/asm_phases.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi, batch

names = ("stage inputs", "geometry + degrees", "scan", "adjacency fill + tile mask", "rank sort + diagonal blocks",
         "envelope metadata", "row loop")
libs = []
for k in range(6):
    lib = ctypes.CDLL(os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_asmstop{k}.so"))
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
    libs.append(lib)
libs.append(_capi.load())


def phases(db, label):
    db.dofmap()
    cum = []
    for lib in libs:
        keep, db.lib = db.lib, lib
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); db.assemble(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        db.lib = keep
        cum.append(float(np.median(ts[1:])))
    db.assemble()   # (the workspace holds the product's output again)
    parts = [cum[0]] + [cum[k] - cum[k - 1] for k in range(1, 7)]
    print(f"{label}: {cum[6]:.3f} ms = " + ", ".join(f"{n} {p:.3f}" for n, p in zip(names, parts)), flush=True)
    return parts


sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=1)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
solver.step(); torch.cuda.synchronize()
total = np.zeros(7)
for bk in solver.buckets:
    db = bk["dev"]
    if not db.small:
        total += phases(db, f"bucket {bk['count']:5d} x {db.rows:4d} rows (nJ <= {db.nJ_max}, nM <= {db.nM_max})")
print(f"all staged buckets: {total.sum():.2f} ms = " + ", ".join(f"{n} {p:.2f}" for n, p in zip(names, total)))
del solver
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    bar = batch.pack_json([json.load(fh)]).replicate(4096)
phases(batch.DeviceBatch(bar, reorder="profile"), "bar-942 x 4096")
