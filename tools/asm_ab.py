#!/usr/bin/env python3
"""The assembly kernel of the product library against a variant build (default `asmold`), bucket by bucket of the
65 536-truss cube batch and on bar-942 x 4096: time per launch, and the slab, load vectors and envelope metadata
compared bit for bit.      python tools/asm_ab.py [variant]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi, batch

tag = sys.argv[1] if len(sys.argv) > 1 else "asmold"
other = ctypes.CDLL(os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so"))
for name, (restype, argtypes) in _capi.SIGNATURES.items():
    fn = getattr(other, name); fn.restype, fn.argtypes = restype, argtypes
product = _capi.load()


def compare(db, label):
    db.dofmap()
    out = {}
    for which, lib in (("product", product), (tag, other)):
        keep, db.lib = db.lib, lib
        db.S.fill_(float("nan")); db.uf.fill_(float("nan")); db.env.zero_()
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); db.assemble(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        db.lib = keep
        out[which] = (float(np.median(ts[1:])), db.S.clone(), db.uf.clone(), db.env.clone())
    a, b = out["product"], out[tag]
    same = bool(torch.equal(torch.nan_to_num(a[1], nan=-7.0), torch.nan_to_num(b[1], nan=-7.0)) and
                torch.equal(torch.nan_to_num(a[2], nan=-7.0), torch.nan_to_num(b[2], nan=-7.0)) and torch.equal(a[3], b[3]))
    print(f"{label}: product {a[0]:.3f} ms, {tag} {b[0]:.3f} ms ({a[0] / b[0]:.3f} x), slab / vectors / metadata bitwise equal: {same}", flush=True)
    return a[0], b[0], same


sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=1)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
solver.step(); torch.cuda.synchronize()
tot, ok = np.zeros(2), True
for bk in solver.buckets:
    db = bk["dev"]
    if not db.small:
        a, b, same = compare(db, f"bucket {bk['count']:5d} x {db.rows:4d} rows")
        tot += (a, b); ok = ok and same
print(f"all staged buckets: product {tot[0]:.2f} ms, {tag} {tot[1]:.2f} ms; bitwise equal everywhere: {ok}")
del solver
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    bar = batch.pack_json([json.load(fh)]).replicate(4096)
compare(batch.DeviceBatch(bar, reorder="profile"), "bar-942 x 4096")
