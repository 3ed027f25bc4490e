#!/usr/bin/env python3
"""The joint-order kernel (`trs_joint_order_rows`: search + renumbered bucket inputs) of the product library against a
variant build (default `ordold`), bucket by bucket of the 65 536-truss cube batch and on bar-942 x 4096 (`trs_joint_order`):
time per launch, every output compared bit for bit.      python tools/order_ab.py [variant]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi, batch

tag = sys.argv[1] if len(sys.argv) > 1 else "ordold"
other = ctypes.CDLL(os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so"))
for name, (restype, argtypes) in _capi.SIGNATURES.items():
    fn = getattr(other, name); fn.restype, fn.argtypes = restype, argtypes
product = _capi.load()


def compare(run, outs, label):
    res = {}
    for which, lib in (("product", product), (tag, other), ("product again", product)):
        for o in outs:
            o.zero_()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(lib); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[which] = (float(np.median(ts[2:])), [o.clone() for o in outs])
    a, b, c = res["product"], res[tag], res["product again"]
    same = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a[1], b[1]))
    print(f"{label}: product {a[0]:.4f} / {c[0]:.4f} ms, {tag} {b[0]:.4f} ms, outputs bitwise equal: {same}", flush=True)
    return min(a[0], c[0]), b[0], same


sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=1)
solver.step(); torch.cuda.synchronize()
inp = solver.inputs
tot, ok = np.zeros(2), True
for bk in solver.buckets:
    db = bk["dev"]
    if db.small or not bk["fused_io"]:
        continue
    ordr = bk["ordered"]

    def run(lib, db=db, bk=bk, ordr=ordr):
        _capi.check(lib.trs_joint_order_rows(
            bk["count"], db.nJ_max, db.nM_max, bk["rows"].data_ptr(), int(inp["xyz"].shape[1]), int(inp["conn"].shape[1]),
            inp["xyz"].data_ptr(), inp["conn"].data_ptr(), inp["cbits"].data_ptr(), inp["loads"].data_ptr(),
            inp["E"].data_ptr(), inp["A"].data_ptr(), inp["nJ"].data_ptr(), inp["nM"].data_ptr(), ordr["perm"].data_ptr(),
            ordr["reach"].data_ptr(), db.xyz.data_ptr(), db.conn.data_ptr(), db.cbits.data_ptr(), db.loads.data_ptr(),
            db.E.data_ptr(), db.A.data_ptr(), db.nJ.data_ptr(), db.nM.data_ptr(), 3,
            torch.cuda.current_stream().cuda_stream), "trs_joint_order_rows")
    a, b, same = compare(run, [ordr["perm"], ordr["reach"], db.xyz, db.conn, db.cbits, db.loads, db.E, db.A, db.nJ, db.nM],
                         f"bucket {bk['count']:5d} x {db.rows:4d} rows")
    tot += (a, b); ok = ok and same
print(f"all staged buckets: product {tot[0]:.3f} ms, {tag} {tot[1]:.3f} ms; bitwise equal everywhere: {ok}")
del solver
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    bar = batch.pack_json([json.load(fh)]).replicate(4096)
raw = {f: torch.from_numpy(np.ascontiguousarray(getattr(bar, f))).cuda() for f in batch.DeviceBatch.INPUT_FIELDS}
ordered = batch.joint_order_device(torch, raw, effort=3)


def run_bar(lib):
    keep, _capi._lib = _capi._lib, lib
    try:
        batch.joint_order_device(torch, raw, effort=3, out=ordered)
    finally:
        _capi._lib = keep
compare(run_bar, [ordered[k] for k in ("perm", "reach", "xyz", "conn", "cbits", "loads")], "bar-942 x 4096")
