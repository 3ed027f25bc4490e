#!/usr/bin/env python3
"""HBM traffic of the cube batch's factorisation by what causes it: run under rocprofv3 --pmc FETCH_SIZE (and again
with WRITE_SIZE), once per knock-out build (tools/potrf_cube_knockout.py lists them):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/x -- python3 tools/potrf_cube_traffic.py nifb

Every bucket is assembled with the product library and factored ONCE with the named build; these are the LAST
dispatches of trs_potrf_narrow_kernel in the counter file (tools/potrf_cube_traffic_sum.py adds them up)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
path = _capi.LIB_PATH if tag == "default" else os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so")
lib = ctypes.CDLL(path)
for name, (restype, argtypes) in _capi.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
sizes, tensors = bench.cube_workload(65536, 0, device="cuda:0")
# (one lane and the 48 GiB budget of round 4 by default, so that the bucket set - and the numbers - compare with R4.1)
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=int(os.environ.get("LANES", 1)),
                            max_slab_bytes=int(os.environ.get("SLAB_GB", 48)) << 30)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
n = 0
for bk in solver.buckets:
    db = bk["dev"]
    if db.small:
        continue
    db.dofmap(); db.assemble()
    keep = db.lib
    db.lib = lib
    db.potrf()
    db.lib = keep
    torch.cuda.synchronize()
    n += 1
print("measured dispatches:", n)
