import os, sys, faulthandler
faulthandler.dump_traceback_later(50, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import _capi
mode = sys.argv[1]
if len(sys.argv) > 2:
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{sys.argv[2]}.so")
from python_stable_3d_truss_analysis_amd import batch as gpu, generate as gen
rng = np.random.default_rng(8)
packed = gen.generate_cube_batch(rng.integers(1, 191, size=1500), gridRange=(6, 6, 6), seed=13)
pinned = packed.pinned()
pool = gpu.ResultPool(tracked=(mode == "tracked"))
reorder = {"profile": "profile", "rcm": "rcm", "none": False}.get(mode, True)
if os.environ.get("PLAIN_STREAMS"):
    dev = torch.device("cuda:0")
    gpu._PIPELINE_STREAMS[str(dev)] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev))
print("mode", mode, "reorder", reorder, flush=True)
want = gpu.solve_batch(packed, reorder=reorder)
if os.environ.get("FUSED_FIRST"):   # the resident fused path first (the sweeps-only order instance runs), as in bench.py
    rs = gpu.RaggedSolver(packed, reorder=reorder)
    rs.step(); torch.cuda.synchronize()
    print("resident fused step ok, all_large buckets:", sum(bk["all_large"] for bk in rs.buckets), flush=True)
for rep in range(3):
    got = gpu.solve_batch_streamed(pinned, reorder=reorder, pool=pool)
    print(rep, "equal", bool(np.array_equal(got.displace, want.displace) and np.array_equal(got.internal, want.internal)), flush=True)
