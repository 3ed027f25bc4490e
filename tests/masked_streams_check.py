"""Run by tests/test_gpu_order.py::test_masked_streams_run_kernels_and_refuse_bad_masks in a process of its own."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from python_stable_3d_truss_analysis_amd import batch as gpu  # noqa: E402

lib = gpu._capi.load()
try:
    streams = gpu._MaskedStreams(torch, torch.device("cuda:0"), lib, 8, 4)
except gpu.HipExtensionError as exc:       # (the pipeline then runs on ordinary streams, `_pipeline_streams`)
    print(f"refused: this runtime refuses CU-masked streams: {exc}")
    sys.exit(0)
P, Z = ctypes.c_void_p, ctypes.c_size_t
src = torch.arange(64 * 100, dtype=torch.float64, device="cuda").reshape(64, 100)
rows = torch.arange(63, -1, -1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
outs = []
for st in streams:
    dst = torch.zeros_like(src)
    assert lib.trs_copy_rows(1, (P * 1)(src.data_ptr()), (Z * 1)(800), (P * 1)(dst.data_ptr()), (Z * 1)(800), (Z * 1)(800),
                             None, None, None, None, 64, rows.data_ptr(), 0, 2, st.cuda_stream) == 0
    outs.append(dst)
torch.cuda.synchronize()
assert all(torch.equal(dst, src.flip(0)) for dst in outs)
streams.close()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for pull in (n_cu, 0):
    try:
        gpu._MaskedStreams(torch, torch.device("cuda:0"), lib, pull, 8)
    except ValueError:
        continue
    raise AssertionError(f"a mask set of {pull} + 8 CUs was not refused")
print("masked streams ok")
