#!/usr/bin/env python3
"""Does the cube buckets' factorisation need every CU?  `trs_potrf_batched` and `trs_assemble` of the largest buckets
on CU-masked streams (consecutive mask bits go round the XCDs), alone and - on disjoint CU sets - side by side:
what an assembly of bucket b + 1 beside the factorisation of bucket b could buy."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

lib = _capi.load()
dev = torch.device("cuda:0")
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
words = (n_cu + 31) // 32

def masked(cus):
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    h = ctypes.c_void_p()
    _capi.check(lib.trs_stream_create_masked(mask, words, ctypes.byref(h)), "trs_stream_create_masked")
    return h, torch.cuda.ExternalStream(h.value, device=dev)

sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
big = [bk["dev"] for bk in solver.buckets if not bk["dev"].small][:2]   # the two largest (they have their own inputs;
a, b2 = big[0], big[1]                                                   # the workspace is shared: give b2 its own)
b2._slab = None
b2._workspace()

def timed(stream, call, reps=3):
    best = 1e9
    for _ in range(reps):
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); call(); e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

for db in (a, b2):
    db.dofmap(); db.assemble()
torch.cuda.synchronize()
print(f"{n_cu} CUs; bucket A {a.B} x {a.rows}, bucket B {b2.B} x {b2.rows}")
handles = []
for n in (n_cu, 224, 192, 160, 128):
    h, s = masked(range(0, n)); handles.append(h)
    a.assemble(); torch.cuda.synchronize()
    t_f = timed(s, a.potrf)
    t_a = timed(s, b2.assemble)
    print(f"  {n:3d} CUs: potrf A {t_f:.3f} ms, assemble B {t_a:.3f} ms")
for n in (224, 192, 160):
    h1, s1 = masked(range(0, n)); h2, s2 = masked(range(n, n_cu)); handles += [h1, h2]
    best = 1e9
    for _ in range(3):
        a.assemble(); torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        with torch.cuda.stream(s1):
            a.potrf(); e1.record(s1)
        with torch.cuda.stream(s2):
            b2.assemble(); e2.record(s2)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(e1), e0.elapsed_time(e2)))
        t1, t2 = e0.elapsed_time(e1), e0.elapsed_time(e2)
    print(f"  side by side, potrf A on {n} CUs ({t1:.3f} ms) + assemble B on {n_cu - n} ({t2:.3f} ms): {best:.3f} ms")
for h in handles:
    lib.trs_stream_destroy(h)
