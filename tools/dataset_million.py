import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from python_stable_3d_truss_analysis_amd import MemberType, TaskType
from python_stable_3d_truss_analysis_amd import data as gdata
kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
          taskType=TaskType.REGRESSION, device="cuda:0", forceScale=1e3, displaceScale=0.1, positionScale=100.)
for _ in gdata.dataset_chunks(2048, chunk=2048, **kw): pass
torch.cuda.synchronize(); t0 = time.perf_counter()
n = bad = 0
first_pass = []
tc = time.perf_counter()
for first, meta, t in gdata.dataset_chunks(1_000_000, chunk=16384, **kw):
    n += meta.B; bad += int(t["info"].ne(0).sum().item())
    t1 = time.perf_counter(); first_pass.append((t1 - tc) * 1e3); tc = t1
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("first pass, ms per chunk:", " ".join(f"{x:.0f}" for x in first_pass))
print(f"config 5 at full size on ONE GPU: {n} samples (two solves each) in {dt:.2f} s = {n / dt / 1e3:.0f} K samples/s, info_nonzero {bad}")
# per-chunk wall times (allocator behaviour over many chunks of slightly different padded shapes)
times = []
t0 = time.perf_counter()
for first, meta, t in gdata.dataset_chunks(400_000, chunk=16384, **kw):
    bad += int(t["info"].ne(0).sum().item())
    t1 = time.perf_counter(); times.append((t1 - t0) * 1e3); t0 = t1
print("ms per chunk:", " ".join(f"{x:.0f}" for x in times))
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated GB", torch.cuda.memory_allocated() / 1e9)
