import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
from python_stable_3d_truss_analysis_amd import batch
sizes, tensors = bench.cube_workload(65536, 0, device="cuda:0")
for reorder in (True, "fast", "rcm"):
    s = batch.RaggedSolver(sizes, reorder=reorder, tensors=tensors)
    s.step(); torch.cuda.synchronize(); s.adopt_launch_hints()
    s.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): s.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    rec = []
    s.step(record=rec); torch.cuda.synchronize()
    st = {}
    for n, a, b in rec: st[n] = st.get(n, 0) + a.elapsed_time(b)
    print(reorder, f"{dt*1e3:.2f} ms/step", {k: round(v, 2) for k, v in st.items()})
    del s
