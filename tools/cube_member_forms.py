#!/usr/bin/env python3
"""The resident ragged cube step in the general and in the table member form (ABI 10), alternated in one process,
results compared bit for bit.      python tools/cube_member_forms.py [rounds] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
table = {k: tensors[k] for k in ("xyz", "cbits", "loads", "nJ", "nM")}
table["conn"] = tensors["conn"].to(torch.uint16)
table["type_idx"] = torch.zeros(tensors["A"].shape, dtype=torch.uint8, device="cuda:0")     # the generator's one member type
table["types"] = torch.tensor([[1.0, 1e7, 0.1]], dtype=torch.float64, device="cuda:0")
assert bool((tensors["A"][tensors["A"] != 0] == 1.0).all())
forms = {"general": tensors, "table": table}
ref, times = None, {k: [] for k in forms}
for r in range(rounds):
    for name, tens in forms.items():
        solver = batch.RaggedSolver(sizes, reorder=True, tensors=tens)
        assert solver.table == (name == "table")
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / steps * 1e3)
        if ref is None:
            ref = (solver.u.clone(), solver.N.clone(), solver.f_ext.clone())
        assert torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]) and torch.equal(solver.f_ext, ref[2]), name
        del solver
        batch.release_workspaces()
for name in forms:
    t = times[name]
    print(f"{name:8s} " + " ".join(f"{v:.2f}" for v in t) + f"   median {np.median(t):.2f} ms per step", flush=True)
print("results bitwise equal")
