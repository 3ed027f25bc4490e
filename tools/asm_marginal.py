#!/usr/bin/env python3
"""What a stage costs the ragged cube STEP under the four lanes, by launching it TWICE (same inputs, same outputs: the
results do not change): the increase of the step is what removing the stage altogether could save at most - the price
tag for any speed-up of it.      python tools/asm_marginal.py [rounds] [steps]"""
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
plain_rows = batch.DeviceBatch.solve_rows


def twice(stage):
    def solve_rows(self, rows, out, nJ_out_max, nM_out_max, types=None):
        if stage in ("assemble", "dofmap+assemble"):
            self.dofmap(); self.assemble()
        if stage == "recover":   # (after the real call: uf holds the solution)
            plain_rows(self, rows, out, nJ_out_max, nM_out_max, types=types)
            self.recover_rows(rows, out, nJ_out_max, nM_out_max)
            return
        plain_rows(self, rows, out, nJ_out_max, nM_out_max, types=types)
    return solve_rows


configs = ["as shipped", "assemble", "recover"]
ref, times = None, {c: [] for c in configs}
for r in range(rounds):
    for cfg in configs:
        batch.DeviceBatch.solve_rows = plain_rows if cfg == "as shipped" else twice(cfg)
        solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        torch.cuda.synchronize()
        times[cfg].append((time.perf_counter() - t0) / steps * 1e3)
        if ref is None:
            ref = (solver.u.clone(), solver.N.clone())
        assert torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]), cfg
        del solver
        batch.release_workspaces()
batch.DeviceBatch.solve_rows = plain_rows
base = float(np.median(times["as shipped"]))
for cfg in configs:
    t = times[cfg]
    extra = "" if cfg == "as shipped" else f"   = +{np.median(t) - base:.2f} ms for a second launch of the stage per bucket"
    print(f"{cfg:12s} " + " ".join(f"{v:.2f}" for v in t) + f"   median {np.median(t):.2f} ms per step" + extra, flush=True)
print("results bitwise equal")
