#!/usr/bin/env python3
"""The host-fed call (`batch.solve_batch_streamed`: page-locked host batch in -> page-locked results out) of the 65 536
cube trusses by member form and number of run lanes, alternated in one process; results compared bit for bit.
    python tools/hostfed_lanes.py [rounds]        CONFIGS="general:1 table:1 table:2 table:3" (form:lanes)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# form:lanes[:pull_cus,push_cus]  (the CU sets of the masked copy streams, TRS_PCIE_CUS; default 16,8)
configs = [(c.split(":")[0], int(c.split(":")[1]), (c.split(":") + ["16,8"])[2]) for c in os.environ.get("CONFIGS", "general:1 table:1 table:2 table:3").split()]
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
host = sizes.to_packed(tensors)
del tensors
torch.cuda.empty_cache()
forms = {"general": host.pinned()}
t0 = time.perf_counter()
table = host.table()
print(f"PackedBatch.table(): {time.perf_counter() - t0:.1f} s on the host for {int(host.nM.astype(np.int64).sum())} members", flush=True)
forms["table"] = table.pinned()
pool = batch.ResultPool(tracked=True)
ref, times = None, {c: [] for c in configs}
for r in range(rounds + 1):
    for cfg in configs:
        form, lanes, cus = cfg
        os.environ["TRS_PCIE_CUS"] = cus
        t0 = time.perf_counter()
        res = batch.solve_batch_streamed(forms[form], "cuda:0", reorder=True, pool=pool, lanes=lanes)
        dt = time.perf_counter() - t0
        if r:   # (round 0 warms: page-locked result arrays, workspaces, masked streams)
            times[cfg].append(dt * 1e3)
        if ref is None:
            ref = (np.array(res.displace), np.array(res.internal), np.array(res.external))
        assert np.array_equal(res.displace, ref[0]) and np.array_equal(res.internal, ref[1]) and np.array_equal(res.external, ref[2]), cfg
        assert not res.info.any()
for cfg in configs:
    t = times[cfg]
    print(f"{cfg[0]:8s} {cfg[1]} run lane(s), copy CUs {cfg[2]:5s}: " + " ".join(f"{v:.1f}" for v in t) + f"   median {np.median(t):.1f} ms per call = "
          f"{sizes.B / np.median(t) / 1e3:.3f} M solves/s", flush=True)
print("all calls bitwise equal")
