#!/usr/bin/env python3
"""Multi-lane steps compared with the one-lane results, truss by truss: how many differ, by how much, in which buckets
and lanes (EXPERIMENTS R4.9: one of three bench runs saw different bits).   python tools/lanes_diff.py [lanes] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi
if os.environ.get("VARIANT"):   # a variant build (tools/build_variants.sh) instead of the product library
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{os.environ['VARIANT']}.so")
from python_stable_3d_truss_analysis_amd import batch

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
one = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=1)
one.step(); torch.cuda.synchronize()
ref_u, ref_N, ref_f = one.u.clone(), one.N.clone(), one.f_ext.clone()
nJ_max = int(one.u.shape[1])
ref_perm = torch.full([sizes.B, nJ_max], -1, dtype=torch.int32, device="cuda")   # the one-lane run's joint order per truss
for bk in one.buckets:
    if bk["dev"].joint_out is not None:
        w = bk["dev"].joint_out.shape[1]
        ref_perm[bk["rows"], :w] = bk["dev"].joint_out
del one
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=lanes)
owner = np.zeros(sizes.B, dtype=np.int64)
for k, bk in enumerate(solver.buckets):
    owner[bk["idx"]] = k
bad_steps = 0
for it in range(steps):
    solver.u.fill_(float("nan")); solver.N.fill_(float("nan"))
    solver.step(); torch.cuda.synchronize()
    du = (torch.nan_to_num(solver.u, nan=1e300) != ref_u).flatten(1).any(1)
    dn = (torch.nan_to_num(solver.N, nan=1e300) != ref_N).flatten(1).any(1)
    # (rows beyond a truss's own joints / members were NaN-filled here and are zero in the reference: mask by counts)
    nJ, nM = tensors["nJ"].long(), tensors["nM"].long()
    mj = torch.arange(solver.u.shape[1], device="cuda")[None, :] < nJ[:, None]
    mm = torch.arange(solver.N.shape[1], device="cuda")[None, :] < nM[:, None]
    du = ((solver.u != ref_u).any(2) & mj).any(1)
    dn = ((solver.N != ref_N) & mm).any(1)
    wrong = torch.nonzero(du | dn).flatten().cpu().numpy()
    if len(wrong):
        bad_steps += 1
        rel = float(((solver.u - ref_u).abs() * mj[..., None]).nan_to_num(nan=1e300).amax() / ref_u.abs().amax())
        ks = sorted(set(owner[wrong].tolist()))
        for g in wrong[:4]:      # is the truss's joint order still a permutation, and the one-lane run's?
            bk = solver.buckets[int(owner[g])]
            row = int(np.flatnonzero(bk["idx"] == g)[0]) if not isinstance(bk["idx"], torch.Tensor) else None
            rows_host = bk["rows"].cpu().numpy()
            row = int(np.flatnonzero(rows_host == g)[0])
            if bk["dev"].joint_out is not None:
                nj = int(tensors["nJ"][g])
                perm = bk["dev"].joint_out[row, :nj].cpu().numpy()
                refp = ref_perm[g, :nj].cpu().numpy()
                print(f"   truss {g}: nJ {nj}, info {int(solver.info[g])}, order is a permutation: {sorted(perm.tolist()) == list(range(nj))}, "
                      f"equals the one-lane order: {bool((perm == refp).all())}", flush=True)
        print(f"step {it}: {len(wrong)} trusses differ (max |du| / max |u| = {rel:.2e}); buckets "
              + ", ".join(f"{k} (lane {solver.buckets[k]['lane']}, {solver.buckets[k]['count']} x {solver.buckets[k]['dev'].rows}: "
                          f"{int((owner[wrong] == k).sum())})" for k in ks), flush=True)
print(f"{lanes} lanes, {len(solver.buckets)} buckets, {steps} steps: {bad_steps} with different results")
