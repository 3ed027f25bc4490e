#!/usr/bin/env python3
"""Device joint order (trs_joint_order) against the host version (trs_profile_order + trs_apply_joint_order):
time per batch, agreement of the permutations, and the end-to-end effect on solve_batch.

    python tools/order_speed.py [--cubes 65536]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json
import numpy as np
import torch
from python_stable_3d_truss_analysis_amd import batch, generate as gen

ap = argparse.ArgumentParser()
ap.add_argument("--cubes", type=int, default=65536)
args = ap.parse_args()


def device_time(packed, effort, reps=5):
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tensors = {f: up(getattr(packed, f)) for f in ("xyz", "conn", "cbits", "loads", "nJ", "nM")}
    out = batch.joint_order_device(torch, tensors, effort=effort)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = batch.joint_order_device(torch, tensors, effort=effort); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out


rng = np.random.default_rng(0)
cubes = gen.generate_cube_batch(rng.integers(8, 191, size=args.cubes), gridRange=(6, 6, 6), seed=7)
with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    bar = batch.pack_json([json.load(fh)]).replicate(4096)
for name, packed in (("bar-942 x 4096", bar), (f"{args.cubes} mixed cube trusses", cubes)):
    for effort in (2, 1, 0):
        ms, out = device_time(packed, effort)
        t0 = time.perf_counter()
        perm = batch.profile_permutation(packed, effort=effort)
        t_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        batch.permute_joints(packed, perm)
        t_apply = time.perf_counter() - t0
        same = bool((out["perm"].cpu().numpy() == perm).all())
        print(f"{name}, effort {effort}: device {ms:.3f} ms ({packed.B / ms * 1e-3:.2f} M trusses/s) | host order "
              f"{t_host * 1e3:.1f} ms + apply {t_apply * 1e3:.1f} ms | same permutation: {same} | "
              f"max reach {int(out['reach'].max())}")
pinned, pool = cubes.pinned(), batch.ResultPool()
for reorder in ("profile", "host-profile", "host-fast", False):
    for attempt in range(3):
        t0 = time.perf_counter()
        res = batch.solve_batch(pinned, reorder=reorder, pool=pool)
        dt = time.perf_counter() - t0
    print(f"solve_batch(pinned inputs, result pool, reorder={reorder!r}): {dt:.3f} s = {cubes.B / dt:.0f} solves/s, "
          f"info_nonzero={int((res.info != 0).sum())}")
