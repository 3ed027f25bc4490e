#!/usr/bin/env python3
"""Experiment for DESIGN section 8 item 7 (the K / L round trip through HBM): the bar-942 x 4096 step cut into
SLICES of trusses that go through assemble -> factor (+ substitute) -> recover one after the other on one stream,

  --shared     every slice uses the SAME slab region (K and L of a slice overwrite the previous slice's: the live
               footprint is one slice - 256 trusses = 0.17 GB of slab address space, 83 MB of touched tiles, which
               the 256 MB memory-side cache could hold),
  (default)    every slice has its own region of the full slab (footprint as the unsliced step).

Prints ms per 4096 trusses for each slice size; run under `rocprofv3 --pmc FETCH_SIZE ...` / `WRITE_SIZE` with
--only SIZE --steps N to read the HBM traffic of one variant (tools/slice_pmc.sh).

    python tools/slice_pipeline.py [--sizes 4096 2048 1024 512 256] [--shared] [--only SIZE --steps N]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from python_stable_3d_truss_analysis_amd import batch

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", type=int, nargs="+", default=[4096, 2048, 1024, 512, 256])
ap.add_argument("--shared", action="store_true")
ap.add_argument("--only", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=4096)
args = ap.parse_args()

with open(os.path.join(ROOT, "tests", "golden", "data", "bar-942_input_0.json")) as fh:
    packed = batch.pack_json([json.load(fh)]).replicate(args.batch)
full = batch.DeviceBatch(packed, reorder="profile")
full.solve(); torch.cuda.synchronize()
ref_u = full.u[0].clone()


def sliced(size, shared):
    subs = []
    S, uf, work, env = full._workspace()
    for lo in range(0, args.batch, size):
        hi = min(args.batch, lo + size)
        t = {f: getattr(full, f)[lo:hi] for f in batch.DeviceBatch.INPUT_FIELDS}
        sub = batch.DeviceBatch.from_device(t, full.n_max, joint_out=full.joint_out[lo:hi], all_narrow=full.all_narrow)
        a, b = (0, hi - lo) if shared else (lo, hi)
        sub._slab = (S[a:b], uf[a:b], work[a:b], env[a:b])
        sub.u, sub.f_ext, sub.N, sub.info = full.u[lo:hi], full.f_ext[lo:hi], full.N[lo:hi], full.info[lo:hi]
        sub.free_index, sub.n_free = full.free_index[lo:hi], full.n_free[lo:hi]
        subs.append(sub)
    return subs


def run(subs, steps):
    for s in subs:
        s.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for s in subs:
            s.solve()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


sizes = [args.only] if args.only else args.sizes
for size in sizes:
    subs = sliced(size, args.shared)
    full.u.fill_(float("nan"))
    ms = run(subs, args.steps)
    err = float((full.u[-1] - ref_u).abs().max() / ref_u.abs().max())
    print(f"slice {size:5d} x {len(subs):3d} ({'shared slab region' if args.shared else 'own slab regions'}): "
          f"{ms:.3f} ms per {args.batch} trusses = {args.batch / ms * 1e-3:.2f} M solves/s, last truss vs unsliced: {err:.1e}")
