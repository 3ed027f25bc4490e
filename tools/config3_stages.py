#!/usr/bin/env python3
"""Where the time of the mixed cube batch (config 3) goes: per size bucket the stage times (events on the
launch stream), the routing (wave-per-matrix / work-group factorisation), the stored tiles and the MFMA
work of the factorisation inside the envelopes (sampled per bucket with bench.potrf_tile_flops).

    python tools/config3_stages.py [--cubes 65536] [--bucket 64] [--sample 24]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from python_stable_3d_truss_analysis_amd import _capi, batch, generate as gen
if os.environ.get("TRS_LIB_VARIANT"):  # A/B builds of tools/build_variants.sh
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants",
                                  f"libtrs_{os.environ['TRS_LIB_VARIANT']}.so")
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--cubes", type=int, default=65536)
ap.add_argument("--bucket", type=int, default=64)
ap.add_argument("--sample", type=int, default=24)
ap.add_argument("--order", default="profile", choices=("profile", "rcm", "none"), help="joint order (batch.joint_order)")
args = ap.parse_args()
rng = np.random.default_rng(0)
packed = gen.generate_cube_batch(rng.integers(8, 191, size=args.cubes), gridRange=(6, 6, 6), seed=7)
if args.order != "none":
    packed = batch.permute_joints(packed, batch.joint_order(packed, args.order))
groups = batch.size_buckets(packed, 48 << 30, args.bucket)
STAGES = ("dofmap", "assemble", "potrf", "potrs", "recover")
tot = {s: 0.0 for s in STAGES}
tot_flops = tot_tiles = 0.0
n_wide = 0
print(f"{'n_pad':>6} {'B':>6} {'wide':>5} {'tiles':>7} {'MFLOP':>7} | " + " ".join(f"{s:>8}" for s in STAGES[1:]) + " |  potrf TF/s  potrs TB/s")
for g in groups:
    sub = packed.take(g).trimmed()
    dev = batch.DeviceBatch(sub)
    dev.solve(); torch.cuda.synchronize()
    best = {s: 1e9 for s in STAGES}
    for _ in range(3):
        evs = []
        for s in STAGES:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); getattr(dev, s)(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for s, (e0, e1) in zip(STAGES, evs):
            best[s] = min(best[s], e0.elapsed_time(e1))
    env = dev.env.cpu().numpy()
    nchm, npan = dev.rows // 16, dev.rows // 64
    slack = env[:, nchm + npan]
    narrow = (slack & 0xff) == 1
    cend = env[:, nchm + npan + 8: nchm + npan + 8 + nchm]
    nfree = sub.n_free
    tiles = np.zeros(len(g))
    for b in range(len(g)):
        nch = (int(nfree[b]) + 63) // 64 * 4
        tiles[b] = (cend[b, :nch] - np.arange(nch)).sum() + (0 if narrow[b] else nch)
    pick = rng.choice(len(g), size=min(args.sample, len(g)), replace=False)
    fl = np.mean([bench.potrf_tile_flops(int(nfree[b]), env[b, :nchm], env[b, nchm:nchm + npan], cend[b], bool(narrow[b]))
                  for b in pick])
    for s in STAGES:
        tot[s] += best[s]
    tot_flops += fl * len(g)
    tot_tiles += tiles.sum()
    n_wide += int((~narrow).sum())
    print(f"{dev.rows:6d} {len(g):6d} {int((~narrow).sum()):5d} {tiles.mean():7.0f} {fl / 1e6:7.1f} | "
          + " ".join(f"{best[s]:8.3f}" for s in STAGES[1:])
          + f" | {fl * len(g) / best['potrf'] / 1e9:8.1f}   {tiles.sum() * 2048 / best['potrs'] / 1e9:8.2f}", flush=True)
    del dev
    torch.cuda.empty_cache()
t = sum(tot.values())
print(json.dumps({"B": args.cubes, "buckets": len(groups), "wide": n_wide, "ms": {k: round(v, 3) for k, v in tot.items()},
                  "total_ms": round(t, 3), "solves_per_s": args.cubes / t * 1e3,
                  "potrf_tflops": tot_flops / tot["potrf"] / 1e9, "tiles_per_truss": tot_tiles / args.cubes,
                  "mflop_per_truss": tot_flops / args.cubes / 1e6}))
