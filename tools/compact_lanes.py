#!/usr/bin/env python3
"""the compact form (K_ff never in HBM as tiles) under the four lanes of the ragged cube step - all
buckets, only the buckets from / below a row count, with the factorisation kernels at other occupancies (variant
libraries of tools/build_variants.sh, switched inside ONE process).  Every configuration is measured several times in
alternation; results are compared bit for bit with the first configuration's.
    python tools/compact_lanes.py [rounds] [steps]      CONFIGS="slab compact c>=512 ..."   LANES=4"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from python_stable_3d_truss_analysis_amd import _capi, batch

PRODUCT = _capi.LIB_PATH
VARDIR = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants")


def use_lib(tag):
    path = PRODUCT if not tag else os.path.join(VARDIR, f"libtrs_{tag}.so")
    if _capi.LIB_PATH != path or _capi._lib is None:
        _capi._lib, _capi.LIB_PATH = None, path
        _capi.load()


def parse(cfg):
    """'slab' | 'compact' | 'c>=N' | 'c<N', optionally '@variant' behind it"""
    form, _, tag = cfg.partition("@")
    if form == "slab":
        opts = {}
    elif form == "compact":
        opts = {"compact": True}
    elif form.startswith("c>="):
        opts = {"compact_rows": (int(form[3:]), 1 << 30)}
    elif form.startswith("c<"):
        opts = {"compact_rows": (0, int(form[2:]))}
    else:
        raise ValueError(cfg)
    return opts, tag


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lanes = int(os.environ.get("LANES", batch.DEFAULT_LANES))
configs = os.environ.get("CONFIGS", "slab compact c>=512 c>=704 c<512 slab@w2 compact@fw3 c>=512@fw3").split()
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
ref, times, info, same = None, {c: [] for c in configs}, {}, {}
for r in range(rounds):
    for cfg in configs:
        opts, tag = parse(cfg)
        use_lib(tag)
        solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, lanes=lanes, options=opts)
        solver.step(); torch.cuda.synchronize()
        solver.adopt_launch_hints()
        solver.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        torch.cuda.synchronize()
        times[cfg].append((time.perf_counter() - t0) / steps * 1e3)
        ncomp = sum(1 for bk in solver.buckets if bk["dev"].options["compact"] and not bk["dev"].small)
        if os.environ.get("TILES"):   # stored tiles per truss under this configuration's joint orders (sum of cend[t] - t)
            tiles = 0
            for bk in solver.buckets:
                db = bk["dev"]
                if db.small:
                    continue
                db.dofmap(); db.assemble()
                env = db.env[:bk["count"]].cpu().numpy()
                nchm, npan = db.rows // 16, db.rows // 64
                cend = env[:, nchm + npan + 8: nchm + npan + 8 + nchm].astype(np.int64)
                nch = (sizes.n_free[bk["idx"]].astype(np.int64) + 63) // 64 * 4
                t = np.arange(nchm)[None, :]
                tiles += int(((cend - t) * (t < nch[:, None])).sum())
            info[cfg] = f"{len(solver.buckets)} buckets, {tiles / sizes.B:.1f} stored tiles per truss"
        else:
            info[cfg] = f"{len(solver.buckets)} buckets, {ncomp} compact"
        if ref is None:
            ref = (solver.u.clone(), solver.N.clone(), solver.f_ext.clone())
        same[cfg] = bool(torch.equal(solver.u, ref[0]) and torch.equal(solver.N, ref[1]) and torch.equal(solver.f_ext, ref[2])
                         and not bool(solver.info.any().item()))
        del solver
        batch.release_workspaces()
for cfg in configs:
    t = times[cfg]
    print(f"{cfg:16s} {info[cfg]:24s} " + " ".join(f"{v:.2f}" for v in t) +
          f"   median {np.median(t):.2f} ms   bitwise = first config: {same[cfg]}", flush=True)
