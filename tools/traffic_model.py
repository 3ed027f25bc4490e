#!/usr/bin/env python3
"""Tile traffic of the factorisation of the ragged cube-truss batch under several kernel organisations, from the
envelopes alone (host only: native generator + native profile order; no GPU).  Units: 16 x 16 FP64 tiles (2 KB).

    python tools/traffic_model.py [--cubes 4096]

Organisations priced (every one stores / reads the same envelope tiles; they differ in the RE-READS of factor rows):
  narrow      wave per matrix, 64-column panels, two-chunk items, nothing cached (upper bound of today's kernel)
  narrow_b0   the same with every item's block-side rows served on chip (= panel rows read once per panel)
  p128_b0     128-column panels, block-side rows on chip
  window      every tile read once and written once (the whole active window on chip)
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from python_stable_3d_truss_analysis_amd import batch, generate as gen
from python_stable_3d_truss_analysis_amd.data import dataset_sizes


def envelope(conn, cbits, nJ, nM):
    """ft (non-decreasing row envelope, tiles), cend (stored extents) and n_pad of one truss in the given numbering."""
    free = np.zeros([nJ, 3], dtype=bool)
    for a in range(3):
        free[:, a] = (cbits[:nJ] >> a) & 1 == 0
    idx = np.cumsum(free.ravel()) - 1
    n = int(free.sum())
    npad = (n + 63) // 64 * 64
    nch = npad // 16
    first = idx.reshape(nJ, 3)                     # reduced index of each DOF (valid where free)
    jlo = np.where(free, first, 1 << 30).min(axis=1)   # lowest reduced row of the joint
    jhi = np.where(free, first, -1).max(axis=1)
    ft = np.arange(nch)                             # diagonal tile at least
    c = conn[:nM]
    for a, b in ((c[:, 0], c[:, 1]), (c[:, 1], c[:, 0])):
        ok = (jhi[a] >= 0) & (jhi[b] >= 0)
        # rows of joint a (chunks of its dofs) couple to columns of joint b (lowest column -> tile)
        for dof in range(3):
            r = first[a, dof]
            okk = ok & free[a, dof]
            np.minimum.at(ft, r[okk] // 16, jlo[b][okk] // 16)
    # joints' own 3x3 blocks
    okj = jhi >= 0
    for dof in range(3):
        okk = okj & free[:, dof]
        np.minimum.at(ft, first[okk, dof] // 16, jlo[okk] // 16)
    ft = np.minimum.accumulate(ft[::-1])[::-1]
    lastc = np.zeros(nch, dtype=np.int64)
    for t in range(nch):
        lastc[t] = np.nonzero(ft <= t)[0].max()
    cend = np.maximum(lastc + 1, 4 * (np.arange(nch) // 4) + 4)
    return ft, cend, nch


def price(ft, cend, nch, panel_tiles=4, item=2, b_on_chip=False):
    """Tile reads of factor rows by the left-looking update (diagonal block + items), for panels of `panel_tiles`."""
    P = panel_tiles
    rd_b = rd_a = 0
    for j in range(0, nch, P):
        blk = list(range(j, min(j + P, nch)))
        kd = ft[j]
        rd_b += sum(max(0, j - max(kd, ft[u])) for u in blk)   # block update: the panel's own rows, once
        lastq = max(int(np.nonzero(ft <= blk[-1])[0].max()), blk[-1])
        c0 = blk[-1] + 1
        while c0 <= lastq:
            nv = min(item, lastq - c0 + 1)
            if not b_on_chip:
                rd_b += len(blk) * max(0, j - ft[c0])
            rd_a += sum(max(0, j - max(ft[c0], ft[c0 + v])) for v in range(nv))
            c0 += nv
    return rd_a, rd_b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cubes", type=int, default=2048)
    ap.add_argument("--scale-to", type=int, default=65536)
    args = ap.parse_args()
    sizes = dataset_sizes(7, 0, args.cubes, (8, 190))
    packed = gen.generate_cube_batch(sizes, gridRange=(6, 6, 6), seed=7)
    perm = batch.profile_permutation(packed)
    pk = batch.permute_joints(packed, perm)
    tot = {}
    reach_hist = {}
    front_hist = {}
    for b in range(pk.B):
        ft, cend, nch = envelope(pk.conn[b], pk.cbits[b], int(pk.nJ[b]), int(pk.nM[b]))
        stored = int((cend - np.arange(nch)).sum())
        front = int((cend - np.arange(nch)).max())
        front_hist[front] = front_hist.get(front, 0) + 1
        window = int(max((cend[t] - t) * (cend[t] - t + 1) // 2 for t in range(nch)))
        rows = {"stored": (stored, 0, 0)}
        rows["narrow"] = (stored,) + price(ft, cend, nch, 4, 2, False)
        rows["narrow_i4"] = (stored,) + price(ft, cend, nch, 4, 4, False)
        rows["narrow_b0"] = (stored,) + price(ft, cend, nch, 4, 2, True)
        rows["p128_b0"] = (stored,) + price(ft, cend, nch, 8, 2, True)
        rows["p256_b0"] = (stored,) + price(ft, cend, nch, 16, 2, True)
        for k, v in rows.items():
            t = tot.setdefault(k, [0, 0, 0])
            for i in range(3):
                t[i] += v[i]
        t = tot.setdefault("window_tiles_max", [0, 0, 0])
        t[0] = max(t[0], window); t[1] += window
    scale = args.scale_to / pk.B * 2048 / 1e9
    st = tot["stored"][0]
    print(f"{pk.B} trusses, scaled to {args.scale_to}: stored tiles {st * scale:.1f} GB; algorithmic (K read, L written, L read by "
          f"the substitution) {3 * st * scale:.1f} GB")
    for k in ("narrow", "narrow_i4", "narrow_b0", "p128_b0", "p256_b0"):
        s, a, bb = tot[k]
        total = 3 * s + a + bb
        print(f"  {k:10s}: item-side re-reads {a * scale:6.1f} GB, block-side {bb * scale:6.1f} GB -> total "
              f"{total * scale:6.1f} GB = {total / (3 * st):.2f} x algorithmic")
    print(f"  window    : {3 * st * scale:6.1f} GB = 1.00 x; largest window {tot['window_tiles_max'][0]} tiles "
          f"({tot['window_tiles_max'][0] * 2} KB), mean of per-truss maxima {tot['window_tiles_max'][1] / pk.B:.0f} tiles")
    print("  widest stored column (tiles) histogram:", dict(sorted(front_hist.items())))


if __name__ == "__main__":
    main()
