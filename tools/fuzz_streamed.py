#!/usr/bin/env python3
"""One-off consistency fuzz: `solve_batch_streamed` (host-fed pipeline) and the resident bucket pipeline with section
variants against the staged `solve_batch`, bit for bit, over random batch sizes (1 .. 1000), size mixes (small-only,
mixed, large-only) and order plans."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch, generate as gen
if os.environ.get("LANES"):
    batch.DEFAULT_LANES = int(os.environ["LANES"])
ONLY = [int(v) for v in os.environ.get("ONLY_TRIALS", "").split()]   # (replay: only these trials are solved, the draws stay)
rng = np.random.default_rng(0)
bad = 0
pool = batch.ResultPool()
TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
BIG = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for trial in range(TRIALS):
    B = int(rng.choice([1, 2, 5, 17, 64, 300, 1000, BIG]))
    lo, hi = [(1, 3), (1, 30), (20, 60), (1, 190), (150, 190)][trial % 5]
    packed = gen.generate_cube_batch(rng.integers(lo, hi + 1, size=B), gridRange=(6, 6, 6), seed=int(rng.integers(1 << 30)))
    reorder = [True, False, "rcm", "fast"][trial % 4]
    if ONLY and trial not in ONLY:
        continue
    ref = batch.solve_batch(packed, reorder=reorder)
    got = batch.solve_batch_streamed(packed.pinned(), reorder=reorder, pool=pool if trial % 3 else None)
    ok = all(np.array_equal(getattr(got, k), getattr(ref, k)) for k in ("displace", "external", "internal", "info"))
    # resident solver with variants against sections
    t = {f: torch.from_numpy(np.ascontiguousarray(getattr(packed, f))).cuda() for f in batch.DeviceBatch.INPUT_FIELDS}
    sec = [None, (2.0, 2e7, 0.3)]
    want = batch.solve_batch(packed, reorder=reorder, sections=sec)
    res = batch.solve_batch(packed, reorder=reorder, sections=sec, device_inputs=t, on_device=True)
    ok2 = all(np.array_equal(r.displace.cpu().numpy(), w.displace) and np.array_equal(r.internal.cpu().numpy(), w.internal)
              and np.array_equal(r.external.cpu().numpy(), w.external) for r, w in zip(res, want))
    # the table member form (ABI 10; several member types so that the table has more than one row): resident bucket
    # pipeline with the same section variants and the host-fed pipeline, against the general form's staged results
    multi = gen.generate_cube_batch(rng.integers(lo, hi + 1, size=min(B, 200)), gridRange=(6, 6, 6), seed=int(rng.integers(1 << 30)),
                                    memberTypes=((1., 1e7, 0.1), (2.5, 2e7, 0.2), (0.75, 3e7, 0.4)))
    tab = multi.table()
    want_t = batch.solve_batch(multi, reorder=reorder, sections=sec)
    got_t = batch.solve_batch(tab, reorder=reorder, sections=sec)
    fed_t = batch.solve_batch_streamed(tab.pinned(), reorder=reorder)
    ok3 = len(tab.types) == len({tuple(r) for r in np.stack([multi.A, multi.E, multi.rho], -1)[np.arange(multi.nM_max)[None, :] < multi.nM[:, None]]}) and \
        all(np.array_equal(getattr(g, k), getattr(w, k)) for g, w in zip(got_t, want_t) for k in ("displace", "external", "internal", "info")) and \
        all(np.array_equal(getattr(fed_t, k), getattr(want_t[0], k)) for k in ("displace", "external", "internal", "info"))
    if not ok3:
        bad += 1
        print("MISMATCH (table member form) trial", trial, B, (lo, hi), reorder)
    if not (ok and ok2):
        bad += 1
        print("MISMATCH trial", trial, B, (lo, hi), reorder, ok, ok2)
        for v, (r, w) in enumerate(zip(res, want)):
            du = np.abs(r.displace.cpu().numpy() - w.displace).reshape(B, -1).max(1)
            dn = np.abs(r.internal.cpu().numpy() - w.internal).reshape(B, -1).max(1)
            rows = np.flatnonzero((du > 0) | (dn > 0) | ~np.isfinite(du))
            print(f"   variant {v}: {len(rows)} trusses differ, rows {rows[:8].tolist()}, n_free {packed.n_free[rows[:8]].tolist()}, "
                  f"max |du| {np.nanmax(du) if len(du) else 0:.3e} of {np.abs(w.displace).max():.3e}, info {r.info.cpu().numpy()[rows[:8]].tolist()}")
        again = batch.solve_batch(packed, reorder=reorder, sections=sec, device_inputs=t, on_device=True)
        print("   the same call again equals the staged results:", all(
            np.array_equal(r.displace.cpu().numpy(), w.displace) and np.array_equal(r.internal.cpu().numpy(), w.internal)
            for r, w in zip(again, want)))
print("trials done, mismatches:", bad)
