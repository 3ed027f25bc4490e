"""GPU parity tests: the HIP pipeline (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Run on the GPU box with `-m gpu`.

Tolerance: BASELINE.json's north_star asks for 1e-6 relative on the bundled data cases, measured
with the densified max-scaled comparator of SURVEY.md section 4 (`max|a-b| / max|ref|`).  The FP64
pipeline is far inside that; the stage tests below use tighter per-stage bounds so that a layout
bug cannot hide behind the end-to-end tolerance.
"""
import json

import numpy as np
import pytest

from oracle import truss_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_NORTH_STAR = 1e-6     # BASELINE.json: displacements and forces to 1e-6 relative
TOL_FP64 = 1e-9           # what the FP64 Cholesky path actually holds on these cases (cond <= 6e6)


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from python_stable_3d_truss_analysis_amd import batch
    return batch


def _device_batch(gpu, datas, **kw):
    return gpu.DeviceBatch(gpu.pack_json(datas), **kw)


def _upper_from_slab(S, npad):
    """Dense upper-triangular U (npad x npad) and the rhs column from one slab."""
    U = np.triu(S[:npad, :npad])
    return U, S[:npad, npad].copy()


@pytest.mark.parametrize("compact_mode", [False, True], ids=["slab", "compact"])
@pytest.mark.parametrize("name", ["bar-6_input_0", "bar-25_input_0", "bar-47_input_0", "bar-120_input_0",
                                  "bar-942_input_0", "cube-fixture"])
def test_stages_against_oracle(gpu, name, compact_mode):
    if name == "cube-fixture":   # a generated cube truss in a sweep order: its envelope has tiles without an entry of K_ff
        cases = list(H.ragged_cube_cases())
        data = max((d for _, d, _ in cases), key=lambda d: len(d["member"]) if len(d["member"]) < 900 else 0)
        perm = gpu.profile_permutation(gpu.pack_json([data]))[0][:len(data["joint"])]
        inv = np.argsort(perm)
        data = {"joint": [data["joint"][int(j)] for j in perm],
                "force": [[int(inv[j]), f] for j, f in data["force"]],
                "member": [[[int(inv[a]), int(inv[b])], t] for (a, b), t in data["member"]]}
    else:
        data = H.load_json(name)
    ref = orc.solve(data)
    dev = _device_batch(gpu, [data], options={"compact": compact_mode})   # per-batch switch -> per-call flags
    n = int(dev.packed.n_free[0])
    npad = (n + 63) // 64 * 64

    # --- dofmap -------------------------------------------------------------------------
    dev.dofmap()
    fi = dev.free_index.cpu().numpy()[0]
    dim = orc.truss_dim(data)
    m = ref["mask"].reshape(-1, dim)
    full = np.zeros([len(m), 3], dtype=bool)
    full[:, :dim] = m
    expect = np.where(full.ravel(), np.cumsum(full.ravel()) - 1, -1)
    assert int(dev.n_free.cpu()[0]) == n == int(full.sum())
    np.testing.assert_array_equal(fi[: len(expect)], expect)

    # --- assemble (full symmetric, so the whole K_ff can be compared) -------------------------
    dev.assemble(flags=1)
    S = dev.S.cpu().numpy()[0]
    K = S[:n, :n]
    assert H.max_scaled_err(K, ref["K_ff"]) <= 1e-14, "assembled K_ff differs from the oracle"
    f_free = orc.force_vector(data)[ref["mask"]]
    np.testing.assert_array_equal(S[n:npad, n:npad], np.eye(npad - n))
    narrow_any = (int(dev.env.cpu().numpy()[0][dev.rows // 16 + dev.rows // 64]) & 0xff) == 1
    if narrow_any:   # wave-per-matrix factorisation: the load vector travels in uf, not in the slab
        np.testing.assert_array_equal(dev.uf.cpu().numpy()[0][:n], f_free)
        assert not dev.uf.cpu().numpy()[0][n:npad].any()
    else:            # work-group factorisation: 16-wide load-column chunk behind the matrix
        np.testing.assert_allclose(S[:n, npad], f_free, rtol=0, atol=0)
        assert not S[:npad, npad + 1: npad + 16].any()

    # --- envelope metadata against the non-zero pattern of the oracle's K_ff ----------------------
    nch = npad // 16
    first = np.arange(nch)
    for c in range(n):
        first[c // 16] = min(first[c // 16], int(np.flatnonzero(ref["K_ff"][c])[0]) // 16)
    ft = np.minimum.accumulate(first[::-1])[::-1]
    last = [max(q for q in range(nch) if ft[q] <= 4 * j + 3) for j in range(nch // 4)]
    env = dev.env.cpu().numpy()[0]
    env_ft, env_last = env[:nch], env[dev.rows // 16: dev.rows // 16 + nch // 4]
    # the device envelope is STRUCTURAL (joint coupling), so it may only be wider than the numerical
    # pattern (a member along an axis gives exact zeros inside its 3x3 blocks)
    assert (env_ft <= ft).all() and (env_ft <= np.arange(nch)).all() and (np.diff(env_ft) >= 0).all()
    assert (env_last >= np.array(last)).all()
    assert env_last.tolist() == [max(q for q in range(nch) if env_ft[q] <= 4 * j + 3) for j in range(nch // 4)]
    # stored extent per 16-row chunk (csrc/trs_common.h): exact at tile granularity for the
    # wave-per-matrix kernel (slack 1), rectangular per panel for the work-group kernel (slack 3)
    nchm = dev.rows // 16
    slack = int(env[nchm + dev.rows // 64]) & 0xff
    assert not int(env[nchm + dev.rows // 64]) & 0x100     # TRS_ASM_FULL_SYMMETRIC implies the slab form
    assert not int(env[nchm + dev.rows // 64]) & 0x200     # four-chunk items: off by default (trs_common.h)
    env_cend = env[nchm + dev.rows // 64 + 8: nchm + dev.rows // 64 + 8 + nch]
    lastc = [max(q for q in range(nch) if env_ft[q] <= t) for t in range(nch)]
    if slack == 1:
        assert env_cend.tolist() == [min(nch, max(lastc[t] + 1, (t | 3) + 1)) for t in range(nch)]
    else:
        assert env_cend.tolist() == [min(nch, int(env_last[t // 4]) + 1 + 3) for t in range(nch)]

    # --- assemble (production layout) ------------------------------------------------------------
    dev.S.fill_(float("nan"))
    dev.uf.fill_(float("nan"))
    dev.assemble(flags=0)
    S = dev.S.cpu().numpy()[0]
    meta = dev.env.cpu().numpy()[0][nchm + dev.rows // 64: nchm + dev.rows // 64 + 8]
    # tiles of the envelope that hold no entry of K_ff are not written for the wave-per-matrix kernels (kmask,
    # csrc/trs_common.h: bit d of word t <=> tile (rows of chunk t, columns of chunk t + d) is written)
    kmask = dev.env.cpu().numpy()[0][2 * nchm + dev.rows // 64 + 8: 2 * nchm + dev.rows // 64 + 8 + nch].astype(np.int64) & 0xffffffff
    rows_c, cols_c = np.arange(npad)[:, None] // 16, np.arange(npad)[None, :] // 16
    in_env = (cols_c >= rows_c) & (cols_c < env_cend[rows_c])
    written = in_env & (((kmask[rows_c] >> np.clip(cols_c - rows_c, 0, 63)) & 1) == 1)
    if slack != 1:
        assert (kmask == 0xffffffff).all()          # the work-group kernel's matrices: every stored tile is written
    want_full = np.zeros([npad, npad])
    want_full[:n, :n] = ref["K_ff"]
    want_full[np.arange(n, npad), np.arange(n, npad)] = 1.0           # identity padding
    assert not want_full[in_env & ~written].any()   # a tile that is skipped holds no entry of the oracle's K_ff
    assert written[np.arange(npad), np.arange(npad)].all()
    compact = bool(int(meta[0]) & 0x100)
    # with the option on, narrow envelopes leave the assembly as entry lists
    assert compact == (compact_mode and slack == 1) and int(meta[0]) & 0xff == slack
    if compact:
        # compact form: nothing in the slab; K_ff as per-tile entry lists in `work`, f in uf
        assert np.isnan(S).all()
        work = dev.work.cpu().numpy().reshape(-1)
        off = [int(v) * 16 for v in meta[1:5]]
        tbase = work[off[1]: off[1] + 4 * (nch + 1)].view(np.int32)
        ntile = int(tbase[nch])
        assert tbase.tolist() == np.concatenate([[0], np.cumsum(env_cend - np.arange(nch))]).tolist()
        tdesc = work[off[0]: off[0] + 8 * ntile].view(np.int32).reshape(ntile, 2)
        assert tdesc[0, 0] == 0 and (tdesc[1:, 0] == np.cumsum(tdesc[:-1, 1])).all()   # lists back to back
        total = int(tdesc[-1, 0] + tdesc[-1, 1])
        slots = work[off[2]: off[2] + 2 * total].view(np.uint16).astype(np.int64)
        vals = work[off[3]: off[3] + 8 * total].view(np.float64)
        K = np.zeros([npad, npad])
        seen = np.zeros([npad, npad], dtype=np.int32)
        for t in range(nch):
            for q in range(t, int(env_cend[t])):
                beg, cnt = tdesc[tbase[t] + q - t]
                p = slots[beg: beg + cnt]
                assert ((p >> 8) == q - t).all()                   # (tile - chunk) << 8 | D-form slot
                p = p & 255
                rows = 16 * t + ((p >> 4) & 3) + 4 * (p >> 6)      # slot: r * 64 + lq * 16 + li
                cols = 16 * q + (p & 15)
                K[rows, cols] = vals[beg: beg + cnt]
                np.add.at(seen, (rows, cols), 1)
        assert seen.max() == 1                                       # every entry exactly once
        want = np.zeros([npad, npad])
        want[:n, :n] = ref["K_ff"]
        want[np.arange(n, npad), np.arange(n, npad)] = 1.0           # identity padding
        stored = np.arange(npad)[None, :] >= (np.arange(npad)[:, None] // 16 * 16)
        assert H.max_scaled_err(K[stored], want[stored]) <= 1e-14
        assert not K[~stored].any()
        uf0 = dev.uf.cpu().numpy()[0]
        np.testing.assert_array_equal(uf0[:n], f_free)
        assert not uf0[n:npad].any()
        # bit-identical to the slab form of the same matrix (same sums in the same order)
        dev.options["compact"] = False                               # the slab form of the same matrix
        dev.assemble(flags=0)
        dev.options["compact"] = True
        S2 = dev.S.cpu().numpy()[0]
        inside = stored & (np.arange(npad)[None, :] < 16 * env_cend[np.arange(npad) // 16][:, None])
        assert np.array_equal(S2[:npad, :npad][inside & written], K[inside & written])
        assert not K[inside & ~written].any() and np.isnan(S2[:npad, :npad][inside & ~written]).all()
        dev.S.fill_(float("nan"))
        dev.assemble(flags=0)
    else:
        # every written tile equals the oracle's K_ff; nothing else of the slab was touched (NaN poison)
        assert H.max_scaled_err(S[:npad, :npad][written], want_full[written]) <= 1e-14
        assert np.isnan(S[:npad, :npad][~written]).all()
        assert not want_full[(cols_c >= rows_c) & ~in_env].any()    # beyond the envelope K is structurally zero
        if name == "cube-fixture":
            assert (in_env & ~written).any()                        # (this envelope does contain tiles without an entry)
        if not (in_env & ~written).any():
            assert (kmask == 0xffffffff).all()                      # skipping is all or nothing per matrix

    # --- potrf: U^T U = K_ff, y = L^-1 f ------------------------------------------------------
    # (the stages apart: by default the wave that factors a narrow-envelope matrix substitutes it as well)
    dev.options["fused_substitution"] = False
    try:
        dev.potrf()
    finally:
        dev.options["fused_substitution"] = True
    assert int(dev.info.cpu()[0]) == 0
    S = dev.S.cpu().numpy()[0]
    U, _ = _upper_from_slab(S, npad)
    y = dev.uf.cpu().numpy()[0][:npad]           # y = L^-1 f is left in uf for trs_potrs
    Lref = np.linalg.cholesky(ref["K_ff"])
    # tiles outside the envelope are never written (NaN poison of the test slab): there the factor
    # must be structurally zero
    outside = np.isnan(U[:n, :n])
    assert np.abs(Lref.T[outside]).max(initial=0.0) <= 1e-12 * np.abs(Lref).max()
    U = np.where(np.isnan(U), 0.0, U)
    assert H.max_scaled_err(U[:n, :n], Lref.T) <= 1e-9, "Cholesky factor differs from numpy"
    np.testing.assert_array_equal(np.diag(U)[n:], np.ones(npad - n))
    yref = np.linalg.solve(Lref, f_free)
    assert H.max_scaled_err(y[:n], yref) <= 1e-9
    # the strictly-lower part of every 16 x 16 diagonal tile carries inv(L_tile) for trs_potrs
    for t0 in range(0, n - 15, 16):
        Linv = np.linalg.inv(Lref[t0:t0 + 16, t0:t0 + 16])
        got = np.tril(S[t0:t0 + 16, t0:t0 + 16], -1)
        assert H.max_scaled_err(got, np.tril(Linv, -1)) <= 1e-9

    # --- potrs ----------------------------------------------------------------------------------
    dev.potrs()
    uf = dev.uf.cpu().numpy()[0]
    uref = np.linalg.solve(ref["K_ff"], f_free)
    assert H.max_scaled_err(uf[:n], uref) <= TOL_FP64
    assert not uf[n:npad].any()
    # the default: factorisation and substitution by the same wave - uf holds u after trs_potrf_batched, bit
    # for bit the u of the separate stages, the routing word says so and trs_potrs_batched leaves it alone
    if narrow_any and npad <= 1024:
        u_separate = dev.uf.clone()
        dev.assemble(flags=0)
        dev.potrf()
        dev.torch.cuda.synchronize()
        nchm = dev.rows // 16
        assert int(dev.env.cpu()[0][nchm + dev.rows // 64]) & 0x400
        assert bool((dev.uf.view(dev.torch.int64) == u_separate.view(dev.torch.int64)).all())
        dev.potrs()
        assert bool((dev.uf.view(dev.torch.int64) == u_separate.view(dev.torch.int64)).all())

    # --- recover -------------------------------------------------------------------------------
    dev.recover()
    res = dev.result()
    nJ, nM = len(data["joint"]), len(data["member"])
    assert H.max_scaled_err(res.displace[0, :nJ, :dim], ref["u"]) <= TOL_FP64
    assert H.max_scaled_err(res.external[0, :nJ, :dim], ref["f_ext"]) <= TOL_FP64
    assert H.max_scaled_err(res.internal[0, :nM], ref["N"]) <= TOL_FP64


def test_all_data_cases_one_ragged_batch_vs_reference_goldens(gpu):
    """Every bundled data case (2D and 3D, n from 5 to 696) in ONE ragged batch, against the
    dense vectors captured from the real reference and against its stored output files."""
    names = H.data_case_names()
    datas = [H.load_json(nm) for nm in names]
    res = gpu.solve_batch(gpu.pack_json(datas))
    z = H.dense_golden()
    assert not res.info.any()
    for b, (nm, data) in enumerate(zip(names, datas)):
        dim, nJ, nM = orc.truss_dim(data), len(data["joint"]), len(data["member"])
        stored = H.load_json(nm.replace("_input_", "_output_"))
        for got, key, sparse, width in ((res.displace[b, :nJ, :dim], "u", "displace", dim),
                                        (res.external[b, :nJ, :dim], "f_ext", "external", dim),
                                        (res.internal[b, :nM], "N", "internal", None)):
            err = H.max_scaled_err(got, z[f"{nm}/{key}"])
            assert err <= TOL_NORTH_STAR, (nm, key, err)
            assert err <= TOL_FP64, (nm, key, err)
            count = nM if width is None else nJ
            assert H.max_scaled_err(got, orc.densify(stored[sparse], count, width)) <= TOL_NORTH_STAR
        if dim == 2:  # embedded z axis stays exactly zero
            assert not res.displace[b, :, 2].any() and not res.internal[b, nM:].any()


def test_fused_factorisation_end_to_end(gpu):
    """The opt-in compact form (K_ff as per-tile entry lists, tiles formed inside the factorisation): every
    bundled case and the ragged cube fixtures against the golden vectors, and bit for bit against the slab
    form - the tiles it forms are the tiles the slab form stores."""
    names = H.data_case_names()
    datas = [H.load_json(nm) for nm in names] + [d for _, d, _ in H.ragged_cube_cases()]
    packed = gpu.pack_json(datas)
    z = H.dense_golden()
    golds = [{k: z[f"{nm}/{k}"] for k in ("u", "f_ext", "N")} for nm in names] + \
            [g for _, _, g in H.ragged_cube_cases()]
    res = {}
    for reorder in (False, True):
        fused = gpu.solve_batch(packed, reorder=reorder, options={"compact": True})
        slab = gpu.solve_batch(packed, reorder=reorder)
        assert not fused.info.any()
        for b, (data, gold) in enumerate(zip(datas, golds)):
            dim, nJ, nM = orc.truss_dim(data), len(data["joint"]), len(data["member"])
            assert H.max_scaled_err(fused.displace[b, :nJ, :dim], gold["u"]) <= TOL_FP64, (b, reorder)
            assert H.max_scaled_err(fused.external[b, :nJ, :dim], gold["f_ext"]) <= TOL_FP64, (b, reorder)
            assert H.max_scaled_err(fused.internal[b, :nM], gold["N"]) <= TOL_FP64, (b, reorder)
        np.testing.assert_array_equal(fused.displace, slab.displace)
        np.testing.assert_array_equal(fused.internal, slab.internal)
        np.testing.assert_array_equal(fused.external, slab.external)
    # at BASELINE's batch: bar-942 x 4096, every copy identical to the slab form's
    data = H.load_json("bar-942_input_0")
    big = gpu.pack_json([data]).replicate(4096)
    dev = gpu.DeviceBatch(big, options={"compact": True})
    dev.solve()
    a = dev.result()
    slack = int(dev.env.cpu().numpy()[0][dev.rows // 16 + dev.rows // 64])
    assert slack & 0x100 and (slack & 0xff) == 1
    dev2 = gpu.DeviceBatch(big.take(np.arange(8)))
    dev2.solve()
    b8 = dev2.result()
    assert not a.info.any()
    assert (a.displace == b8.displace[0]).all() and (a.internal == b8.internal[0]).all()


def test_cube7_reference_files(gpu):
    names = H.cube7_case_names()
    datas = [H.load_json(nm) for nm in names]
    res = gpu.solve_batch(gpu.pack_json(datas))
    assert not res.info.any()
    for b, stored in enumerate(datas):
        nJ, nM = len(stored["joint"]), len(stored["member"])
        assert H.max_scaled_err(res.displace[b, :nJ], orc.densify(stored["displace"], nJ, 3)) <= TOL_FP64
        assert H.max_scaled_err(res.external[b, :nJ], orc.densify(stored["external"], nJ, 3)) <= TOL_FP64
        assert H.max_scaled_err(res.internal[b, :nM], orc.densify(stored["internal"], nM)) <= TOL_FP64


def test_ragged_cube_trusses_vs_reference(gpu):
    """GenerateRandomCubeTrusses fixtures, 119 .. 2135 members (n up to ~880), one batch."""
    cases = list(H.ragged_cube_cases())
    res = gpu.solve_batch(gpu.pack_json([d for _, d, _ in cases]))
    assert not res.info.any()
    for b, (name, data, gold) in enumerate(cases):
        nJ, nM = len(data["joint"]), len(data["member"])
        assert H.max_scaled_err(res.displace[b, :nJ], gold["u"]) <= TOL_FP64, name
        assert H.max_scaled_err(res.external[b, :nJ], gold["f_ext"]) <= TOL_FP64, name
        assert H.max_scaled_err(res.internal[b, :nM], gold["N"]) <= TOL_FP64, name


def test_edge_cases(gpu):
    from python_stable_3d_truss_analysis_amd import Truss, TrussNotStableError
    for name, entry in H.edge_cases().items():
        data = entry["input"]
        dim = orc.truss_dim(data)
        truss = Truss(dim).LoadFromJSON(data=data)
        if entry.get("raises") == "TrussNotStableError":
            with pytest.raises(TrussNotStableError):
                truss.Solve()
            continue
        if entry.get("raises") == "LinAlgError":
            with pytest.raises(np.linalg.LinAlgError):
                truss.Solve()
            continue
        truss.Solve()
        nJ, nM = truss.nJoint, truss.nMember
        u = orc.densify([[j, v] for j, v in truss.GetDisplacements().items()], nJ, dim)
        f = orc.densify([[j, v] for j, v in truss.GetExternalForces().items()], nJ, dim)
        n = orc.densify([[m, v] for m, v in truss.GetInternalForces().items()], nM)
        assert H.max_scaled_err(u, entry["u"]) <= TOL_FP64, name
        assert H.max_scaled_err(f, entry["f_ext"]) <= TOL_FP64, name
        assert H.max_scaled_err(n, entry["N"]) <= TOL_FP64, name
        # sparse dict views: same key sets as the genuine reference dicts for these cases
        assert sorted(truss.GetDisplacements()) == [j for j, _ in entry["sparse"]["displace"]], name
        assert sorted(truss.GetInternalForces()) == [m for m, _ in entry["sparse"]["internal"]], name
        resist = truss.GetResistances()
        for j, v in entry["resist"].items():
            assert H.max_scaled_err(resist[int(j)], v) <= 1e-8, name
        assert truss.weight == pytest.approx(entry["weight"], rel=1e-14)


def test_truss_solve_dropin_and_json_roundtrip(gpu, tmp_path):
    """BASELINE config 1 through the drop-in API: LoadFromJSON -> Solve -> DumpIntoJSON."""
    from python_stable_3d_truss_analysis_amd import Truss
    truss = Truss(3).LoadFromJSON(str(H.GOLDEN + "/data/bar-25_input_0.json"))
    assert not truss.isSolved
    truss.Solve()
    assert truss.isSolved
    out = tmp_path / "bar-25_out.json"
    truss.DumpIntoJSON(str(out))
    mine = json.loads(out.read_text())
    stored = H.load_json("bar-25_output_0")
    assert mine["joint"] == stored["joint"] and mine["member"] == stored["member"]
    assert mine["force"] == stored["force"]
    for key, count, width in (("displace", 10, 3), ("external", 10, 3), ("internal", 25, None)):
        assert H.max_scaled_err(orc.densify(mine[key], count, width),
                                orc.densify(stored[key], count, width)) <= TOL_FP64
    assert mine["weight"] == pytest.approx(stored["weight"], rel=1e-14)
    again = truss.Copy()
    assert again.isSolved and sorted(again.GetInternalForces()) == sorted(truss.GetInternalForces())


def test_replicated_batch_is_identical_and_deterministic(gpu):
    """BASELINE config 2 at reduced batch: bar-942 x 64 independent copies.  Every copy must give
    the same answer as copy 0, and the factorisation must be bit-reproducible run to run."""
    data = H.load_json("bar-942_input_0")
    packed = gpu.pack_json([data]).replicate(64)
    dev = gpu.DeviceBatch(packed)
    dev.solve()
    first = dev.result()
    assert not first.info.any()
    z = H.dense_golden()
    assert H.max_scaled_err(first.displace[0], z["bar-942_input_0/u"]) <= TOL_FP64
    # assembly (sorted adjacency), factorisation and substitution all sum in a fixed order:
    # every copy is bit-identical to copy 0
    for b in range(1, 64):
        assert np.array_equal(first.displace[b], first.displace[0])
        assert np.array_equal(first.internal[b], first.internal[0])
    # potrf on identical input is bitwise reproducible (fixed summation order)
    dev.dofmap(); dev.assemble()
    S0, f0 = dev.S.clone(), dev.uf.clone()      # (compact form: the matrix sits in `work`, f in uf)
    dev.potrf()
    U1, y1 = dev.S.clone(), dev.uf.clone()
    dev.S.copy_(S0); dev.uf.copy_(f0)
    dev.potrf()
    assert bool((y1.view(dev.torch.int64) == dev.uf.view(dev.torch.int64)).all())
    same = U1[:, :704, :720].triu().view(dev.torch.int64) == dev.S[:, :704, :720].triu().view(dev.torch.int64)
    assert bool(same.all())   # bitwise (NaN poison outside the envelope compares equal as bits)


def test_property_checks_at_full_batch(gpu):
    """Size-independent properties at BASELINE's batch size (bar-942 x 4096):
    equilibrium residual K u = f at free DOFs via member forces, and sum of reactions = -sum of loads."""
    data = H.load_json("bar-942_input_0")
    packed = gpu.pack_json([data]).replicate(4096)
    rng = np.random.default_rng(7)
    packed.loads *= rng.uniform(0.5, 1.5, size=(4096, 1, 1))      # linearity: scaled loads
    dev = gpu.DeviceBatch(packed)
    dev.solve()
    res = dev.result()
    assert not res.info.any()
    # global equilibrium: external forces (loads + reactions) sum to zero
    total = res.external.sum(axis=1)
    assert np.abs(total).max() <= 1e-7 * np.abs(packed.loads).sum(axis=(1, 2)).max()
    # linearity: u scales with the load factor
    base = dev.packed.loads[:, :, :].reshape(4096, -1)
    k = np.argmax(np.abs(base[0]))
    factor = base[:, k] / base[0, k]
    assert np.abs(res.displace - res.displace[0:1] * factor[:, None, None]).max() \
        <= 1e-9 * np.abs(res.displace).max()


def _strip_truss_json(n_joints, seed, extra_pin=False, dim=2, rollers=0):
    """A triangulated strip (Warren-like) of `n_joints` joints with a pin, a roller and random loads:
    n_free = 2 n_joints - 3 (or - 4 with a second pin) in 2D, so every system size can be hit.
    dim=3: a triangulated TOWER - the joints on a jittered helix, the first three pinned, every further joint tied
    to the three before it (a Henneberg step: rigid), `rollers` (0..2) of the last joints on a one-axis roller:
    n_free = 3 (n_joints - 3) - rollers, i.e. every size from 2 up (1 does not exist in 3D: a joint frees 3, 2 or 0)."""
    rng = np.random.default_rng(seed)
    if dim == 2:
        xy = [[50.0 * i + rng.uniform(-5, 5), (60.0 if i % 2 else 0.0) + rng.uniform(-5, 5)] for i in range(n_joints)]
        joints = [[p, "NO"] for p in xy]
        joints[0][1] = "PIN"
        joints[n_joints - 1 - (n_joints - 1) % 2][1] = "PIN" if extra_pin else "ROLLER_Y"
        members = [[[i, i + 1], [rng.uniform(0.5, 2.0), 1e7, 0.1]] for i in range(n_joints - 1)]
        members += [[[i, i + 2], [rng.uniform(0.5, 2.0), 1e7, 0.1]] for i in range(n_joints - 2)]
        forces = [[j, [float(rng.uniform(-1e3, 1e3)), float(rng.uniform(-1e3, 1e3))]]
                  for j in range(1, n_joints - 1) if joints[j][1] == "NO" and rng.random() < 0.7]
        return {"joint": joints, "force": forces, "member": members}
    assert dim == 3 and n_joints >= 4 and 0 <= rollers <= min(2, n_joints - 4 + 1) and (rollers < 2 or n_joints >= 5)
    joints = [[[float(100.0 * np.cos(2.1 * i) + rng.uniform(-5, 5)), float(100.0 * np.sin(2.1 * i) + rng.uniform(-5, 5)),
                float(25.0 * i + rng.uniform(-3, 3))], "NO"] for i in range(n_joints)]
    for j in range(3):
        joints[j][1] = "PIN"
    for k in range(rollers):
        joints[n_joints - 1 - k][1] = ("ROLLER_Z", "ROLLER_X")[k]
    members = [[[i - d, i], [float(rng.uniform(0.5, 2.0)), 1e7, 0.1]]
               for i in range(1, n_joints) for d in (1, 2, 3) if i - d >= 0]
    forces = [[j, [float(rng.uniform(-1e3, 1e3)) for _ in range(3)]]
              for j in range(3, n_joints) if joints[j][1] == "NO" and rng.random() < 0.7]
    if not forces:   # (a load on a roller joint: its free axes carry it)
        forces = [[n_joints - 1, [100.0, -50.0, 25.0]]]
    return {"joint": joints, "force": forces, "member": members}


def test_every_3d_system_size_through_the_small_and_the_staged_path(gpu):
    """The 3D counterpart of the sweep below (the every-size sweeps were 2D only): tower trusses with
    n_free = 2 ... 201 - every residue modulo 16 and 64, in particular 15 / 16 / 17, 63 / 64 / 65, 127 / 128 / 129 -
    as one ragged batch through `solve_batch` (the sizes up to 128 take the fused small-system kernel, the others the
    staged pipeline: natural order, device order, RCM) and once more with EVERY size forced through the staged
    pipeline; each truss against the oracle."""
    from python_stable_3d_truss_analysis_amd import batch
    cases = [_strip_truss_json(nj, seed=3 * nj + r, dim=3, rollers=r)
             for nj in range(4, 71) for r in (0, 1, 2) if not (nj == 4 and r == 2)]
    packed = batch.pack_json(cases)
    sizes = sorted(set(int(v) for v in packed.n_free))
    assert sizes == list(range(2, 202))
    refs = [orc.solve(data) for data in cases]

    def check(res, tag):
        assert not res.info.any(), tag
        for b, (data, ref) in enumerate(zip(cases, refs)):
            nJ, nM = len(data["joint"]), len(data["member"])
            assert H.max_scaled_err(res.displace[b, :nJ], ref["u"]) <= 1e-8, (b, tag)
            assert H.max_scaled_err(res.external[b, :nJ], ref["f_ext"]) <= 1e-8, (b, tag)
            assert H.max_scaled_err(res.internal[b, :nM], ref["N"]) <= 1e-8, (b, tag)

    for reorder in (False, True, "rcm"):
        check(batch.solve_batch(packed, reorder=reorder), reorder)
    staged = gpu.DeviceBatch(packed, use_small=False)      # one padded batch, every size on the staged kernels
    assert not staged.small
    staged.solve()
    check(staged.result(), "staged")
    ordered = gpu.DeviceBatch(packed, use_small=False, reorder=True)
    ordered.solve()
    check(ordered.result(), "staged, device order")


def test_every_system_size_around_the_tile_and_panel_boundaries(gpu):
    """Padding, envelope and tile logic at awkward sizes: 2D strip trusses with n_free = 1 ... 200
    (every residue modulo 16 and 64 many times over), natural and RCM order, one ragged batch, each
    truss against the oracle."""
    from python_stable_3d_truss_analysis_amd import batch
    cases = [_strip_truss_json(nj, seed=nj, extra_pin=(nj % 3 == 0)) for nj in range(3, 104)]
    packed = batch.pack_json(cases)
    sizes = sorted(set(int(v) for v in packed.n_free))
    assert sizes[0] <= 3 and sizes[-1] >= 200 and len(sizes) >= 90 and {15, 16, 17, 63, 64, 65} & set(sizes)
    for reorder in (False, True, "rcm"):
        res = batch.solve_batch(packed, reorder=reorder)
        assert not res.info.any()
        for b, data in enumerate(cases):
            ref = orc.solve(data)
            nJ, nM = len(data["joint"]), len(data["member"])
            assert H.max_scaled_err(res.displace[b, :nJ, :2], ref["u"]) <= 1e-8, (b, reorder)
            assert H.max_scaled_err(res.external[b, :nJ, :2], ref["f_ext"]) <= 1e-8, (b, reorder)
            assert H.max_scaled_err(res.internal[b, :nM], ref["N"]) <= 1e-8, (b, reorder)
            assert not res.displace[b, :nJ, 2].any()   # a 2D truss never moves in z


def test_recover_without_lds_staging_matches(gpu):
    """The recovery kernel's path for trusses whose tables exceed a CU's LDS (u and f_ext live in the
    output arrays instead of LDS; the member ends at the supports in LDS lists, or - the fall-back - found by a
    walk over all members), forced at bar-942 size: displacements, member forces AND reactions bit for bit those
    of the staged path (one summation order, no floating-point atomics: `include/trs_solver.h`, "bit for bit")."""
    data = H.load_json("bar-942_input_0")
    dev = _device_batch(gpu, [data, data, data])
    dev.solve()
    ref = dev.result()
    assert np.abs(ref.external).max() > 0
    for scan in (False, True):
        dev.options["recover_unstaged"], dev.options["recover_scan"] = True, scan
        dev.u.fill_(float("nan")); dev.f_ext.fill_(float("nan")); dev.N.fill_(float("nan"))
        dev.recover()
        got = dev.result()
        dev.u.fill_(float("nan")); dev.f_ext.fill_(float("nan"))
        dev.solve()                       # the hints travel through trs_solve as well
        again = dev.result()
        for res in (got, again):
            np.testing.assert_array_equal(res.displace, ref.displace)
            np.testing.assert_array_equal(res.internal, ref.internal)
            np.testing.assert_array_equal(res.external, ref.external)


def test_recover_of_a_truss_beyond_the_lds_is_deterministic(gpu):
    """A 12 x 12 x 12-grid cube truss (~1 900 joints, ~10 000 members: `trs_recover` takes the path without LDS
    staging by itself): its two forms - member-end lists in LDS, walk over all members - agree bit for bit, run
    after run, in generator order and through a joint order (`joint_out`), and with the oracle to 1e-8."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    p = gen.generate_cube_batch([950, 700], gridRange=(12, 12, 12), seed=3)
    assert int(p.nJ.max()) > 1600
    ref = orc.solve(gen.packed_to_json(p, 0))
    nJ = int(p.nJ[0])
    for reorder in (False, True):
        dev = gpu.DeviceBatch(p, reorder=reorder)
        dev.solve()
        first = dev.result()
        assert not first.info.any()
        assert H.max_scaled_err(first.external[0, :nJ], ref["f_ext"]) <= 1e-8
        for scan in (False, True, False):
            dev.options["recover_scan"] = scan
            dev.options["recover_unstaged"] = scan
            dev.f_ext.fill_(float("nan"))
            dev.recover()
            got = dev.result()
            np.testing.assert_array_equal(got.external, first.external)
            np.testing.assert_array_equal(got.displace, first.displace)
            np.testing.assert_array_equal(got.internal, first.internal)


def test_resident_batch_with_a_joint_order_delivers_results_in_the_given_numbering(gpu):
    """`DeviceBatch(..., reorder=...)`: the kernels work on renumbered joints (narrower envelope), and
    `trs_recover` writes u / f_ext through `joint_out` - the caller sees the given numbering.  Checked
    against the goldens and against the batch solved in the given numbering, through `trs_solve` (one
    call), through the separate stages, and on the recovery path without LDS staging."""
    names = ["bar-942_input_0", "bar-120_input_0", "bar-47_input_0"]
    datas = [H.load_json(nm) for nm in names] + [d for _, d, _ in H.ragged_cube_cases()]
    z = H.dense_golden()
    golds = [{k: z[f"{nm}/{k}"] for k in ("u", "f_ext", "N")} for nm in names] + \
            [g for _, _, g in H.ragged_cube_cases()]
    packed = gpu.pack_json(datas)
    given = gpu.DeviceBatch(packed)
    given.solve()
    ref = given.result()
    cend_of = lambda dev: dev.env.cpu().numpy()[:, dev.rows // 16 + dev.rows // 64 + 8: 2 * (dev.rows // 16) + dev.rows // 64 + 8]
    tiles = lambda dev: int((cend_of(dev) - np.arange(dev.rows // 16)).clip(0).sum())
    for order in ("profile", "rcm", "fast", gpu.profile_permutation(packed)):
        dev = gpu.DeviceBatch(packed, reorder=order)
        assert dev.joint_out is not None and dev.joint_out.dtype == dev.torch.int32
        for how in ("one call", "stages", "unstaged recover"):
            dev.u.fill_(float("nan")); dev.f_ext.fill_(float("nan"))
            if how == "one call":
                dev.solve()
            else:
                dev.options["recover_unstaged"] = how == "unstaged recover"
                try:
                    dev.dofmap(); dev.assemble(); dev.potrf(); dev.potrs(); dev.recover()
                    dev.torch.cuda.synchronize()
                finally:
                    dev.options["recover_unstaged"] = False
            got = dev.result()
            assert not got.info.any()
            for b, (data, gold) in enumerate(zip(datas, golds)):
                dim, nJ, nM = orc.truss_dim(data), len(data["joint"]), len(data["member"])
                tag = (b, how, order if isinstance(order, str) else "array")
                assert H.max_scaled_err(got.displace[b, :nJ, :dim], gold["u"]) <= TOL_FP64, tag
                assert H.max_scaled_err(got.external[b, :nJ, :dim], gold["f_ext"]) <= TOL_FP64, tag
                assert H.max_scaled_err(got.internal[b, :nM], gold["N"]) <= TOL_FP64, tag
                assert not got.displace[b, nJ:].any() and not got.external[b, nJ:].any(), tag   # padding stays zero
            assert H.max_scaled_err(got.displace, ref.displace) <= 1e-9
        if isinstance(order, str) and order == "profile":
            assert tiles(dev) < tiles(given)      # the point of it: fewer stored tiles
    # a batch on the fused small-system path ignores the request (nothing to gain), from_device refuses it
    small = gpu.DeviceBatch(gpu.pack_json([H.load_json("bar-25_input_0")]), reorder=True)
    assert small.small and small.joint_out is None
    small.solve()
    assert not small.result().info.any()
    with pytest.raises(ValueError):
        gpu.DeviceBatch.from_device({f: getattr(small, f) for f in gpu.DeviceBatch.INPUT_FIELDS}, small.n_max,
                                    joint_out=small.torch.zeros([1, small.nJ_max], dtype=small.torch.int32,
                                                                device=small.device))


def test_launch_hints_and_forced_narrow_routing(gpu):
    """`envelope_reach` (host) equals what `trs_assemble` derives on the device; a batch it clears runs with
    TRS_ASM_ALL_NARROW / TRS_HINT_NO_WIDE / TRS_HINT_SUBSTITUTED (the kernels that would find nothing are not
    launched) and gives the same bits; a truss with a WIDE envelope forced through the wave-per-matrix kernels
    still matches the oracle (the hint can cost time, never correctness)."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    rng = np.random.default_rng(2)
    cubes = gen.generate_cube_batch(rng.integers(60, 191, size=24), gridRange=(6, 6, 6), seed=9)
    datas = [H.load_json("bar-942_input_0"), H.load_json("bar-120_input_0")]
    for packed in (gpu.pack_json(datas), cubes, gpu.permute_joints(cubes, gpu.profile_permutation(cubes))):
        plain = gpu.DeviceBatch(packed)
        plain.all_narrow = False
        plain.solve()
        ref = plain.result()
        env = plain.env.cpu().numpy()
        nchm, npan = plain.rows // 16, plain.rows // 64
        last = env[:, nchm:nchm + npan]
        npads = (packed.n_free + 63) // 64
        widest = np.array([max(int(last[b, j]) - (4 * j + 3) for j in range(int(npads[b]))) for b in range(packed.B)])
        np.testing.assert_array_equal(gpu.envelope_reach(packed), widest)
        wide_present = bool((env[:, nchm + npan] & 0xff != 1).any())
        assert wide_present == bool(widest.max() > gpu.NARROW_MAX_BELOW)
        forced = gpu.DeviceBatch(packed)
        forced.all_narrow = True                       # also for the batch that holds wide envelopes
        for how in ("one call", "stages"):
            forced.u.fill_(float("nan"))
            if how == "one call":
                forced.solve()
            else:
                forced.dofmap(); forced.assemble(); forced.potrf(); forced.potrs(); forced.recover()
            got = forced.result()
            assert not got.info.any()
            assert (forced.env.cpu().numpy()[:, nchm + npan] & 0xff == 1).all()      # every matrix routed narrow
            if not wide_present:                       # same kernels, same bits
                for k in ("displace", "external", "internal"):
                    np.testing.assert_array_equal(getattr(got, k), getattr(ref, k))
            else:
                assert H.max_scaled_err(got.displace, ref.displace) <= 1e-9
                assert H.max_scaled_err(got.internal, ref.internal) <= 1e-9
    # and against the oracle for one truss of the wide batch
    b = int(np.argmax(gpu.envelope_reach(cubes)))
    assert gpu.envelope_reach(cubes)[b] > gpu.NARROW_MAX_BELOW
    data = gen.packed_to_json(cubes, b)
    want = orc.solve(data)
    nJ, nM = len(data["joint"]), len(data["member"])
    # ... also with the compact form asked for: a matrix only FORCED narrow may hold more tiles per chunk than
    # the entry lists are sized for, so it keeps the slab (round 2 overran the list tables there); the
    # narrow-by-reach neighbour in the same batch does leave as lists
    for options in (None, {"compact": True}):
        dev = gpu.DeviceBatch(cubes.take(np.array([b, int(np.argmin(gpu.envelope_reach(cubes)))])), options=options)
        dev.all_narrow = True
        dev.solve()
        got = dev.result()
        assert not got.info.any()
        assert H.max_scaled_err(got.displace[0, :nJ], want["u"]) <= 1e-8
        assert H.max_scaled_err(got.internal[0, :nM], want["N"]) <= 1e-8
        route = dev.env.cpu().numpy()[:, dev.rows // 16 + dev.rows // 64]
        assert (route & 0xff == 1).all() and not route[0] & 0x100
        assert bool(route[1] & 0x100) == (options is not None and
                                          gpu.envelope_reach(cubes).min() <= gpu.NARROW_MAX_BELOW)


def test_streamed_solver_drops_the_templates_launch_hints(gpu):
    """`StreamedSolver` builds its resident slots from a template; batches submitted later may have other
    envelopes, so the template's `all_narrow` must not stay (round 2: it did, and wide envelopes ran
    forced-narrow).  `same_topology=True` keeps it for streams that only vary coordinates / sections / loads."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    narrow = gpu.pack_json([H.load_json("bar-942_input_0")]).replicate(4)
    assert gpu.DeviceBatch(narrow).all_narrow
    pipe = gpu.StreamedSolver(narrow)
    assert not any(d.all_narrow for d in pipe.dev)
    keep = gpu.StreamedSolver(narrow, same_topology=True)
    assert all(d.all_narrow for d in keep.dev)
    for p in (pipe, keep):
        p.submit(p.stage(narrow))
        res = p.drain()[0]
        ref = gpu.solve_batch(narrow)
        np.testing.assert_array_equal(res.displace, ref.displace)


def test_empty_batch_and_fully_constrained_truss(gpu):
    """Degenerate inputs: a batch of zero trusses, and a truss without free DOFs next to a normal one
    (the reference's solve of a 0 x 0 system gives zero displacements, forces and reactions)."""
    from python_stable_3d_truss_analysis_amd import batch
    empty = batch.solve_batch([])
    assert empty.displace.shape[0] == 0 and empty.info.shape == (0,)
    pinned = {"joint": [[[0.0, 0.0, 0.0], "PIN"], [[100.0, 0.0, 0.0], "PIN"]],
              "force": [[1, [5.0, 0.0, 0.0]]], "member": [[[0, 1], [1.0, 1e7, 0.1]]]}
    normal = H.load_json("bar-25_input_0")
    res = batch.solve_batch(batch.pack_json([pinned, normal]))
    assert not res.info.any()
    assert not res.displace[0].any() and not res.internal[0].any() and not res.external[0].any()
    ref = orc.solve(normal)
    assert H.max_scaled_err(res.displace[1, :len(normal["joint"])], ref["u"]) <= TOL_FP64


def test_get_k_matrix_matches_the_reference_layout(gpu):
    """`Truss.GetKMatrix()` (reference truss.py:307-316) through the HIP assembly: the FULL nDOF x nDOF matrix
    (no support eliminated), DOF = joint * dim + axis, dim-sized for 2D trusses - against the oracle's member
    loop on all ten bundled cases, against the reference's own K_ff (captured by import, <= 120 bars) through the
    free-DOF mask, and in one ragged batch."""
    from python_stable_3d_truss_analysis_amd import Truss
    z = H.dense_golden()
    names = H.data_case_names()
    assert len(names) == 10
    for name in names:
        data = H.load_json(name)
        dim = orc.truss_dim(data)
        truss = Truss(dim).LoadFromJSON(data=data)
        K = truss.GetKMatrix()
        want = orc.global_K(data)
        assert isinstance(K, np.ndarray) and K.dtype == np.float64 and K.shape == want.shape == (truss.nJoint * dim,) * 2
        assert H.max_scaled_err(K, want) <= 1e-14, name
        np.testing.assert_array_equal(K, K.T)                    # the two halves are formed from the same products
        assert np.abs(K.sum(axis=0)).max() <= 1e-12 * np.abs(K).max()   # rigid translations carry no force
        mask = truss.GetDisplacementUnknownMask()
        if f"{name}/K_ff" in z.files:                            # the reference's own matrix, matK[mask,:][:,mask]
            assert H.max_scaled_err(K[mask][:, mask], z[f"{name}/K_ff"]) <= 1e-14, name
    datas = [H.load_json(n) for n in names]
    many = gpu.global_stiffness(gpu.pack_json(datas))
    assert len(many) == len(datas)
    for data, K in zip(datas, many):
        assert H.max_scaled_err(K, orc.global_K(data)) <= 1e-14
    assert gpu.global_stiffness([]) == []


def test_tile_hint_is_adopted_only_where_no_matrix_skips_tiles(gpu):
    """`DeviceBatch.adopt_tile_hint`: a tower-like truss (bar-942) has too few envelope tiles without an entry of
    K_ff for the skipping to pay - every matrix reports "all tiles written", the hint is adopted and the hinted
    solves (TRS_ASM_ALL_TILES: no masks formed) return the same bits; generated cube trusses do skip tiles and keep
    the default."""
    tower = gpu.DeviceBatch(gpu.pack_json([H.load_json("bar-942_input_0")]).replicate(8), reorder="profile")
    tower.solve()
    before = tower.result()
    assert tower.adopt_tile_hint() and tower.all_tiles
    tower.S.fill_(float("nan"))
    tower.solve()
    after = tower.result()
    for k in ("displace", "external", "internal", "info"):
        np.testing.assert_array_equal(getattr(after, k), getattr(before, k))
    cubes = gpu.DeviceBatch(gpu.pack_json([d for _, d, _ in H.ragged_cube_cases()][6:]), reorder="profile", use_small=False)
    cubes.solve()
    assert not cubes.adopt_tile_hint() and not cubes.all_tiles
    tower.upload(tower.pinned_inputs(tower.packed))      # another topology may follow: the hint goes
    assert not tower.all_tiles
