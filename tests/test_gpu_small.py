"""The fused small-system kernel `trs_solve_small` (csrc/small.hip; reference path truss.py:329-364 for
n_free <= 128) through the C ABI, `-m gpu`: against the golden vectors, against the staged pipeline,
bit-reproducibility, the GA reductions, pivot failures and size sweeps across its block boundaries."""
import copy

import numpy as np
import pytest

from oracle import truss_oracle as orc
from tests import helpers as H
from tests.test_gpu_parity import _strip_truss_json

pytestmark = pytest.mark.gpu
TOL_FP64 = 1e-9


def _small_cases():
    names = [n for n in H.data_case_names() if "942" not in n] + H.cube7_case_names()
    return names, [H.load_json(n) for n in names]


def test_small_kernel_is_selected_and_matches_goldens():
    from python_stable_3d_truss_analysis_amd import batch
    names, datas = _small_cases()
    for d in datas:   # cube-7 files carry results: inputs only
        for k in ("displace", "external", "internal", "weight"):
            d.pop(k, None)
    packed = batch.pack_json(datas)
    assert packed.n_max <= batch.SMALL_N
    dev = batch.DeviceBatch(packed)
    assert dev.small and dev._slab is None
    dev.solve()
    res = dev.result()
    assert dev._slab is None                                   # the stiffness slab was never allocated
    assert not res.info.any()
    np.testing.assert_array_equal(dev.n_free.cpu().numpy(), packed.n_free)
    z = H.dense_golden()
    for b, (name, data) in enumerate(zip(names, datas)):
        ref = {k: z[f"{name}/{k}"] for k in ("u", "f_ext", "N")} if f"{name}/u" in z.files else orc.solve(data)
        dim, nJ, nM = orc.truss_dim(data), len(data["joint"]), len(data["member"])
        assert H.max_scaled_err(res.displace[b, :nJ, :dim], ref["u"]) <= TOL_FP64, name
        assert H.max_scaled_err(res.external[b, :nJ, :dim], ref["f_ext"]) <= TOL_FP64, name
        assert H.max_scaled_err(res.internal[b, :nM], ref["N"]) <= TOL_FP64, name
        assert not res.displace[b, nJ:].any() and not res.internal[b, nM:].any() and not res.external[b, nJ:].any()
    # against the staged pipeline on the same inputs (other factorisation order: agreement, not identity)
    staged = batch.DeviceBatch(packed, use_small=False)
    assert not staged.small
    staged.solve()
    ref = staged.result()
    assert H.max_scaled_err(res.displace, ref.displace) <= 1e-10
    assert H.max_scaled_err(res.internal, ref.internal) <= 1e-10
    assert H.max_scaled_err(res.external, ref.external) <= 1e-10
    np.testing.assert_array_equal(dev.free_index.cpu().numpy(), staged.free_index.cpu().numpy())


def test_small_kernel_is_bit_reproducible_and_copies_agree():
    from python_stable_3d_truss_analysis_amd import batch
    packed = batch.pack_json([H.load_json("bar-120_input_0")]).replicate(512)
    dev = batch.DeviceBatch(packed)
    assert dev.small
    dev.solve()
    a = dev.result()
    dev.solve()
    b = dev.result()
    for x, y in ((a.displace, b.displace), (a.external, b.external), (a.internal, b.internal)):
        np.testing.assert_array_equal(x, y)
        assert (x == x[0:1]).all()


def test_every_size_up_to_the_small_limit():
    """2D strips with n_free = 1 ... 129 in one ragged batch: every block count and every remainder
    modulo 16 of the fused kernel, natural and RCM order; the sizes beyond 128 take the staged path."""
    from python_stable_3d_truss_analysis_amd import batch
    cases = [_strip_truss_json(nj, seed=100 + nj, extra_pin=pin) for nj in range(3, 68) for pin in (False, True)]
    packed = batch.pack_json(cases)
    assert set(range(3, 130)) <= set(int(v) for v in packed.n_free)     # n = 2 nJ - 3 and 2 nJ - 4
    groups = batch.size_buckets(packed)
    assert any(packed.n_free[g].max() <= batch.SMALL_N and len(g) > 100 for g in groups)
    for reorder in (False, True):
        res = batch.solve_batch(packed, reorder=reorder)
        assert not res.info.any()
        for b, data in enumerate(cases):
            ref = orc.solve(data)
            nJ, nM = len(data["joint"]), len(data["member"])
            assert H.max_scaled_err(res.displace[b, :nJ, :2], ref["u"]) <= 1e-8, (b, reorder)
            assert H.max_scaled_err(res.external[b, :nJ, :2], ref["f_ext"]) <= 1e-8, (b, reorder)
            assert H.max_scaled_err(res.internal[b, :nM], ref["N"]) <= 1e-8, (b, reorder)


def test_small_kernel_fitness_reductions_match_trs_fitness():
    from python_stable_3d_truss_analysis_amd import batch
    data = H.load_json("bar-120_input_0")
    packed = batch.pack_json([data]).replicate(96)
    rng = np.random.default_rng(3)
    packed.A[:] = rng.uniform(0.5, 8.0, size=packed.A.shape)
    packed.rho[:] = rng.uniform(0.1, 1.0, size=packed.rho.shape)
    dev = batch.DeviceBatch(packed)
    dev.solve()
    first = dev.result()
    stress = np.abs(first.internal[:, :120]) / packed.A[:, :120]
    allow_s = float(np.median(stress.max(axis=1)))                       # about half the trusses violate
    allow_d = float(np.median(np.linalg.norm(first.displace, axis=2).max(axis=1)))
    w, sv, dv = (t.cpu().numpy() for t in dev.solve_fitness(allow_s, allow_d))
    res = dev.result()
    assert not res.info.any() and (sv > 0).any() and (dv > 0).any() and (sv == 0).any() and (dv == 0).any()
    w2, sv2, dv2 = (t.cpu().numpy() for t in dev.fitness(allow_s, allow_d))   # trs_fitness on the same u, N
    np.testing.assert_array_equal(w, w2)
    np.testing.assert_allclose(sv, sv2, rtol=1e-13, atol=0)
    np.testing.assert_allclose(dv, dv2, rtol=1e-13, atol=0)
    for b in (0, 17, 95):   # and against the oracle's restatement of ga.py:139-149
        d = copy.deepcopy(data)
        for m, mem in enumerate(d["member"]):
            mem[1] = [float(packed.A[b, m]), float(packed.E[b, m]), float(packed.rho[b, m])]
        ref = orc.solve(d)
        fit, ok_s, ok_d = orc.fitness_terms(d, ref, allow_s, allow_d)
        mine = w[b] + sv[b] / allow_s * 1e5 + dv[b] / allow_d * 1e5
        assert mine == pytest.approx(fit, rel=1e-9)


def test_small_kernel_reports_pivot_failures_and_refuses_oversized_systems():
    from python_stable_3d_truss_analysis_amd import batch, _capi
    cases = H.edge_cases()
    singular = [e["input"] for e in cases.values() if e.get("raises") == "LinAlgError"]
    assert singular
    good = H.load_json("bar-25_input_0")
    packed = batch.pack_json([good] + [s for s in singular if orc.truss_dim(s) == 3] + [good])
    dev = batch.DeviceBatch(packed)
    assert dev.small
    dev.solve()
    res = dev.result()
    assert res.info[0] == 0 and res.info[-1] == 0 and (res.info[1:-1] > 0).all()
    ref = orc.solve(good)
    assert H.max_scaled_err(res.displace[-1, :10], ref["u"]) <= TOL_FP64   # neighbours are unaffected
    lib = _capi.load()
    assert lib.trs_solve_small_fits(49, 120, 111) == 1 and lib.trs_solve_small_fits(244, 942, 696) == 0
    assert lib.trs_solve_small_fits(40000, 100, 100) == 0
    # a wrong host-side bound is refused per truss (info = -1), never an LDS overrun
    import torch
    big = batch.DeviceBatch(batch.pack_json([H.load_json("bar-120_input_0")]))
    big.n_max = 64
    big._solve_small()
    torch.cuda.synchronize()
    assert int(big.info.cpu()[0]) == -1
