"""ABI 10: the TABLE member form (uint16 end joints, a uint8 type index per member, one (a, e, density) table per batch -
the reference's own description of a member, `[[j0, j1], [a, e, density]]`, truss.py:406-413, type.py:5-27) against the
general form (int32 end joints, E and A per member): the same bits from every entry point that reads members, on the
bundled cases, on generated cube trusses, resident and host-fed."""
import numpy as np
import pytest

from oracle import truss_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = ["bar-6_input_0", "bar-25_input_0", "bar-47_input_0", "bar-120_input_0", "bar-942_input_0"]


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from python_stable_3d_truss_analysis_amd import batch
    return batch


def _equal(a, b, what):
    for f in ("displace", "external", "internal", "info"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f"{what}: {f}")


@pytest.mark.parametrize("name", CASES)
def test_stages_give_the_same_bits_in_both_member_forms(gpu, name):
    """dofmap -> assemble -> potrf -> potrs -> recover, stage by stage through `trs_*` and `trs_*_tab`: the slab the
    assembly writes, the factor, and the results are bit for bit the same; the table form against the oracle."""
    data = H.load_json(name)
    general = gpu.pack_json([data, data])
    table = gpu.pack_json([data, data], members="table")
    devs = [gpu.DeviceBatch(p, use_small=False, reorder=False) for p in (general, table)]
    assert not devs[0].table and devs[1].table
    for d in devs:
        d.S.fill_(float("nan"))
        d.dofmap(); d.assemble()
    S = [d.S.cpu().numpy() for d in devs]
    np.testing.assert_array_equal(S[0], S[1])               # (NaN poison included: the same entries are written)
    for d in devs:
        d.potrf(); d.potrs(); d.recover()
    res = [d.result() for d in devs]
    _equal(res[0], res[1], name)
    np.testing.assert_array_equal(devs[0].S.cpu().numpy(), devs[1].S.cpu().numpy())
    ref = orc.solve(data)
    nJ, nM = len(data["joint"]), len(data["member"])
    dim = len(data["joint"][0][0])
    assert H.max_scaled_err(res[1].displace[0, :nJ, :dim], ref["u"]) <= 1e-9
    assert H.max_scaled_err(res[1].internal[0, :nM], ref["N"]) <= 1e-9
    assert H.max_scaled_err(res[1].external[0, :nJ, :dim], ref["f_ext"]) <= 1e-9


@pytest.mark.parametrize("reorder", [True, "profile", "rcm", "fast"])
def test_resident_batch_with_a_joint_order_in_the_table_form(gpu, reorder):
    """`DeviceBatch(table-form batch, reorder=...)`: the order found on the device (`trs_joint_order_tab`: uint16 end
    joints read and written) or on the host (the end joints renumbered, the type indices untouched); results in the
    caller's numbering, bit for bit those of the general form, through one call and through the stages."""
    datas = [H.load_json(n) for n in ("bar-942_input_0", "bar-120_input_0", "bar-47_input_0")]
    general, table = gpu.pack_json(datas), gpu.pack_json(datas, members="table")
    devs = [gpu.DeviceBatch(p, use_small=False, reorder=reorder) for p in (general, table)]
    assert devs[1].table and devs[1].joint_out is not None
    np.testing.assert_array_equal(devs[0].joint_out.cpu().numpy(), devs[1].joint_out.cpu().numpy())
    for d in devs:
        d.solve()
    res = [d.result() for d in devs]
    _equal(res[0], res[1], f"reorder={reorder}")
    for d in devs:
        d.u.fill_(float("nan")); d.N.fill_(float("nan"))
        d.dofmap(); d.assemble(); d.potrf(); d.potrs(); d.recover()
    _equal(devs[0].result(), devs[1].result(), f"stages, reorder={reorder}")
    _equal(devs[1].result(), res[0], "stages vs one call")
    ref = orc.solve(datas[0])
    assert H.max_scaled_err(res[1].displace[0, :len(datas[0]["joint"])], ref["u"]) <= 1e-9


def test_solve_batch_of_the_bundled_cases_in_the_table_form(gpu):
    """All bundled cases in ONE ragged batch through `solve_batch`: small ones on the fused kernel
    (`trs_solve_small_tab`), the rest staged; device joint order, a host joint order ("rcm") and none; two section
    variants; the recovery without LDS staging - all bit for bit the general form's results."""
    datas = [H.load_json(n) for n in CASES + ["bar-10_input_0", "bar-72_input_0"]]
    general = gpu.pack_json(datas)
    table = gpu.pack_json(datas, members="auto")
    assert table.is_table
    for reorder in (False, True, "rcm"):
        _equal(gpu.solve_batch(general, reorder=reorder), gpu.solve_batch(table, reorder=reorder), f"reorder={reorder}")
    fixed = (2.5, 1.0e7, 0.3)
    a = gpu.solve_batch(general, reorder=True, sections=[None, fixed])
    b = gpu.solve_batch(table, reorder=True, sections=[None, fixed])
    for k in range(2):
        _equal(a[k], b[k], f"section variant {k}")
    assert np.abs(a[0].internal - a[1].internal).max() > 0
    _equal(gpu.solve_batch(general, reorder=True, options={"recover_unstaged": True}),
           gpu.solve_batch(table, reorder=True, options={"recover_unstaged": True}), "unstaged recovery")
    _equal(gpu.solve_batch(general, reorder=True, options={"compact": True}),
           gpu.solve_batch(table, reorder=True, options={"compact": True}), "compact form")


def test_fused_small_kernel_with_fitness_in_the_table_form(gpu):
    """`trs_solve_small_tab` with the GA reductions: areas and densities come from the type table."""
    data = H.load_json("bar-120_input_0")
    general, table = gpu.pack_json([data] * 3), gpu.pack_json([data] * 3, members="table")
    got = []
    for p in (general, table):
        dev = gpu.DeviceBatch(p)
        assert dev.small
        w, sv, dv = dev.solve_fitness(30000.0, 10.0)
        got.append((dev.result(), w.cpu().numpy(), sv.cpu().numpy(), dv.cpu().numpy()))
    _equal(got[0][0], got[1][0], "small")
    for k in (1, 2, 3):
        np.testing.assert_array_equal(got[0][k], got[1][k])
    assert got[1][1][0] > 0


def test_ragged_cube_batch_resident_and_host_fed_in_the_table_form(gpu):
    """1 500 generated cube trusses (the generator's one member type -> a table of one row): the resident bucket pipeline
    (`trs_joint_order_rows_tab` -> `trs_solve_rows_tab`, four lanes) and the host-fed pipeline (5 instead of 24 bytes per
    member pulled over PCIe) against the general form, bit for bit; the bytes the host batch holds."""
    import torch
    from python_stable_3d_truss_analysis_amd import generate as gen
    rng = np.random.default_rng(5)
    general = gen.generate_cube_batch(rng.integers(1, 191, size=1500), gridRange=(6, 6, 6), seed=9)
    table = general.table()
    assert table.types.shape == (1, 3) and not table.type_idx.any()
    res = []
    for p in (general, table):
        solver = gpu.RaggedSolver(p, reorder=True)
        assert solver.table == p.is_table
        solver.step()
        solver.adopt_launch_hints()
        solver.step()
        res.append(solver.result())
        del solver
    _equal(res[0], res[1], "resident")
    assert not res[0].info.any()
    streamed = [gpu.solve_batch_streamed(p.pinned(), reorder=True) for p in (general, table)]
    _equal(streamed[0], streamed[1], "host-fed")
    _equal(streamed[1], res[0], "host-fed vs resident")
    member_bytes = lambda p: sum(getattr(p, f).nbytes for f in ("conn", "E", "A", "type_idx") if getattr(p, f) is not None)
    assert member_bytes(table) * 24 == member_bytes(general) * 5
    torch.cuda.empty_cache()


def test_large_trusses_on_the_work_group_kernels(gpu):
    """`options={"all_wide": True}` (TRS_ASM_ALL_WIDE / TRS_HINT_ALL_WIDE): every matrix on the work-group factorisation
    and substitution whatever its envelope - same answers as the wave-per-matrix route to rounding, both against the
    oracle."""
    from python_stable_3d_truss_analysis_amd import generate as gen
    p = gen.generate_cube_batch([320, 400, 60, 5], gridRange=(8, 8, 8), seed=21)
    narrow = gpu.solve_batch(p, reorder=True)
    wide = gpu.solve_batch(p, reorder=True, options={"all_wide": True})
    assert not narrow.info.any() and not wide.info.any()
    scale = np.abs(narrow.displace).max(axis=(1, 2), keepdims=True)
    assert (np.abs(wide.displace - narrow.displace) / scale).max() <= 1e-9
    ref = orc.solve(gen.packed_to_json(p, 0))
    nJ, nM = int(p.nJ[0]), int(p.nM[0])
    assert H.max_scaled_err(wide.displace[0, :nJ], ref["u"]) <= 1e-8
    assert H.max_scaled_err(wide.internal[0, :nM], ref["N"]) <= 1e-8
    assert H.max_scaled_err(wide.external[0, :nJ], ref["f_ext"]) <= 1e-8
