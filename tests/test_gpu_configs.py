"""GPU parity at the shapes BASELINE.json states (`-m gpu`): dense mode, config 3 at the full 65 536
batch, config 5 at 16 384 samples through the sharded entry point, and ill-conditioned SPD systems.

The oracle is the checker on samples it finishes in seconds; the full batches are checked through
size-independent properties (info == 0, global equilibrium, determinism)."""
import copy

import numpy as np
import pytest

from oracle import truss_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_NORTH_STAR = 1e-6
TOL_FP64 = 1e-9


def _check_against(res, b, data, gold, tol, tag):
    nJ, nM = len(data["joint"]), len(data["member"])
    dim = orc.truss_dim(data)
    assert H.max_scaled_err(res.displace[b, :nJ, :dim], gold["u"]) <= tol, (tag, "u")
    assert H.max_scaled_err(res.external[b, :nJ, :dim], gold["f_ext"]) <= tol, (tag, "f_ext")
    assert H.max_scaled_err(res.internal[b, :nM], gold["N"]) <= tol, (tag, "N")


# ---- dense mode (env = NULL): the path SURVEY 8(d)'s FLOP figure and bench.py --dense describe -------

def _solve_device(packed, use_envelope):
    from python_stable_3d_truss_analysis_amd import batch
    dev = batch.DeviceBatch(packed, use_envelope=use_envelope)
    dev.solve()
    return dev.result()


def test_dense_mode_matches_golden_vectors_and_envelope_mode():
    from python_stable_3d_truss_analysis_amd import batch
    z = H.dense_golden()
    for name in ("bar-942_input_0", "bar-120_input_0"):
        data = H.load_json(name)
        packed = batch.pack_json([data]).replicate(3)
        dense, env = _solve_device(packed, False), _solve_device(packed, True)
        assert not dense.info.any() and not env.info.any()
        gold = {k: z[f"{name}/{k}"] for k in ("u", "f_ext", "N")}
        for b in range(3):
            _check_against(dense, b, data, gold, TOL_FP64, (name, "dense"))
        assert np.array_equal(dense.displace[1], dense.displace[0])        # deterministic per copy
        assert H.max_scaled_err(dense.displace, env.displace) <= 1e-10     # same answer, other tile set
        assert H.max_scaled_err(dense.internal, env.internal) <= 1e-10
    cubes = list(H.ragged_cube_cases())
    picks = [cubes[1], cubes[-1]]                                          # a small and the largest fixture
    for name, data, gold in picks:                                         # one batch each: no bucketing
        packed = batch.pack_json([data])
        dense, env = _solve_device(packed, False), _solve_device(packed, True)
        assert not dense.info.any()
        _check_against(dense, 0, data, gold, TOL_FP64, (name, "dense"))
        assert H.max_scaled_err(dense.displace, env.displace) <= 1e-9


def test_dense_mode_ragged_batch_padding():
    """Dense mode on a ragged batch in ONE DeviceBatch (padded to the largest system): the identity
    padding and the per-truss sizes must not leak between trusses."""
    from python_stable_3d_truss_analysis_amd import batch
    names = [n for n in H.data_case_names() if "942" not in n]
    datas = [H.load_json(n) for n in names]
    res = _solve_device(batch.pack_json(datas), False)
    z = H.dense_golden()
    assert not res.info.any()
    for b, (name, data) in enumerate(zip(names, datas)):
        _check_against(res, b, data, {k: z[f"{name}/{k}"] for k in ("u", "f_ext", "N")}, TOL_FP64, name)


# ---- ill-conditioned but SPD: Cholesky must not raise where the reference's LU returns an answer ------

@pytest.mark.parametrize("spread,name", [(9, "bar-120_input_0"), (12, "bar-120_input_0"), (12, "bar-72_input_0")])
def test_ill_conditioned_spd_systems(spread, name):
    """Member stiffnesses spread over `spread` decades (cond(K_ff) 4e8 ... 1e13).  The reference
    (LU) solves these; the Cholesky path must too (info == 0).  Agreement with the reference is bounded
    by the conditioning, not by the method: both are backward stable, so the test asserts
    (a) |u - u_ref| <= cond * eps * |u| (the forward-error scale), (b) the north-star 1e-6 whenever
    cond * eps allows it, and (c) the componentwise backward error |K u - f| <= 8 n eps (|K| |u| + |f|)
    independently of cond."""
    from python_stable_3d_truss_analysis_amd import Truss, batch
    data = copy.deepcopy(H.load_json(name))
    rng = np.random.default_rng(1)
    for m in data["member"]:
        m[1][1] = m[1][1] * 10 ** rng.uniform(-spread / 2, spread / 2)
    ref = orc.solve(data)
    cond = np.linalg.cond(ref["K_ff"])
    assert cond > 1e8
    res = batch.solve_batch(batch.pack_json([data]))
    assert int(res.info[0]) == 0
    nJ, nM = len(data["joint"]), len(data["member"])
    err = H.max_scaled_err(res.displace[0, :nJ], ref["u"])
    assert err <= cond * 2.2e-16, (cond, err)
    if cond * 2.2e-16 <= 1e-7:
        assert err <= TOL_NORTH_STAR
    u_free = res.displace[0, :nJ].reshape(-1)[ref["mask"]]
    f_free = orc.force_vector(data)[ref["mask"]]
    bound = 8 * len(u_free) * 2.2e-16 * (np.abs(ref["K_ff"]) @ np.abs(u_free) + np.abs(f_free))
    assert (np.abs(ref["K_ff"] @ u_free - f_free) <= bound).all()
    truss = Truss(3).LoadFromJSON(data=data)
    truss.Solve()                                            # the drop-in path: no LinAlgError
    assert truss.isSolved


# ---- config 3: GenerateRandomCubeTrusses-like mixed batch, B = 65 536, one GPU ------------------------

def test_config3_full_batch_65536_random_cube_trusses():
    from python_stable_3d_truss_analysis_amd import batch
    from python_stable_3d_truss_analysis_amd import generate as gen
    B = 65536
    rng = np.random.default_rng(0)
    packed = gen.generate_cube_batch(rng.integers(8, 191, size=B), gridRange=(6, 6, 6), seed=7)
    assert packed.B == B and packed.nM.min() >= 100 and packed.nM.max() >= 2000
    res = batch.solve_batch(packed, reorder=True)
    assert not res.info.any()
    assert np.isfinite(res.displace).all() and np.isfinite(res.internal).all()
    # global equilibrium of every truss: loads + reactions sum to zero
    total = res.external.sum(axis=1)
    scale = np.abs(packed.loads).sum(axis=(1, 2))
    assert (np.abs(total).max(axis=1) <= 1e-8 * scale).all()
    # reactions only at supports, applied loads untouched at free joints
    free = packed.cbits == 0
    assert np.array_equal(res.external[free], packed.loads[free])
    # oracle sample across the size range (every bucket of 64 free DOFs that occurs, >= 16 trusses)
    n_pad = (packed.n_free + 63) // 64
    picks = [int(np.flatnonzero(n_pad == v)[0]) for v in np.unique(n_pad)]
    picks += [int(i) for i in rng.choice(B, size=max(0, 16 - len(picks)), replace=False)]
    assert len(picks) >= 14
    for b in picks:
        data = gen.packed_to_json(packed, b)
        ref = orc.solve(data)
        _check_against(res, b, data, ref, 1e-8, ("config3", b, int(packed.n_free[b])))
    # the generator order (no reordering) and plain RCM give the same answers on a slice
    part = packed.take(np.arange(0, B, 64))
    for order in (False, "rcm"):
        other = batch.solve_batch(part, reorder=order)
        assert not other.info.any()
        assert H.max_scaled_err(other.displace, res.displace[::64, :other.displace.shape[1]]) <= 1e-7, order
        assert H.max_scaled_err(other.internal, res.internal[::64, :other.internal.shape[1]]) <= 1e-7, order


def test_pinned_inputs_and_result_pool_give_the_same_results():
    """`PackedBatch.pinned()` + `solve_batch(..., pool=ResultPool())`: DMA upload, download into reused
    page-locked buffers, joint order found on a worker thread - bitwise the results of the plain call."""
    from python_stable_3d_truss_analysis_amd import batch
    from python_stable_3d_truss_analysis_amd import generate as gen
    rng = np.random.default_rng(3)
    packed = gen.generate_cube_batch(rng.integers(8, 191, size=2048), gridRange=(6, 6, 6), seed=5)
    plain = batch.solve_batch(packed, reorder=True)
    pool = batch.ResultPool()
    pinned = packed.pinned()
    for f in ("xyz", "conn", "E", "A", "cbits", "loads", "nJ", "nM", "n_free"):
        np.testing.assert_array_equal(getattr(pinned, f), getattr(packed, f))
    for _ in range(2):                      # the second call reuses the pool's buffers
        got = batch.solve_batch(pinned, reorder=True, pool=pool)
        for k in ("displace", "external", "internal", "info"):
            np.testing.assert_array_equal(getattr(got, k), getattr(plain, k))
    # two solves of one call land in two buffers of the pool (the generator's members all have the section
    # (1, 1e7, 0.1), so doubling the area halves the displacements and leaves the member forces alone)
    two = batch.solve_batch(pinned, reorder="rcm", pool=pool, sections=[None, (2.0, 1e7, 0.1)])
    assert not two[0].info.any() and not two[1].info.any()
    assert H.max_scaled_err(two[0].displace, plain.displace) <= 1e-8
    assert H.max_scaled_err(2.0 * two[1].displace, two[0].displace) <= 1e-8
    assert H.max_scaled_err(two[1].internal, two[0].internal) <= 1e-8


# ---- config 5: dataset generation sharded over worker processes, HeteroData-shaped tensors -------------

def test_config5_dataset_16384_samples_through_the_sharded_entry_point():
    import torch
    from python_stable_3d_truss_analysis_amd import MemberType, TaskType, MetapathType
    from python_stable_3d_truss_analysis_amd import data as gdata
    from python_stable_3d_truss_analysis_amd import generate as gen
    from python_stable_3d_truss_analysis_amd import shard
    B = 16384
    rng = np.random.default_rng(5)
    types = [[0.5 + 0.25 * i, 1e7, 0.1] for i in range(8)]
    packed = gen.generate_cube_batch(rng.integers(4, 41, size=B), gridRange=(5, 5, 5), memberTypes=types, seed=11)
    fixed = MemberType(1., 1e7, 0.1)
    ndev = torch.cuda.device_count()
    with shard.ShardedSolver([f"cuda:{i % ndev}" for i in range(max(2, ndev))]) as pool:
        actual, prior = gdata.solve_actual_and_prior(packed, fixed, reorder=True, pool=pool)
    assert not actual.info.any() and not prior.info.any()
    graphs = gdata.hetero_tensors_batch(packed, actual, prior, fixed.a, TaskType.REGRESSION,
                                        MetapathType.NO_IMPLICIT, forceScale=1e3, displaceScale=0.1,
                                        positionScale=100.)
    assert len(graphs) == B
    for b in (0, 1, B // 2, B - 1):
        nJ, nM = int(packed.nJ[b]), int(packed.nM[b])
        g = graphs[b]
        assert tuple(g["joint"].x.shape) == (nJ, 10) and tuple(g["member"].x.shape) == (nM, 10)
        assert tuple(g["joint"].y.shape) == (nJ, 3) and tuple(g["member"].y.shape) == (nM, 1)
        assert tuple(g["joint", "j2m", "member"].edge_index.shape) == (2, 2 * nM)
    # the sharded two-solve result against the oracle on a sample, both section sets
    for b in [int(i) for i in rng.choice(B, size=8, replace=False)]:
        data = gen.packed_to_json(packed, b)
        _check_against(actual, b, data, orc.solve(data), 1e-8, ("config5 actual", b))
        for m in data["member"]:
            m[1] = [1., 1e7, 0.1]
        _check_against(prior, b, data, orc.solve(data), 1e-8, ("config5 prior", b))
        nJ, nM = len(data["joint"]), len(data["member"])
        want = (np.where(np.abs(actual.internal[b, :nM]) < 1e-10, 0.0, actual.internal[b, :nM])
                / packed.A[b, :nM] / 1e3).astype(np.float32)
        np.testing.assert_array_equal(graphs[b]["member"].y.numpy().ravel(), want)
    # single-process result is bitwise the same as the sharded one
    one = gdata.solve_actual_and_prior(packed.take(np.arange(512)), fixed, device="cuda:0", reorder=True)
    assert np.array_equal(one[0].displace, actual.displace[:512, :one[0].displace.shape[1]])
    assert np.array_equal(one[1].internal, prior.internal[:512, :one[1].internal.shape[1]])


# ---- feed path: overlapped upload / solve / download ------------------------------------------------------

def test_streamed_solver_pipelines_a_stream_of_batches():
    """`batch.StreamedSolver`: six different batches (bar-942 copies with scaled loads and sections)
    through the three-stream pipeline with two resident slots; every result equals the plain solve of
    that batch bit for bit, in submission order."""
    from python_stable_3d_truss_analysis_amd import batch
    base = batch.pack_json([H.load_json("bar-942_input_0")]).replicate(64)
    rng = np.random.default_rng(3)
    batches = []
    for k in range(6):
        p = base.take(np.arange(64))
        p.loads *= rng.uniform(0.5, 2.0, size=(64, 1, 1))
        p.A *= rng.uniform(0.8, 1.25, size=p.A.shape)
        batches.append(p)
    pipe = batch.StreamedSolver(base, slots=2)
    got = []
    for p in batches:
        done = pipe.submit(pipe.stage(p))
        if done is not None:
            got.append(batch.BatchResult(*(np.array(a) for a in (done.displace, done.external, done.internal, done.info))))
    got += [batch.BatchResult(*(np.array(a) for a in (d.displace, d.external, d.internal, d.info))) for d in pipe.drain()]
    assert len(got) == 6
    for p, res in zip(batches, got):
        dev = batch.DeviceBatch(p)
        dev.solve()
        want = dev.result()
        assert not want.info.any()
        np.testing.assert_array_equal(res.displace, want.displace)
        np.testing.assert_array_equal(res.external, want.external)
        np.testing.assert_array_equal(res.internal, want.internal)


def test_config5_one_ranks_share_of_the_million_sample_dataset_through_dataset_chunks():
    """BASELINE config 5 at a rank's REAL share: rank 0 of 8 of a 1e6-sample dataset = the chunks 0, 8, 16, ...
    of `data.dataset_chunks` (131 072 samples >= 125 000), two solves per sample (actual sections + the fixed
    prior, reference data.py:107-114), joint order and feature kernel on the device, HeteroData-shaped float32
    tensors per chunk.  Size checks on every chunk, every solve status clean, regression targets finite, and an
    oracle sample per chunk (both solves, through the features: joint.y = u / displaceScale, member.y = N / A /
    forceScale, joint.x[6:9] = prior displacement)."""
    import torch
    from python_stable_3d_truss_analysis_amd import MemberType, TaskType
    from python_stable_3d_truss_analysis_amd import data as gdata
    from python_stable_3d_truss_analysis_amd import generate as gen
    fixed = MemberType(1., 1e7, 0.1)
    types = [[0.5 + 0.25 * i, 1e7, 0.1] for i in range(8)]
    scales = dict(forceScale=1e3, displaceScale=0.1, positionScale=100.)
    chunk, total, world = 16384, 1_000_000, 8
    seen, firsts = 0, []
    rng = np.random.default_rng(1)
    for first, packed, t in gdata.dataset_chunks(total, rank=0, world=world, chunk=chunk, seed=3, numCubeRange=(8, 190),
                                                 gridRange=(6, 6, 6), fixedMemberType=fixed,
                                                 taskType=TaskType.REGRESSION, memberTypes=types, **scales):
        B = packed.B
        firsts.append(first)
        assert B == min(chunk, total - first) and first % (chunk * world) == 0
        assert tuple(t["info"].shape) == (2, B) and not bool(t["info"].any().item())
        assert tuple(t["joint_x"].shape) == (B, packed.nJ_max, 10) and tuple(t["member_x"].shape) == (B, packed.nM_max, 10)
        assert tuple(t["joint_y"].shape) == (B, packed.nJ_max, 3) and tuple(t["member_y"].shape) == (B, packed.nM_max, 1)
        assert t["joint_x"].dtype == torch.float32 and t["joint_x"].is_cuda
        assert bool(torch.isfinite(t["joint_y"]).all().item()) and bool(torch.isfinite(t["member_y"]).all().item())
        assert int(packed.nM.min()) >= 100 and int(packed.n_free.max()) <= 882
        # the chunk is the dataset's samples first .. first + B - 1, whatever the chunking (global-index keyed)
        np.testing.assert_array_equal(gdata.dataset_sizes(3, first, 4, (8, 190)),
                                      gdata.dataset_sizes(3, 0, first + 4, (8, 190))[first:])
        host = packed.to_packed(t["inputs"])      # (generated on the device: `packed` carries the sizes only)
        for b in [int(i) for i in rng.choice(B, size=2, replace=False)]:
            data = gen.packed_to_json(host, b)
            nJ, nM = len(data["joint"]), len(data["member"])
            ref = orc.solve(data)
            got_u = t["joint_y"][b, :nJ].cpu().numpy().astype(np.float64) * scales["displaceScale"]
            got_s = t["member_y"][b, :nM, 0].cpu().numpy().astype(np.float64) * scales["forceScale"]
            assert H.max_scaled_err(got_u, ref["u"]) <= 2e-6, (first, b)            # float32 features
            assert H.max_scaled_err(got_s, ref["N"] / host.A[b, :nM]) <= 2e-6, (first, b)
            for m in data["member"]:
                m[1] = [fixed.a, fixed.e, fixed.density]
            pri = orc.solve(data)
            got_p = t["joint_x"][b, :nJ, 6:9].cpu().numpy().astype(np.float64) * scales["displaceScale"]
            assert H.max_scaled_err(got_p, pri["u"]) <= 2e-6, (first, b)
        seen += B
    assert firsts == [k * chunk * world for k in range(len(firsts))] and len(firsts) == 8
    assert seen == 8 * chunk >= 125_000
