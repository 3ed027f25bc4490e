"""Truss -> heterogeneous graph tensors (BASELINE config 5 sink).

Field layout, scaling, edge lists and the two-solve recipe (actual sections + one fixed
`MemberType` as a "prior") follow the reference's `TrussHeteroDataCreator`
(`slientruss3d/data.py:11-282`, `detail/to_PyG.md:138-189`):

  joint.x  = [position/positionScale, load/forceScale, priorDisplacement/displaceScale, isSupport]
  member.x = [centre/positionScale, 4 direction features, length/positionScale,
              priorStress/forceScale (, area for regression)]
  joint.y  = displacement/displaceScale        (regression)
  member.y = stress/forceScale                 (regression)  or  index into usedMemberTypes
  edges    = joint<->member incidence (+ implicit joint-joint / member-member when USE_IMPLICIT)

What differs from the reference: both solves of a whole batch of trusses run as TWO batched GPU
calls (`hetero_tensors_batch`), the features are built with array arithmetic, and the result is a
`torch_geometric.data.HeteroData` when PyG is importable, otherwise the same nested mapping
(`GraphStores`).  The reference solves twice per sample in Python (`data.py:20-23,107-114`).
"""
import os

import numpy as np

from .batch import BatchResult, PackedBatch, pack_trusses
from .truss import Truss
from .type import MemberType, MetapathType, SupportType, TaskType
from .utils import ZERO_EPS, InvalidTaskTypeError


class _Store(dict):
    """Attribute-style dict, the part of a PyG storage object this module needs."""
    __getattr__ = dict.get

    def __setattr__(self, key, value):
        self[key] = value


class GraphStores(dict):
    """Stand-in for `HeteroData` when torch_geometric is absent: `g['joint'].x`,
    `g['joint', 'j2m', 'member'].edge_index`, `g['src']`."""

    def __getitem__(self, key):
        if key not in self:
            dict.__setitem__(self, key, _Store())
        return dict.__getitem__(self, key)


_GRAPH_FACTORY = None


def _new_graph():
    global _GRAPH_FACTORY
    if _GRAPH_FACTORY is None:  # looked up once: a failed import is slow when repeated per graph
        try:
            from torch_geometric.data import HeteroData
            _GRAPH_FACTORY = HeteroData
        except ImportError:
            _GRAPH_FACTORY = GraphStores
    return _GRAPH_FACTORY()


def _sparsified(values, per_row):
    """The reference reads results through its sparse dicts: a joint whose every component is below
    1e-10 (or a member force below 1e-10) is absent there and contributes zeros (truss.py:344-359)."""
    v = np.array(values, dtype=np.float64, copy=True)
    if per_row:
        v[(np.abs(v) < ZERO_EPS).all(axis=1)] = 0.0
    else:
        v[np.abs(v) < ZERO_EPS] = 0.0
    return v


def _angles(p0, p1):
    """Vectorised `GetAngles` (utils.py:105-113): lower end first; (xy/L, z/L, y/xy, x/xy)."""
    swap = ~(p0[:, -1] < p1[:, -1])
    lo = np.where(swap[:, None], p1, p0)
    hi = np.where(swap[:, None], p0, p1)
    d = hi - lo
    full = np.sqrt((d ** 2).sum(axis=1))
    plan = np.sqrt((d[:, :2] ** 2).sum(axis=1))
    flat = np.abs(plan) < ZERO_EPS
    safe = np.where(flat, 1.0, plan)
    return np.stack([plan / full, d[:, 2] / full, np.where(flat, 0.0, d[:, 1] / safe),
                     np.where(flat, 0.0, d[:, 0] / safe)], axis=1)


def graph_arrays(xyz, conn, sections, support, loads, dim, actual, prior, taskType, metapathType,
                 forceScale=1., displaceScale=1., positionScale=1., usedMemberTypes=None):
    """Feature / target / edge arrays of ONE truss from dense arrays.

    xyz [nJ,dim], conn [nM,2], sections [nM,3]=(a,e,density), support [nJ] bool, loads [nJ,dim];
    `actual` / `prior` = (u [nJ,dim], N [nM]) of the solve with the real sections / the fixed
    section (prior may be None: isUseFixed=False)."""
    if taskType not in (TaskType.OPTIMIZATION, TaskType.REGRESSION):
        raise InvalidTaskTypeError(f"Invalid task type [{taskType}].")
    nJ, nM = len(xyz), len(conn)
    area = sections[:, 0]
    jx = [xyz / positionScale, loads / forceScale]
    if prior is not None:
        jx.append(_sparsified(prior[0], True) / displaceScale)
    jx.append(support.astype(np.float64)[:, None])
    p0, p1 = xyz[conn[:, 0]], xyz[conn[:, 1]]
    length = np.sqrt(((p1 - p0) ** 2).sum(axis=1))
    ang = _angles(p0, p1) if dim == 3 else None
    if ang is None:
        raise NotImplementedError("graph features are defined for 3D trusses (utils.py:105-113)")
    mx = [0.5 * (p0 + p1) / positionScale, ang, (length / positionScale)[:, None]]
    if prior is not None:
        mx.append((_sparsified(prior[1], False) / prior[2] / forceScale)[:, None])
    out = {"joint_x": np.concatenate(jx, axis=1)}
    if taskType == TaskType.REGRESSION:
        mx.append(area[:, None])
        out["joint_y"] = _sparsified(actual[0], True) / displaceScale
        out["member_y"] = (_sparsified(actual[1], False) / area / forceScale)[:, None]
    elif usedMemberTypes is not None:
        table = [MemberType(*row) for row in sections]
        out["member_y"] = np.array([[usedMemberTypes.index(t)] for t in table], dtype=np.int64)
    out["member_x"] = np.concatenate(mx, axis=1)
    members = np.arange(nM)
    out["j2m"] = np.stack([conn.reshape(-1), np.repeat(members, 2)])
    out["m2j"] = out["j2m"][::-1].copy()
    if metapathType == MetapathType.USE_IMPLICIT:
        out["j2j"], out["m2m"] = _implicit_edges(conn, nJ, nM)
    return out


def _to_graph(arrays, weight, source):
    import torch
    g = _new_graph()
    g["src"] = source
    g["originWeight"] = weight
    g["joint"].x = torch.tensor(arrays["joint_x"], dtype=torch.float32)
    g["member"].x = torch.tensor(arrays["member_x"], dtype=torch.float32)
    if "joint_y" in arrays:
        g["joint"].y = torch.tensor(arrays["joint_y"], dtype=torch.float32)
    if "member_y" in arrays:
        g["member"].y = torch.tensor(arrays["member_y"], dtype=torch.float32)
    g["joint", "j2m", "member"].edge_index = torch.tensor(arrays["j2m"], dtype=torch.long)
    g["member", "m2j", "joint"].edge_index = torch.tensor(arrays["m2j"], dtype=torch.long)
    if "j2j" in arrays:
        g["joint", "j2j", "joint"].edge_index = torch.tensor(arrays["j2j"], dtype=torch.long)
        g["member", "m2m", "member"].edge_index = torch.tensor(arrays["m2m"], dtype=torch.long)
    return g


def _implicit_edges(conn, nJ, nM):
    members = np.arange(nM)
    inc = np.zeros([nJ, nM], dtype=bool)
    inc[conn[:, 0], members] = True
    inc[conn[:, 1], members] = True
    jj = (inc.astype(np.int32) @ inc.T.astype(np.int32)) > 0
    mm = (inc.T.astype(np.int32) @ inc.astype(np.int32)) > 0
    return np.stack(np.nonzero(jj)), np.stack(np.nonzero(mm))


def _feature_shapes(B, nJm, nMm, has_prior, regression):
    FJ = 7 + (3 if has_prior else 0)
    FM = 8 + (1 if has_prior else 0) + (1 if regression else 0)
    return FJ, FM


def feature_tensors_host(packed: PackedBatch, actual: BatchResult, prior: BatchResult, fixedArea, taskType,
                         forceScale=1., displaceScale=1., positionScale=1.):
    """float32 feature tensors of a solved batch, formed natively on the HOST (`csrc/graphfeat.c`,
    OpenMP over the batch; same formulas as `graph_arrays`, which stays the single-truss path):
    dict joint_x [B,nJ,FJ], member_x [B,nM,FM], joint_y / member_y (regression), weight [B] (numpy)."""
    import ctypes
    import torch
    from .generate import _load
    B, nJm, nMm = packed.B, packed.nJ_max, packed.nM_max
    regression = taskType == TaskType.REGRESSION
    FJ, FM = _feature_shapes(B, nJm, nMm, prior is not None, regression)
    joint_x = torch.empty([B, nJm, FJ], dtype=torch.float32)
    member_x = torch.empty([B, nMm, FM], dtype=torch.float32)
    joint_y = torch.empty([B, nJm, 3], dtype=torch.float32) if regression else None
    member_y = torch.empty([B, nMm, 1], dtype=torch.float32) if regression else None
    weight = np.empty([B], dtype=np.float64)
    c = lambda a, t: np.ascontiguousarray(a, dtype=t)
    keep = [c(packed.xyz, np.float64), c(packed.conn, np.int32), c(packed.A, np.float64), c(packed.rho, np.float64),
            c(packed.cbits, np.uint8), c(packed.loads, np.float64), c(packed.nJ, np.int32), c(packed.nM, np.int32)]
    res = [c(actual.displace, np.float64), c(actual.internal, np.float64)] if regression else [None, None]
    res += [c(prior.displace, np.float64), c(prior.internal, np.float64)] if prior is not None else [None, None]
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    tptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    lib = _load()
    lib.trs_graph_features.restype = ctypes.c_int
    rc = lib.trs_graph_features(
        ctypes.c_int(B), ctypes.c_int(nJm), ctypes.c_int(nMm), *(ptr(a) for a in keep), *(ptr(a) for a in res),
        ctypes.c_double(fixedArea if prior is not None else 1.0), ctypes.c_double(forceScale),
        ctypes.c_double(displaceScale), ctypes.c_double(positionScale), ctypes.c_int(int(regression)),
        tptr(joint_x), tptr(member_x), tptr(joint_y), tptr(member_y), ptr(weight))
    if rc != 0:
        raise RuntimeError(f"trs_graph_features failed ({rc})")
    return {"joint_x": joint_x, "member_x": member_x, "joint_y": joint_y, "member_y": member_y,
            "weight": weight, "conn": torch.from_numpy(keep[1].astype(np.int64))}


def feature_tensors_device(packed, fixedMemberType, taskType, forceScale=1., displaceScale=1.,
                           positionScale=1., device=None, reorder=False, device_inputs=None):
    """The dataset sample pipeline entirely on the GPU (BASELINE config 5): both solves
    (`solve_batch(..., sections=[None, fixed], on_device=True)`: one upload, one reordering) and the
    feature kernel `trs_graph_features_dev` (`csrc/graphfeat.hip`) on the resident results - nothing but
    the float32 feature tensors ever needs to leave the device.  Returns the same dict as
    `feature_tensors_host` with torch tensors ON THE DEVICE (weight included), plus `info` [2,B]
    (status of the two solves).  Bit-identical to the host path.  `device_inputs`: the batch already lives
    on the device (`generate.generate_cube_batch_device`); `packed` then only carries its sizes."""
    import torch
    from . import _capi
    from .batch import solve_batch
    if (np.asarray(packed.dim) != 3).any():
        raise NotImplementedError("graph features are defined for 3D trusses (utils.py:105-113)")
    regression = taskType == TaskType.REGRESSION
    sections = [None] + ([(fixedMemberType.a, fixedMemberType.e, fixedMemberType.density)]
                         if fixedMemberType is not None else [])
    out = solve_batch(packed, device, reorder=reorder, sections=sections, on_device=True, device_inputs=device_inputs)
    actual, prior = out[0], (out[1] if fixedMemberType is not None else None)
    inp = actual.inputs
    dev = actual.displace.device
    B, nJm, nMm = packed.B, packed.nJ_max, packed.nM_max
    FJ, FM = _feature_shapes(B, nJm, nMm, prior is not None, regression)
    joint_x = torch.empty([B, nJm, FJ], dtype=torch.float32, device=dev)
    member_x = torch.empty([B, nMm, FM], dtype=torch.float32, device=dev)
    joint_y = torch.empty([B, nJm, 3], dtype=torch.float32, device=dev) if regression else None
    member_y = torch.empty([B, nMm, 1], dtype=torch.float32, device=dev) if regression else None
    weight = torch.empty([B], dtype=torch.float64, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    cont = lambda t: t if t.is_contiguous() else t.contiguous()
    ua, na = (cont(actual.displace), cont(actual.internal)) if regression else (None, None)
    up, npr = (cont(prior.displace), cont(prior.internal)) if prior is not None else (None, None)
    with torch.cuda.device(dev):
        _capi.check(_capi.load().trs_graph_features_dev(
            B, nJm, nMm, inp["xyz"].data_ptr(), inp["conn"].data_ptr(), inp["A"].data_ptr(), inp["rho"].data_ptr(),
            inp["cbits"].data_ptr(), inp["loads"].data_ptr(), inp["nJ"].data_ptr(), inp["nM"].data_ptr(),
            ptr(ua), ptr(na), ptr(up), ptr(npr), float(fixedMemberType.a if prior is not None else 1.0),
            float(forceScale), float(displaceScale), float(positionScale), int(regression), ptr(joint_x),
            ptr(member_x), ptr(joint_y), ptr(member_y), weight.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream), "trs_graph_features_dev")
    info = torch.stack([actual.info, prior.info if prior is not None else torch.zeros_like(actual.info)])
    return {"joint_x": joint_x, "member_x": member_x, "joint_y": joint_y, "member_y": member_y,
            "weight": weight, "conn": inp["conn"], "info": info}


class GraphList:
    """The graphs of a solved batch as a sequence: `len`, indexing, slicing and iteration like a list, but a
    graph OBJECT (HeteroData when torch_geometric is importable) is only built when it is asked for - its
    tensors are slices of the batch feature tensors (host or device), so building all graphs of a
    100 000-sample batch up front would cost more Python time than the two solves and the feature kernel."""

    def __init__(self, packed, tensors, metapathType, sources):
        import torch
        self.packed, self.tensors, self.metapathType, self.sources = packed, tensors, metapathType, sources
        conn = tensors["conn"].long()   # (int32 in the device tensors: a third of their download)
        B, nMm = packed.B, packed.nM_max
        member_ids = torch.arange(nMm, device=conn.device).repeat_interleave(2).expand(B, -1)
        self._j2m = torch.stack([conn.reshape(B, -1), member_ids], dim=1)
        self._m2j = torch.flip(self._j2m, dims=[1])
        weight = tensors["weight"]
        self._weight = weight.cpu().numpy() if hasattr(weight, "cpu") else weight

    def __len__(self):
        return self.packed.B

    def __iter__(self):
        return (self[b] for b in range(len(self)))

    def __getitem__(self, b):
        import torch
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        if b < 0:
            b += len(self)
        if not 0 <= b < len(self):
            raise IndexError(b)
        t, packed = self.tensors, self.packed
        nJ, nM = int(packed.nJ[b]), int(packed.nM[b])
        g = _new_graph()
        g["src"] = None if self.sources is None else self.sources[b]
        g["originWeight"] = float(self._weight[b])
        g["joint"].x = t["joint_x"][b, :nJ]
        g["member"].x = t["member_x"][b, :nM]
        if t["joint_y"] is not None:
            g["joint"].y = t["joint_y"][b, :nJ]
            g["member"].y = t["member_y"][b, :nM]
        g["joint", "j2m", "member"].edge_index = self._j2m[b, :, :2 * nM]
        g["member", "m2j", "joint"].edge_index = self._m2j[b, :, :2 * nM]
        if self.metapathType == MetapathType.USE_IMPLICIT:
            jj, mm = _implicit_edges(packed.conn[b, :nM], nJ, nM)
            dev = t["conn"].device
            g["joint", "j2j", "joint"].edge_index = torch.from_numpy(jj).to(dev)
            g["member", "m2m", "member"].edge_index = torch.from_numpy(mm).to(dev)
        return g


def graphs_from_tensors(packed: PackedBatch, tensors, metapathType=MetapathType.NO_IMPLICIT, sources=None):
    """One graph per truss whose tensors are SLICES of the batch feature tensors (host or device), as a
    lazily materialised sequence (`GraphList`; `list(...)` builds every graph object)."""
    return GraphList(packed, tensors, metapathType, sources)


def hetero_tensors_batch(packed: PackedBatch, actual: BatchResult, prior: BatchResult, fixedArea,
                         taskType=TaskType.OPTIMIZATION, metapathType=MetapathType.NO_IMPLICIT,
                         forceScale=1., displaceScale=1., positionScale=1., sources=None):
    """Graphs of a whole solved batch from HOST results: `actual` / `prior` are the dense results of the
    two batched solves (`prior` may be None).  Returns one graph per truss, holding slices of float32
    batch tensors (`feature_tensors_host`).  `dataset_graphs` is the all-device form."""
    if taskType not in (TaskType.OPTIMIZATION, TaskType.REGRESSION):
        raise InvalidTaskTypeError(f"Invalid task type [{taskType}].")
    if (np.asarray(packed.dim) != 3).any():
        raise NotImplementedError("graph features are defined for 3D trusses (utils.py:105-113)")
    tensors = feature_tensors_host(packed, actual, prior, fixedArea, taskType, forceScale, displaceScale,
                                   positionScale)
    return graphs_from_tensors(packed, tensors, metapathType, sources)


def dataset_graphs(packed: PackedBatch, fixedMemberType=None, taskType=TaskType.OPTIMIZATION,
                   metapathType=MetapathType.NO_IMPLICIT, forceScale=1., displaceScale=1., positionScale=1.,
                   sources=None, device=None, reorder=False, to_host=False):
    """Dataset samples of a packed batch with everything on the GPU: two solves + feature kernel
    (`feature_tensors_device`), graphs holding slices of the DEVICE tensors (`to_host=True`: one download of
    the float32 tensors first).  Raises `LinAlgError` if any of the solves met a non-positive pivot."""
    if taskType not in (TaskType.OPTIMIZATION, TaskType.REGRESSION):
        raise InvalidTaskTypeError(f"Invalid task type [{taskType}].")
    tensors = feature_tensors_device(packed, fixedMemberType, taskType, forceScale, displaceScale,
                                     positionScale, device, reorder)
    if bool(tensors["info"].any().item()):
        raise np.linalg.LinAlgError("Singular matrix")
    if to_host:
        tensors = {k: (v.cpu() if hasattr(v, "cpu") else v) for k, v in tensors.items()}
    return graphs_from_tensors(packed, tensors, metapathType, sources)


def solve_actual_and_prior(packed: PackedBatch, fixedMemberType=None, device=None, reorder=False,
                           devices=None, pool=None, need_actual=True):
    """The two batched GPU solves behind a dataset: real sections, then every member set to
    `fixedMemberType` (reference `data.py:107-114`).  The two solves differ only in A and E, so the
    geometry is uploaded, reordered (`batch.joint_order`) and bucketed ONCE (`solve_batch(..., sections=[...])`).

    Sharding over several GPUs (one worker process per GPU) happens only when the caller
    asks for it: `pool` = a running `shard.ShardedSolver` (keep one for a stream of batches) or `devices` = a
    list of more than one device name (one-shot pool, started and stopped inside this call - worth it for
    large batches only).  Otherwise the batch is solved in THIS process on `device` (default: `cuda:LOCAL_RANK`
    under a launcher, else the current device) - a single truss of `TrussHeteroDataCreator` never starts
    worker processes, and the ranks of a `torchrun` job do not fan out over each other's GPUs.
    `need_actual=False` skips the solve with the real sections (the truss is already solved)."""
    from .batch import solve_batch
    sections = ([None] if need_actual else []) + \
               ([(fixedMemberType.a, fixedMemberType.e, fixedMemberType.density)]
                if fixedMemberType is not None else [])
    if not sections:
        return None, None
    if pool is not None:
        out = pool.solve(packed, reorder=reorder, sections=sections)
    elif devices is not None and len(devices) > 1:
        from .shard import solve_batch_sharded
        out = solve_batch_sharded(packed, devices, reorder=reorder, sections=sections)
    else:
        if devices:
            device = devices[0]
        if device is None and "LOCAL_RANK" in os.environ:
            import torch
            device = f"cuda:{int(os.environ['LOCAL_RANK']) % max(1, torch.cuda.device_count())}"
        out = solve_batch(packed, device, reorder=reorder, sections=sections)
    actual = out[0] if need_actual else None
    prior = out[-1] if fixedMemberType is not None else None
    return actual, prior


def _dense_from_truss(truss):
    """Dense (u [1,nJ,3], f_ext [1,nJ,3], N [1,nM]) of an already solved truss from its sparse result
    dicts (absent = below 1e-10 = zero, truss.py:344-359)."""
    nJ, nM, dim = truss.nJoint, truss.nMember, truss.dim
    u, f, n = np.zeros([1, nJ, 3]), np.zeros([1, nJ, 3]), np.zeros([1, nM])
    for j, v in truss.GetDisplacements(isProtect=False).items():
        u[0, j, :dim] = v
    for j, v in truss.GetExternalForces(isProtect=False).items():
        f[0, j, :dim] = v
    for m, v in truss.GetInternalForces(isProtect=False).items():
        n[0, m] = v
    return BatchResult(u, f, n, np.zeros([1], dtype=np.int32))


_SIZE_BLOCK = 4096   # polycube sizes are drawn in fixed blocks of global sample indices
#: Definition of "the dataset of (seed, n_samples)".  2 (since round 3): sizes keyed by (seed, global index // 4096),
#: default chunk 32 768 - independent of how the dataset is cut into chunks and ranks.  1 (round 2): sizes keyed by
#: (seed, chunk index) with a default chunk of 16 384 - the same (seed, n_samples) gave OTHER samples.  Pinned by
#: tests/test_data_graph.py::test_dataset_definition_is_pinned; the generators' per-truss streams (keyed by the
#: global index) did not change.
DATASET_VERSION = 2


def dataset_sizes(seed, first, count, numCubeRange):
    """Polycube sizes of the dataset samples first .. first + count - 1: sample i draws its size from a stream
    keyed by (seed, i // 4096), at position i % 4096 - a function of the GLOBAL index only, so the dataset does
    not depend on how it is cut into chunks or ranks (the native generator keys its per-truss stream by the
    global index in the same way)."""
    first, count = int(first), int(count)
    out = np.empty([count], dtype=np.int64)
    lo, hi = int(numCubeRange[0]), int(numCubeRange[1]) + 1
    for blk in range(first // _SIZE_BLOCK, (first + count - 1) // _SIZE_BLOCK + 1 if count else 0):
        block = np.random.default_rng([int(seed), blk]).integers(lo, hi, size=_SIZE_BLOCK)
        a, b = max(first, blk * _SIZE_BLOCK), min(first + count, (blk + 1) * _SIZE_BLOCK)
        out[a - first: b - first] = block[a - blk * _SIZE_BLOCK: b - blk * _SIZE_BLOCK]
    return out


def dataset_chunks(n_samples, rank=0, world=1, chunk=32768, seed=0, numCubeRange=(8, 190), gridRange=(6, 6, 6),
                   fixedMemberType=None, taskType=TaskType.OPTIMIZATION, forceScale=1., displaceScale=1.,
                   positionScale=1., device=None, reorder=True, prefetch=True, generate="device", **generator_args):
    """BASELINE config 5 as a generator: this rank's share of a dataset of `n_samples` random cube trusses,
    chunk by chunk - generation, joint order, both solves and the feature kernel, everything on `device`
    (`generate_cube_batch_device`, `feature_tensors_device`): NO host work per sample, so eight ranks on one
    host do not compete for its CPUs.  Yields `(first_index, packed, tensors)`; `tensors` are the float32
    feature tensors on the device plus `inputs` (the chunk's resident input tensors); `packed` is a
    `batch.BatchSizes` (per-truss counts; `packed.to_packed(tensors["inputs"])` downloads the arrays) - or a full
    `PackedBatch` with `generate="host"`, which builds the chunks with the native host generator (bit for bit the
    same trusses), prefetching chunk k + 1 on a worker thread while the GPU works on chunk k.
    The dataset is DEFINED by (seed, global sample index) - polycube sizes (`dataset_sizes`) and the generators'
    per-truss streams alike -: any split into ranks and chunks produces the same samples (rank r owns the chunks
    r, r + world, ...).  One process per GPU, no communication: run it under `torchrun` with rank / world from
    the environment, or in a loop."""
    from concurrent.futures import ThreadPoolExecutor
    from .batch import joint_order, order_plan
    from .generate import generate_cube_batch, generate_cube_batch_device
    if generate not in ("device", "host"):
        raise ValueError("generate must be 'device' or 'host'")
    n_chunks = (int(n_samples) + chunk - 1) // chunk
    mine = list(range(rank, n_chunks, world))
    span = lambda k: (k * chunk, min(chunk, int(n_samples) - k * chunk))

    if generate == "device":
        for k in mine:
            first, count = span(k)
            sizes = dataset_sizes(seed, first, count, numCubeRange)
            # (padded widths in steps of 8 joints / 64 members: the chunks of a stream then share a few tensor
            # shapes, which the caching allocator re-uses, instead of growing its pools chunk after chunk)
            meta, inputs = generate_cube_batch_device(sizes, gridRange=gridRange, seed=seed, first_index=first,
                                                      device=device, pad_to=(8, 64), **generator_args)
            tensors = feature_tensors_device(meta, fixedMemberType, taskType, forceScale, displaceScale,
                                             positionScale, device, reorder, device_inputs=inputs)
            tensors["inputs"] = inputs
            yield first, meta, tensors
        return

    def host_side(k):   # native code (the GIL is released): generation and the joint order of chunk k
        first, count = span(k)
        sizes = dataset_sizes(seed, first, count, numCubeRange)
        packed = generate_cube_batch(sizes, gridRange=gridRange, seed=seed, first_index=first, **generator_args)
        # the joint order: on the GPU with the solves (`trs_joint_order`) whenever the chunk's shape fits that
        # kernel - nothing to do here then -, otherwise natively on this thread while the GPU works on chunk k - 1
        plan = order_plan(reorder, packed.nJ_max, packed.nM_max)
        return first, packed, (joint_order(packed, plan[1]) if plan is not None and plan[0] == "host" else reorder)

    def device_side(packed, order):
        return feature_tensors_device(packed, fixedMemberType, taskType, forceScale, displaceScale,
                                      positionScale, device, order)

    if not prefetch:
        for k in mine:
            first, packed, order = host_side(k)
            yield first, packed, device_side(packed, order)
        return
    # the host side of chunk k + 1 runs on a worker thread while the GPU works on chunk k
    with ThreadPoolExecutor(max_workers=1) as pool:
        ahead = pool.submit(host_side, mine[0]) if mine else None
        for i in range(len(mine)):
            first, packed, order = ahead.result()
            ahead = pool.submit(host_side, mine[i + 1]) if i + 1 < len(mine) else None
            yield first, packed, device_side(packed, order)


class _PackedFeatureJob:
    """The two solves + the packed feature kernel of one batch, split into SET-UP (everything that copies between
    host and device: the bucket index lists of the `RaggedSolver`, the row offsets) and RUN (kernel launches only).
    `dataset_stream` sets a chunk up BEFORE it queues the previous chunk's large device -> host copy: small copies
    of either direction submitted behind a 2 GB copy wait for all of it, and the chunk's device work with them."""

    def __init__(self, packed, fixedMemberType, taskType, forceScale, displaceScale, positionScale, device, reorder,
                 device_inputs):
        import torch
        from .batch import DeviceBatch, RaggedSolver, _require_gpu, shared_workspace
        if (np.asarray(packed.dim) != 3).any():
            raise NotImplementedError("graph features are defined for 3D trusses (utils.py:105-113)")
        torch, dev = _require_gpu(device if device_inputs is None else device_inputs["xyz"].device)
        self.torch, self.dev, self.packed = torch, dev, packed
        self.regression = taskType == TaskType.REGRESSION
        self.fixed = fixedMemberType
        self.scales = (float(forceScale), float(displaceScale), float(positionScale))
        self.sections = [None] + ([(fixedMemberType.a, fixedMemberType.e, fixedMemberType.density)]
                                  if fixedMemberType is not None else [])
        if device_inputs is None:
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            device_inputs = {f: up(getattr(packed, f)) for f in DeviceBatch.INPUT_FIELDS}
        self.inputs = device_inputs
        B = packed.B
        self.solver = RaggedSolver(packed, dev, reorder=reorder, tensors=device_inputs,
                                   workspace=shared_workspace(torch, dev), n_variants=len(self.sections))
        self.FJ, self.FM = _feature_shapes(B, 0, 0, fixedMemberType is not None, self.regression)
        self.joint_off = np.zeros([B + 1], dtype=np.int64)
        self.member_off = np.zeros([B + 1], dtype=np.int64)
        np.cumsum(packed.nJ, out=self.joint_off[1:])
        np.cumsum(packed.nM, out=self.member_off[1:])
        self.offs = torch.from_numpy(np.stack([self.joint_off[:-1], self.member_off[:-1]])).to(dev)
        SJ, SM = int(self.joint_off[-1]), int(self.member_off[-1])
        self.shapes = {"joint_x": ([SJ, self.FJ], torch.float32), "member_x": ([SM, self.FM], torch.float32),
                       "joint_y": ([SJ, 3], torch.float32), "member_y": ([SM, 1], torch.float32),
                       "j2m_joint": ([SM, 2], torch.int32), "weight": ([B], torch.float64)}
        if not self.regression:
            del self.shapes["joint_y"], self.shapes["member_y"]

    def need(self):
        """Elements per flat output buffer (plus `info`: 2 B int32)."""
        out = {name: int(np.prod(shape)) for name, (shape, _) in self.shapes.items()}
        out["info"] = 2 * self.packed.B
        return out

    def run(self, out=None):
        """Launch the solves and the feature kernel on the current stream; `out`: flat device buffers to write
        into (`need()` elements each).  Returns the dict of device tensors."""
        from . import _capi
        torch, dev, packed = self.torch, self.dev, self.packed
        self.solver.step(sections=self.sections)
        actual = self.solver.outs[0]
        prior = self.solver.outs[1] if self.fixed is not None else None
        t = {}
        for name, (shape, dt) in self.shapes.items():
            t[name] = out[name][:int(np.prod(shape))].view(shape) if out is not None else \
                torch.empty(shape, dtype=dt, device=dev)
        inp = self.inputs
        ptr = lambda x: None if x is None else x.data_ptr()
        ua, na = (actual["u"], actual["N"]) if self.regression else (None, None)
        up, npr = (prior["u"], prior["N"]) if prior is not None else (None, None)
        if B := packed.B:
            with torch.cuda.device(dev):
                _capi.check(_capi.load().trs_graph_features_packed(
                    B, packed.nJ_max, packed.nM_max, inp["xyz"].data_ptr(), inp["conn"].data_ptr(),
                    inp["A"].data_ptr(), inp["rho"].data_ptr(), inp["cbits"].data_ptr(), inp["loads"].data_ptr(),
                    inp["nJ"].data_ptr(), inp["nM"].data_ptr(), ptr(ua), ptr(na), ptr(up), ptr(npr),
                    float(self.fixed.a if prior is not None else 1.0), *self.scales, int(self.regression),
                    self.offs[0].data_ptr(), self.offs[1].data_ptr(), ptr(t["joint_x"]), ptr(t["member_x"]),
                    ptr(t.get("joint_y")), ptr(t.get("member_y")), ptr(t["j2m_joint"]), t["weight"].data_ptr(),
                    torch.cuda.current_stream(dev).cuda_stream), "trs_graph_features_packed")
        info = torch.stack([actual["info"], prior["info"] if prior is not None else torch.zeros_like(actual["info"])])
        if out is not None:
            out["info"][:2 * packed.B].view(2, packed.B).copy_(info)
            info = out["info"][:2 * packed.B].view(2, packed.B)
        t["info"] = info
        t.setdefault("joint_y", None)
        t.setdefault("member_y", None)
        return t


def feature_tensors_packed(packed, fixedMemberType, taskType, forceScale=1., displaceScale=1., positionScale=1.,
                           device=None, reorder=False, device_inputs=None, out=None):
    """`feature_tensors_device` with the PACKED output of `trs_graph_features_packed`: the joint rows of all trusses
    back to back (`joint_x` [sum nJ, FJ], `joint_y` [sum nJ, 3]), the member rows likewise (`member_x`, `member_y`),
    `j2m_joint` [sum nM, 2] int32 (the members' end joints = row 0 of every sample's `j2m` edge index), `weight`
    [B], `info` [2, B] - on the device, nothing padded: what a `HeteroData` of every sample slices and what the
    dataset stream sends across PCIe.  `out`: a dict of flat device buffers to write into.  Returns (tensors,
    joint offsets, member offsets) - the offsets as host int64 arrays [B + 1]."""
    job = _PackedFeatureJob(packed, fixedMemberType, taskType, forceScale, displaceScale, positionScale, device,
                            reorder, device_inputs)
    return job.run(out), job.joint_off, job.member_off


class PackedGraphs:
    """The graphs of a chunk of samples over PACKED feature tensors (host or device): `len`, indexing, slicing,
    iteration; a graph object (`HeteroData` when torch_geometric is importable, else `GraphStores`) is built when
    it is asked for and holds SLICES of the chunk's tensors (`joint.x`, `member.x`, `y`) plus its edge indices
    (reference data.py:238-282: `j2m` = [end joints of member 0, 0, 1, 1, ...; 0, 0, 1, 1, ...], `m2j` the rows
    swapped; `j2j` / `m2m` with `MetapathType.USE_IMPLICIT`).  `first` = global index of the chunk's first sample."""

    def __init__(self, first, nJ, nM, joint_off, member_off, tensors, metapathType=MetapathType.NO_IMPLICIT,
                 sources=None):
        self.first, self.nJ, self.nM = int(first), np.asarray(nJ), np.asarray(nM)
        self.joint_off, self.member_off = joint_off, member_off
        self.tensors, self.metapathType, self.sources = tensors, metapathType, sources

    def __len__(self):
        return len(self.nJ)

    def __iter__(self):
        return (self[b] for b in range(len(self)))

    @property
    def nbytes(self):
        """Bytes of the chunk's tensors (what crossed PCIe for a host chunk)."""
        return sum(int(v.numel() * v.element_size()) for v in self.tensors.values() if v is not None)

    def __getitem__(self, b):
        import torch
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        if b < 0:
            b += len(self)
        if not 0 <= b < len(self):
            raise IndexError(b)
        t = self.tensors
        j0, j1 = int(self.joint_off[b]), int(self.joint_off[b + 1])
        m0, m1 = int(self.member_off[b]), int(self.member_off[b + 1])
        g = _new_graph()
        g["src"] = None if self.sources is None else self.sources[b]
        g["originWeight"] = float(t["weight"][b])
        g["joint"].x = t["joint_x"][j0:j1]
        g["member"].x = t["member_x"][m0:m1]
        if t.get("joint_y") is not None:
            g["joint"].y = t["joint_y"][j0:j1]
            g["member"].y = t["member_y"][m0:m1]
        ends = t["j2m_joint"][m0:m1]
        members = torch.arange(m1 - m0, device=ends.device).repeat_interleave(2)
        j2m = torch.stack([ends.reshape(-1).long(), members])
        g["joint", "j2m", "member"].edge_index = j2m
        g["member", "m2j", "joint"].edge_index = torch.flip(j2m, dims=[0])
        if self.metapathType == MetapathType.USE_IMPLICIT:
            jj, mm = _implicit_edges(ends.cpu().numpy().astype(np.int64), j1 - j0, m1 - m0)
            g["joint", "j2j", "joint"].edge_index = torch.from_numpy(jj).to(ends.device)
            g["member", "m2m", "member"].edge_index = torch.from_numpy(mm).to(ends.device)
        return g


class _PackedRing:
    """`slots` pairs of grow-only flat buffers - device staging + page-locked host memory - for the packed feature
    tensors of the dataset stream, and the events that order their reuse."""
    KINDS = {"joint_x": "float32", "member_x": "float32", "joint_y": "float32", "member_y": "float32",
             "j2m_joint": "int32", "weight": "float64", "info": "int32"}

    def __init__(self, torch, device, slots):
        self.torch, self.device = torch, device
        self.dev = [dict() for _ in range(slots)]
        self.host = [dict() for _ in range(slots)]
        self.done = [None] * slots      # event: the slot's last D2H copies have finished

    def reserve(self, slot, need):
        """Buffers of slot `slot` with at least `need[name]` elements (12 % headroom when one has to grow)."""
        t = self.torch
        for name, count in need.items():
            have = self.dev[slot].get(name)
            if have is None or have.numel() < count:
                if self.done[slot] is not None:
                    self.done[slot].synchronize()    # nothing of the old buffers is in flight any more
                cap = max(1, int(count) + int(count) // 8)
                dt = getattr(t, self.KINDS[name])
                self.dev[slot][name] = t.empty([cap], dtype=dt, device=self.device)
                self.host[slot][name] = t.empty([cap], dtype=dt, pin_memory=True)
        return self.dev[slot], self.host[slot]


_RINGS = {}


def release_stream_buffers():
    """Drop the page-locked rings and staging buffers `dataset_stream` keeps per device."""
    _RINGS.clear()


def dataset_stream(n_samples, rank=0, world=1, chunk=32768, seed=0, numCubeRange=(8, 190), gridRange=(6, 6, 6),
                   fixedMemberType=None, taskType=TaskType.OPTIMIZATION, metapathType=MetapathType.NO_IMPLICIT,
                   forceScale=1., displaceScale=1., positionScale=1., device=None, reorder=True, slots=2,
                   record=None, **generator_args):
    """BASELINE config 5 END TO END: "cube-truss dataset generation ... results streamed to PyG HeteroData".  This
    rank's share of a dataset of `n_samples` random cube trusses, chunk by chunk, DELIVERED ON THE HOST: generation,
    joint order, both solves and the feature kernel on `device` as in `dataset_chunks`, the features written in
    packed form (`trs_graph_features_packed`: no padding, 60-70 KB per sample of 8 .. 190 cubes instead of ~100 KB
    of padded rows) into a device staging buffer and copied by DMA into page-locked host memory on a second
    stream WHILE the device works on the next chunk.  Yields `PackedGraphs` whose tensors are views of the
    page-locked ring: `graphs[i]` is sample `graphs.first + i` as a `HeteroData` (torch_geometric present) or
    `GraphStores`.  A yielded chunk stays valid until the generator has been advanced `slots - 1` more times (and, after
    the generator has ended, until the next stream on this device starts: the rings are reused) -
    a consumer that keeps samples longer copies them (`torch.save`, a collate into its own batch, ...).
    `record` (a list): (name, start event, end event) of every chunk's device work and copy are appended.
    The dataset is defined by (seed, global sample index) exactly as in `dataset_chunks`."""
    import torch
    from .batch import _require_gpu
    from .generate import generate_cube_batch_device
    torch, dev = _require_gpu(device)
    slots = max(2, int(slots))
    n_chunks = (int(n_samples) + chunk - 1) // chunk
    mine = list(range(rank, n_chunks, world))
    # (rings outlive the call: page-locking a few GB costs as much as solving a chunk.  A running stream OWNS its ring -
    # it is taken out of the cache here and put back when the generator ends or is closed -, so two streams on one
    # device never share buffers)
    ring_key = (str(dev), slots)
    ring = _RINGS.pop(ring_key, None) or _PackedRing(torch, dev, slots)
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    regression = taskType == TaskType.REGRESSION
    span = lambda k: (k * chunk, min(chunk, int(n_samples) - k * chunk))

    def plan(k):
        # The generator's size pass + its ONE small device -> host readback.  Placed at the END of a chunk's device
        # work and BEFORE that chunk's large copy is queued: device -> host copies share a DMA queue, and a small
        # readback submitted behind a 2 GB copy waits for all of it - the stream then runs compute and copy one
        # after the other (measured: 104 instead of 64 ms per chunk of 32 768).
        first, count = span(k)
        sizes = dataset_sizes(seed, first, count, numCubeRange)
        return generate_cube_batch_device(sizes, gridRange=gridRange, seed=seed, first_index=first, device=dev,
                                          pad_to=(8, 64), plan_only=True, **generator_args)

    def prepare(k):
        meta, fill = plan(k)
        inputs = fill()
        return meta, _PackedFeatureJob(meta, fixedMemberType, taskType, forceScale, displaceScale, positionScale, dev,
                                       reorder, inputs)

    def launch(i, meta, job):
        """Chunk mine[i]: kernels only, into staging slot i % slots; returns what the copy and the hand-over need."""
        slot = i % slots
        need = job.need()
        dbuf, hbuf = ring.reserve(slot, need)
        if ring.done[slot] is not None:
            main.wait_event(ring.done[slot])      # the slot's previous chunk has left the staging buffer
        if record is not None:
            c0 = torch.cuda.Event(enable_timing=True); c0.record(main)
        job.run(dbuf)
        packed_ev = torch.cuda.Event(enable_timing=record is not None)
        packed_ev.record(main)
        if record is not None:
            record.append((f"chunk {mine[i]} device work", c0, packed_ev))
        return slot, need, dbuf, hbuf, packed_ev, meta, job

    def copy_out(i, slot, need, dbuf, hbuf, packed_ev, meta, job):
        """Queue the chunk's device -> host copies on the side stream; returns (graphs, done event)."""
        host = {}
        with torch.cuda.stream(side):
            side.wait_event(packed_ev)
            if record is not None:
                d0 = torch.cuda.Event(enable_timing=True); d0.record(side)
            for name, cnt in need.items():
                hbuf[name][:cnt].copy_(dbuf[name][:cnt], non_blocking=True)
                host[name] = hbuf[name][:cnt]
            ring.done[slot] = torch.cuda.Event(enable_timing=record is not None)
            ring.done[slot].record(side)
            if record is not None:
                record.append((f"chunk {mine[i]} copy", d0, ring.done[slot]))
        SJ, SM, count = int(job.joint_off[-1]), int(job.member_off[-1]), meta.B
        host = {"joint_x": host["joint_x"].view(SJ, job.FJ), "member_x": host["member_x"].view(SM, job.FM),
                "joint_y": host["joint_y"].view(SJ, 3) if regression else None,
                "member_y": host["member_y"].view(SM, 1) if regression else None,
                "j2m_joint": host["j2m_joint"].view(SM, 2), "weight": host["weight"],
                "info": host["info"].view(2, count)}
        first = span(mine[i])[0]
        return PackedGraphs(first, meta.nJ, meta.nM, job.joint_off, job.member_off, host, metapathType), ring.done[slot]

    # Order of one round (chunk k running on the device):
    #   set up chunk k + 1 (size readback - waits for chunk k's kernels -, index lists, offsets: all the small copies)
    #   -> queue chunk k's large copy -> launch chunk k + 1 (kernels only) -> hand over chunk k - 1
    # so that no small copy is ever submitted while a large one is in flight, and the device always has the next
    # chunk's kernels queued while the host is with the consumer.
    try:
        if not mine:
            return
        running = launch(0, *prepare(mine[0]))
        pending = None
        for i in range(len(mine)):
            nxt = prepare(mine[i + 1]) if i + 1 < len(mine) else None
            copied = copy_out(i, *running)
            if nxt is not None:
                running = launch(i + 1, *nxt)
            if pending is not None:
                pending[1].synchronize()
                yield pending[0]
            pending = copied
        pending[1].synchronize()
        yield pending[0]
    finally:
        for ev in ring.done:          # nothing of this stream is in flight when the ring changes hands
            if ev is not None:
                ev.synchronize()
        _RINGS.setdefault(ring_key, ring)


class TrussHeteroDataCreator:
    """Reference-compatible front end (`data.py:11-44`)."""

    def __init__(self, metapathType=MetapathType.NO_IMPLICIT, taskType=TaskType.OPTIMIZATION):
        self.metapathType, self.taskType = metapathType, taskType
        self.jointIndexToID, self.memberIndexToID, self.source, self.truss = [], [], None, None

    def FromJSON(self, trussJSONFile, trussDim, forceScale=1., displaceScale=1., positionScale=1.,
                 usedMemberTypes=None, fixedMemberType=None, isUseFixed=True, isOutputFile=False):
        truss = Truss(trussDim).LoadFromJSON(trussJSONFile, isOutputFile=isOutputFile)
        return self.FromTruss(truss, forceScale, displaceScale, positionScale, usedMemberTypes,
                              fixedMemberType, isUseFixed, trussSrc=trussJSONFile)

    def FromTruss(self, truss, forceScale=1., displaceScale=1., positionScale=1., usedMemberTypes=None,
                  fixedMemberType=None, isUseFixed=True, trussSrc=None, _results=None):
        """One truss -> graph.  `_results` lets tests inject (actual, prior) dense results; otherwise
        the truss (if unsolved) and its fixed-section copy are solved on the GPU as one batch of two."""
        fixed = (fixedMemberType or MemberType(1., 1e7, 0.1)) if isUseFixed else None
        packed = pack_trusses([truss])
        if _results is not None:
            actual, prior = _results
        else:
            # as the reference (data.py:20-21,34-35): a truss that already carries results (solved, or
            # loaded from an output file) is not solved again; an unstable one raises before any solve
            solved = truss.isSolved
            if not truss.isStable:
                from .utils import TrussNotStableError
                raise TrussNotStableError("The truss is not stable !")
            actual, prior = solve_actual_and_prior(packed, fixed, need_actual=not solved)
            if (actual is not None and actual.info.any()) or (prior is not None and prior.info.any()):
                raise np.linalg.LinAlgError("Singular matrix")
            if solved:
                actual = _dense_from_truss(truss)
            else:
                truss.AdoptDenseResults(actual.displace[0], actual.external[0], actual.internal[0])
        self.truss, self.source = truss, trussSrc
        self.jointIndexToID, self.memberIndexToID = truss.GetJointIDs(), truss.GetMemberIDs()
        dim, nJ, nM = truss.dim, truss.nJoint, truss.nMember
        sections = np.stack([packed.A[0, :nM], packed.E[0, :nM], packed.rho[0, :nM]], axis=1)
        arrays = graph_arrays(
            packed.xyz[0, :nJ, :dim], packed.conn[0, :nM], sections,
            np.array([truss.GetSupportType(j) != SupportType.NO for j in range(nJ)]),
            packed.loads[0, :nJ, :dim], dim,
            (actual.displace[0, :nJ, :dim], actual.internal[0, :nM]),
            None if prior is None else (prior.displace[0, :nJ, :dim], prior.internal[0, :nM], fixed.a),
            self.taskType, self.metapathType, forceScale, displaceScale, positionScale, usedMemberTypes)
        return _to_graph(arrays, truss.weight, trussSrc)

    def AddDenseEdges(self, graphData):
        """Fully connected joint-member (and, with implicit metapaths, joint-joint / member-member)
        edges (`data.py:46-78`)."""
        import torch
        if not self.truss:
            raise RuntimeError("No truss has been assigned.")
        nJ, nM = self.truss.nJoint, self.truss.nMember
        grid = lambda a, b: torch.stack([torch.arange(a).repeat_interleave(b), torch.arange(b).repeat(a)])
        jm = grid(nJ, nM)
        graphData["joint", "jFCm", "member"].edge_index = jm
        graphData["member", "mFCj", "joint"].edge_index = jm.flip(0)
        if self.metapathType == MetapathType.USE_IMPLICIT:
            graphData["joint", "jFCj", "joint"].edge_index = grid(nJ, nJ)
            graphData["member", "mFCm", "member"].edge_index = grid(nM, nM)
        return graphData

    def AddMasterNode(self, graphData, embeddingDim=1, fillValue=1.):
        """One master node linked to every joint and member (`data.py:81-98`)."""
        import torch
        if not self.truss:
            raise RuntimeError("No truss has been assigned.")
        nJ, nM = self.truss.nJoint, self.truss.nMember
        star = lambda n: torch.stack([torch.arange(n), torch.zeros(n, dtype=torch.long)])
        graphData["master"].x = torch.tensor([[fillValue] for _ in range(embeddingDim)])
        graphData["joint", "j2M", "master"].edge_index = star(nJ)
        graphData["master", "M2j", "joint"].edge_index = star(nJ).flip(0)
        graphData["member", "m2M", "master"].edge_index = star(nM)
        graphData["master", "M2m", "member"].edge_index = star(nM).flip(0)
        return graphData
