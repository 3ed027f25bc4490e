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
