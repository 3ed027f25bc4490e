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


def hetero_tensors_batch(packed: PackedBatch, actual: BatchResult, prior: BatchResult, fixedArea,
                         taskType=TaskType.OPTIMIZATION, metapathType=MetapathType.NO_IMPLICIT,
                         forceScale=1., displaceScale=1., positionScale=1., sources=None):
    """Graphs of a whole solved batch: `actual` / `prior` are the dense results of the two batched
    solves (`prior` may be None).  Returns one graph per truss.

    The features of ALL trusses are formed natively (`csrc/graphfeat.c`, OpenMP over the batch; same
    formulas as `graph_arrays`, which stays the single-truss path) straight into float32 batch
    tensors; a truss's graph holds slices of those batch tensors."""
    import ctypes
    import torch
    from .generate import _load
    if taskType not in (TaskType.OPTIMIZATION, TaskType.REGRESSION):
        raise InvalidTaskTypeError(f"Invalid task type [{taskType}].")
    if (np.asarray(packed.dim) != 3).any():
        raise NotImplementedError("graph features are defined for 3D trusses (utils.py:105-113)")
    B, nJm, nMm = packed.B, packed.nJ_max, packed.nM_max
    regression = taskType == TaskType.REGRESSION
    FJ = 7 + (3 if prior is not None else 0)
    FM = 8 + (1 if prior is not None else 0) + (1 if regression else 0)
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
    conn = torch.from_numpy(keep[1].astype(np.int64))
    j2m = torch.stack([conn.reshape(B, -1), torch.arange(nMm).repeat_interleave(2).expand(B, -1)], dim=1)
    m2j = torch.flip(j2m, dims=[1])
    graphs = []
    for b in range(B):
        nJ, nM = int(packed.nJ[b]), int(packed.nM[b])
        g = _new_graph()
        g["src"] = None if sources is None else sources[b]
        g["originWeight"] = float(weight[b])
        g["joint"].x = joint_x[b, :nJ]
        g["member"].x = member_x[b, :nM]
        if joint_y is not None:
            g["joint"].y = joint_y[b, :nJ]
            g["member"].y = member_y[b, :nM]
        g["joint", "j2m", "member"].edge_index = j2m[b, :, :2 * nM]
        g["member", "m2j", "joint"].edge_index = m2j[b, :, :2 * nM]
        if metapathType == MetapathType.USE_IMPLICIT:
            jj, mm = _implicit_edges(packed.conn[b, :nM], nJ, nM)
            g["joint", "j2j", "joint"].edge_index = torch.from_numpy(jj)
            g["member", "m2m", "member"].edge_index = torch.from_numpy(mm)
        graphs.append(g)
    return graphs


def solve_actual_and_prior(packed: PackedBatch, fixedMemberType=None, device=None, reorder=False,
                           devices=None, pool=None, need_actual=True):
    """The two batched GPU solves behind a dataset: real sections, then every member set to
    `fixedMemberType` (reference `data.py:107-114`).  The two solves differ only in A and E, so the
    geometry is uploaded, RCM-reordered and bucketed ONCE (`solve_batch(..., sections=[...])`).

    With more than one GPU (`devices` = list of device names, `pool` = a running
    `shard.ShardedSolver`, or - when neither `device` nor `devices` is given - every visible GPU) the
    batch is sharded over one worker process per GPU.
    `need_actual=False` skips the solve with the real sections (the truss is already solved)."""
    from .batch import solve_batch
    sections = ([None] if need_actual else []) + \
               ([(fixedMemberType.a, fixedMemberType.e, fixedMemberType.density)]
                if fixedMemberType is not None else [])
    if not sections:
        return None, None
    if pool is None and devices is None and device is None:
        from .shard import visible_devices
        seen = visible_devices()
        devices = seen if len(seen) > 1 else None
    if pool is not None:
        out = pool.solve(packed, reorder=reorder, sections=sections)
    elif devices is not None and len(devices) > 1:
        from .shard import solve_batch_sharded
        out = solve_batch_sharded(packed, devices, reorder=reorder, sections=sections)
    else:
        out = solve_batch(packed, devices[0] if devices else device, reorder=reorder, sections=sections)
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
