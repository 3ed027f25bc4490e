"""Random cube-truss generation (BASELINE configs 3 and 5).

`generate_cube_batch` drives the native generator (`csrc/cubegen.c`, host-side C) and returns a
`PackedBatch` directly - no Python objects per truss - so that feeding the GPU solver is not bound
by the generator (the reference's `GenerateRandomCubeTrusses`, `slientruss3d/generate.py:314-376`,
costs 1-50 ms per truss in Python).  Same construction and output schema as the reference, own RNG.

`GenerateRandomCubeTrusses` keeps the reference's signature and returns `list[Truss]`; with
`isDoStructuralAnalysis=True` all generated trusses are solved in one batched GPU call.
"""
import ctypes
import os

import numpy as np

from .batch import PackedBatch, count_free
from .truss import Truss
from .type import GenerateMethod, LinkType, MemberType
from .utils import HipExtensionError

_HERE = os.path.dirname(os.path.abspath(__file__))
GENLIB_PATH = os.path.join(_HERE, "libtrs_host.so")
_gen = None


def available_cpus():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup CPU quota (v2
    `cpu.max`, v1 `cpu.cfs_quota_us`) when the container has one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = int(fq.read()), int(fp.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


_thread_share = None   # number of processes that share this host's CPUs with this one (set_host_thread_share)


def host_thread_budget(sharers=None):
    """Threads the native host helpers of THIS process may use: the CPUs the container really has
    (`available_cpus`) divided by the number of processes that work side by side on this host - the ranks of a
    `torchrun` launch (`LOCAL_WORLD_SIZE`) or the workers of a `shard.ShardedSolver`.  Eight ranks that each
    start a team of every CPU oversubscribe the host eight times exactly where the ragged workloads are
    host-bound (the joint order, the generator)."""
    if sharers is None:
        sharers = _thread_share
    if sharers is None:
        try:
            sharers = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        except ValueError:
            sharers = 1
    return max(1, available_cpus() // max(1, int(sharers)))


def set_host_thread_share(sharers):
    """Declare that `sharers` processes share this host (a rank of an N-process job, a worker of a pool of
    N): the OpenMP teams of the native helpers are sized to `host_thread_budget()` from now on
    (`OMP_NUM_THREADS` in the environment still wins).  Returns the budget."""
    global _thread_share
    _thread_share = max(1, int(sharers))
    if _gen is not None and "OMP_NUM_THREADS" not in os.environ:
        _gen.trs_host_threads(host_thread_budget())
    return host_thread_budget()


def host_threads():
    """Size of the OpenMP team the native helpers use right now."""
    return int(_load().trs_host_threads(0))


def _load():
    global _gen
    if _gen is None:
        if not os.path.exists(GENLIB_PATH):
            raise HipExtensionError(f"{GENLIB_PATH} is missing: run __graft_entry__.build()")
        lib = ctypes.CDLL(GENLIB_PATH)
        P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        lib.trs_cubegen.restype = I
        lib.trs_cubegen.argtypes = [I, ctypes.c_uint64, I, I, I, P, I, I, I, D, D, P, I, I, P, I, I, I,
                                    P, P, P, P, P, P, P, P, P, P, ctypes.c_int64]
        lib.trs_cubegen_bounds.restype = I
        lib.trs_cubegen_bounds.argtypes = [I, I, I, I, I, P, P]
        lib.trs_host_threads.restype = I
        lib.trs_host_threads.argtypes = [I]
        if "OMP_NUM_THREADS" not in os.environ:   # a team of every logical CPU is throttled under a CPU quota
            lib.trs_host_threads(host_thread_budget())
        _gen = lib
    return _gen


def _fresh_seed():
    """`seed=None` in the reference means "do not re-seed": every call continues the ambient `random`
    stream and returns fresh trusses (`generate.py:338-339`).  Same here: the 64-bit seed of the native
    generator is drawn from Python's `random`, so `random.seed(s)` before the call makes it repeatable
    and repeated unseeded calls differ."""
    import random
    return random.getrandbits(64)


def _type_table(memberTypes):
    return np.ascontiguousarray([t.Serialize() if isinstance(t, MemberType) else list(t)
                                 for t in memberTypes], dtype=np.float64).reshape(-1, 3)


def generate_cube_batch(num_cubes, gridRange=(5, 5, 5), lengthRange=(50, 150),
                        forceRange=((-30000, 30000), (-30000, 30000), (-30000, 30000)),
                        nForceRange=None, method=GenerateMethod.Random, linkType=LinkType.Random,
                        memberTypes=((1., 1e7, 0.1),), isAllowParallel=False, seed=0,
                        return_retries=False, isAddPinSupport=True, first_index=0):
    """One cube truss per entry of `num_cubes` (polycube sizes), as a `PackedBatch`.

    Arguments as the reference's `GenerateRandomCubeTrusses` (`generate.py:314-316`).  With
    `isAddPinSupport=False` no joint is supported (an augmenter is expected to add supports).
    Truss b draws from a stream keyed by (seed, first_index + b): generating a dataset in chunks or
    shards (`first_index` = global index of the chunk's first truss) gives the same trusses."""
    lib = _load()
    num_cubes = np.ascontiguousarray(num_cubes, dtype=np.int32).ravel()
    B = len(num_cubes)
    gx, gy, gz = (int(v) for v in gridRange)
    nJ_max, nM_max = ctypes.c_int(), ctypes.c_int()
    lib.trs_cubegen_bounds(gx, gy, gz, int(num_cubes.max(initial=1)), int(isAllowParallel),
                           ctypes.byref(nJ_max), ctypes.byref(nM_max))
    nJ_bound, nM_bound = nJ_max.value, nM_max.value
    nJ = np.empty([B], dtype=np.int32); nM = np.empty([B], dtype=np.int32)
    table = _type_table(memberTypes)
    frange = np.ascontiguousarray(forceRange, dtype=np.float64).reshape(3, 2)
    lo, hi = (-1, -1) if nForceRange is None else tuple(-1 if v is None else int(v) for v in nForceRange)
    retries = ctypes.c_int64(0)
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    def run(nJ_max, nM_max, xyz, conn, E, A, rho, cbits, loads):
        rc = lib.trs_cubegen(B, int(seed) & (2 ** 64 - 1), gx, gy, gz, ptr(num_cubes), int(method),
                             int(linkType), int(bool(isAllowParallel)) | (0 if isAddPinSupport else 2),
                             float(lengthRange[0]),
                             float(lengthRange[1]), ptr(frange), lo, hi, ptr(table), len(table), nJ_max,
                             nM_max, ptr(xyz), ptr(conn), ptr(E), ptr(A), ptr(rho), ptr(cbits), ptr(loads),
                             ptr(nJ), ptr(nM), ctypes.cast(ctypes.byref(retries), ctypes.c_void_p),
                             int(first_index))
        if rc != 0:
            raise RuntimeError(f"trs_cubegen failed ({rc})")

    # pass 1: sizes only (same per-truss RNG streams) -> exact padding; pass 2: the batch itself
    run(nJ_bound, nM_bound, None, None, None, None, None, None, None)
    jm, mm = max(1, int(nJ.max(initial=1))), max(1, int(nM.max(initial=1)))
    xyz = np.empty([B, jm, 3]); loads = np.empty([B, jm, 3])
    conn = np.empty([B, mm, 2], dtype=np.int32)
    E = np.empty([B, mm]); A = np.empty([B, mm]); rho = np.empty([B, mm])
    cbits = np.empty([B, jm], dtype=np.uint8)
    run(jm, mm, xyz, conn, E, A, rho, cbits, loads)
    packed = PackedBatch(xyz, conn, E, A, rho, cbits, loads, nJ, nM, np.full([B], 3, dtype=np.int32),
                         count_free(cbits, nJ))
    return (packed, retries.value) if return_retries else packed


def generate_cube_batch_device(num_cubes, gridRange=(5, 5, 5), lengthRange=(50, 150),
                               forceRange=((-30000, 30000), (-30000, 30000), (-30000, 30000)),
                               nForceRange=None, method=GenerateMethod.Random, linkType=LinkType.Random,
                               memberTypes=((1., 1e7, 0.1),), isAllowParallel=False, seed=0,
                               isAddPinSupport=True, first_index=0, device=None, return_retries=False,
                               pad_to=(1, 1), plan_only=False):
    """`generate_cube_batch` ON THE GPU (`trs_cubegen_dev`, `csrc/cubegen.hip`): the same trusses, bit for bit, as
    the host generator gives for the same arguments (same per-truss streams keyed by (seed, first_index + b)),
    written straight into device tensors - nothing of the batch ever exists on the host.

    Returns `(sizes, tensors)`: `sizes` = `batch.BatchSizes` (nJ, nM, n_free per truss on the host: they decide
    buckets and slab shapes), `tensors` = dict of device tensors by `DeviceBatch.INPUT_FIELDS` name, padded to
    the batch's own maxima.  Two passes like the host generator (sizes only, then the arrays); one
    synchronisation in between for the maxima.  `pad_to=(j, m)`: the padded widths are rounded up to multiples of
    j joints / m members (a stream of chunks then repeats a few tensor shapes instead of one per chunk, which
    the caching allocator can reuse; the extra padding is inert).  `plan_only=True` returns `(sizes, fill)` after
    the size pass: `fill()` launches the second pass and returns the tensors - a pipeline can then place the
    generator's one device-to-host readback where it does not queue behind a large copy (`data.dataset_stream`)."""
    import torch
    from . import _capi
    from .batch import BatchSizes, _require_gpu
    torch, dev = _require_gpu(device)
    lib = _capi.load()
    num_cubes = np.ascontiguousarray(num_cubes, dtype=np.int32).ravel()
    B = len(num_cubes)
    gx, gy, gz = (int(v) for v in gridRange)
    nJ_b, nM_b = ctypes.c_int(), ctypes.c_int()
    _load().trs_cubegen_bounds(gx, gy, gz, int(num_cubes.max(initial=1)), int(isAllowParallel),
                               ctypes.byref(nJ_b), ctypes.byref(nM_b))
    frange = np.ascontiguousarray(forceRange, dtype=np.float64).reshape(3, 2)
    lo, hi = (-1, -1) if nForceRange is None else tuple(-1 if v is None else int(v) for v in nForceRange)
    d_cubes = torch.from_numpy(num_cubes).to(dev)
    d_types = torch.from_numpy(_type_table(memberTypes)).to(dev)
    i32 = lambda n: torch.empty([n], dtype=torch.int32, device=dev)
    nJ, nM, n_free, status = i32(B), i32(B), i32(B), torch.zeros([2], dtype=torch.int32, device=dev)
    flags = int(bool(isAllowParallel)) | (0 if isAddPinSupport else 2)

    def run(nJ_max, nM_max, t):
        ptr = lambda k: t[k].data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _capi.check(lib.trs_cubegen_dev(
                B, int(seed) & (2 ** 64 - 1), gx, gy, gz, d_cubes.data_ptr(), int(method), int(linkType), flags,
                float(lengthRange[0]), float(lengthRange[1]), frange.ctypes.data_as(ctypes.c_void_p), lo, hi,
                d_types.data_ptr(), int(d_types.shape[0]), nJ_max, nM_max, ptr("xyz"), ptr("conn"), ptr("E"), ptr("A"),
                ptr("rho"), ptr("cbits"), ptr("loads"), nJ.data_ptr(), nM.data_ptr(), n_free.data_ptr(),
                status.data_ptr(), int(first_index), torch.cuda.current_stream(dev).cuda_stream), "trs_cubegen_dev")

    run(nJ_b.value, nM_b.value, None)                      # pass 1: sizes only -> exact padding
    # (ONE small device -> host copy: the four arrays side by side)
    h_all = torch.cat([nJ, nM, n_free, status]).cpu().numpy()
    h_nJ, h_nM, h_free, h_status = h_all[:B].copy(), h_all[B:2 * B].copy(), h_all[2 * B:3 * B].copy(), h_all[3 * B:]
    if B and h_status[1]:
        raise RuntimeError("trs_cubegen_dev: a truss does not fit the grid's bounds")
    retries = int(h_status[0])
    up_to = lambda v, q: (max(1, int(v)) + int(q) - 1) // int(q) * int(q)
    jm, mm = up_to(h_nJ.max(initial=1), pad_to[0]), up_to(h_nM.max(initial=1), pad_to[1])
    sizes = BatchSizes(h_nJ, h_nM, h_free, jm, mm)

    def fill():
        """Pass 2: the batch itself (same per-truss streams), asynchronous on the current stream."""
        f64 = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        tensors = {"xyz": f64(B, jm, 3), "loads": f64(B, jm, 3), "cbits": torch.empty([B, jm], dtype=torch.uint8, device=dev),
                   "conn": torch.empty([B, mm, 2], dtype=torch.int32, device=dev), "E": f64(B, mm), "A": f64(B, mm),
                   "rho": f64(B, mm), "nJ": nJ, "nM": nM}
        status.zero_()
        if B:
            run(jm, mm, tensors)
        return tensors

    if plan_only:   # (sizes now - the one host synchronisation of the generator -, arrays when `fill()` is called)
        return (sizes, fill, retries) if return_retries else (sizes, fill)
    tensors = fill()
    return (sizes, tensors, retries) if return_retries else (sizes, tensors)


_SUPPORT_3D = {0: "NO", 7: "PIN", 1: "ROLLER_X", 2: "ROLLER_Y", 4: "ROLLER_Z"}
_SUPPORT_2D = {0: "NO", 3: "PIN", 1: "ROLLER_X", 2: "ROLLER_Y"}


def packed_to_json(packed: PackedBatch, b):
    """Truss `b` of a packed batch as the reference's JSON dict (`detail/combine_with_JSON.md:71-163`)."""
    dim, nJ, nM = int(packed.dim[b]), int(packed.nJ[b]), int(packed.nM[b])
    names = _SUPPORT_3D if dim == 3 else _SUPPORT_2D
    mask = 7 if dim == 3 else 3          # a 2D truss carries an extra z bit in the packed form
    xyz, loads = packed.xyz[b, :nJ, :dim].tolist(), packed.loads[b, :nJ, :dim]
    loaded = np.flatnonzero((np.abs(loads) >= 1e-10).any(axis=1))
    return {
        "joint": [[p, names[c & mask]] for p, c in zip(xyz, packed.cbits[b, :nJ].tolist())],
        "force": [[int(j), loads[j].tolist()] for j in loaded],
        "member": [[ends, [a, e, rho]] for ends, a, e, rho in zip(
            packed.conn[b, :nM].tolist(), packed.A[b, :nM].tolist(), packed.E[b, :nM].tolist(),
            packed.rho[b, :nM].tolist())],
    }


def GenerateRandomCubeTrusses(gridRange=(5, 5, 5), numCubeRange=(5, 5), numEachRange=(1, 10),
                              lengthRange=(50, 150),
                              forceRange=((-30000, 30000), (-30000, 30000), (-30000, 30000)),
                              nForceRange=None, method=GenerateMethod.Random, linkType=LinkType.Random,
                              memberTypes=((1., 1e7, 0.1),), isAddPinSupport=True, isAllowParallel=False,
                              isDoStructuralAnalysis=False, isPlotTruss=False, isPrintMessage=True,
                              saveFolder=None, augmenter=None, seed=None):
    """Reference-compatible front end (`generate.py:314-376`): `list[Truss]`, one per
    (numCube, case) pair, optionally solved (one batched GPU call) and dumped as
    `cube-{numCube}_case_{i}.json`.  Plotting is out of scope (`isPlotTruss` must stay False)."""
    if isPlotTruss:
        raise NotImplementedError("plotting is outside the solver path (SURVEY.md section 2, row 12)")
    if augmenter is None:
        augmenter = NoChange()
    cases = [(num, i) for num in range(numCubeRange[0], numCubeRange[1] + 1)
             for i in range(numEachRange[0], numEachRange[1] + 1)]
    if seed is not None:   # as the reference (generate.py:338-339): the augmenters draw from `random`
        import random
        random.seed(seed)
    seed = _fresh_seed() if seed is None else int(seed)
    gen = lambda nums, s: generate_cube_batch(nums, gridRange, lengthRange, forceRange, nForceRange, method,
                                              linkType, memberTypes, isAllowParallel, seed=s,
                                              isAddPinSupport=isAddPinSupport)
    packed = gen([num for num, _ in cases], seed)
    trusses = []
    for b, (num, _) in enumerate(cases):
        truss = Truss(3).LoadFromJSON(data=augmenter(packed_to_json(packed, b)))
        attempt = 0
        while not truss.isStable:   # an augmenter may leave too few supports: draw again (generate.py:353-374)
            attempt += 1
            if attempt > 1000:
                from .utils import TrussNotStableError
                raise TrussNotStableError("no stable cube truss in 1000 draws (supports missing?)")
            if isPrintMessage:
                print("\nTruss is not stable. Re-genrating...\n")
            again = gen([num], (seed + 0x9E3779B97F4A7C15 * (b * 1000 + attempt)) & (2 ** 64 - 1))
            truss = Truss(3).LoadFromJSON(data=augmenter(packed_to_json(again, 0)))
        trusses.append(truss)
    if isDoStructuralAnalysis:
        from .batch import solve_batch
        res = solve_batch(trusses, reorder=True)   # generator order is far from banded
        for b, truss in enumerate(trusses):
            if int(res.info[b]) != 0:
                raise np.linalg.LinAlgError("Singular matrix")
            truss.AdoptDenseResults(res.displace[b], res.external[b], res.internal[b])
    if saveFolder is not None:
        for (num, i), truss in zip(cases, trusses):
            truss.DumpIntoJSON(os.path.join(saveFolder, f"cube-{num}_case_{i}.json"))
    if isPrintMessage:
        print(f"generated {len(trusses)} cube trusses")
    return trusses


# ---- the reference's object-level generator API (generate.py:150-311) ----------------------------------
# `CubeGrid` / `CubeTruss` for callers who build polycubes cube by cube in Python.  Same constructor arguments,
# methods, return values and `random` call order as the reference, so a seeded script carries over unchanged
# (tests/test_generate.py compares against structures captured from the reference).  The batch paths above
# (`generate_cube_batch`, `generate_cube_batch_device`) do not go through these classes.

_CUBE_FACE_DIAGONALS = (((0, 5), (1, 4)), ((1, 7), (3, 5)), ((3, 6), (2, 7)), ((2, 4), (0, 6)), ((4, 7), (5, 6)),
                        ((0, 3), (1, 2)))
_CUBE_EDGES = ((4, 5), (5, 7), (6, 7), (4, 6), (0, 1), (0, 2), (1, 3), (2, 3), (0, 4), (1, 5), (2, 6), (3, 7))


class CubeTruss:
    """One grid cell as eight joint ids: vertex v sits at `coordinate + ((v & 1), (v >> 1) & 1, (v >> 2) & 1)`;
    vertices shared with earlier cubes keep their ids (`usedDict`: vertex -> joint id)."""

    def __init__(self, coordinate, usedDict=None):
        self._coordinate = tuple(coordinate)
        self.jointIDs = [None] * 8
        self.GenerateNew({} if usedDict is None else usedDict)

    def __repr__(self):
        return str(self.jointIDs)

    def __getitem__(self, i):
        return self.jointIDs[i]

    def __setitem__(self, i, value):
        self.jointIDs[i] = value

    def GetCubeVertices(self):
        return [tuple(c + ((v >> axis) & 1) for axis, c in enumerate(self._coordinate))
                for v in range(1 << len(self._coordinate))]

    def GenerateNew(self, usedDict=None):
        usedDict = {} if usedDict is None else usedDict
        nextID = max(usedDict.values()) + 1 if usedDict else 0
        for v, vertex in enumerate(self.GetCubeVertices()):
            if vertex not in usedDict:
                usedDict[vertex] = nextID
                nextID += 1
            self[v] = usedDict[vertex]

    def LinkMember(self, linkType, hasLinked):
        """The cube's members as joint-id pairs: per face one diagonal, the other or both (`linkType`; a fresh
        `random.sample` per face when `LinkType.Random`), then the twelve edges; with a set `hasLinked` an ordered
        pair already in it is skipped and new ones are recorded there."""
        import random
        links = []

        def add(pair):
            link = [self[pair[0]], self[pair[1]]]
            if hasLinked is None:
                links.append(link)
            elif tuple(link) not in hasLinked:
                hasLinked.add(tuple(link))
                links.append(link)

        for first, second in _CUBE_FACE_DIAGONALS:
            choice = random.sample(range(3), k=1)[0] if linkType == LinkType.Random else linkType
            for pair in ((first,), (second,), (first, second))[choice]:
                add(pair)
        for pair in _CUBE_EDGES:
            add(pair)
        return links


class CubeGrid:
    """A box of `xMax * yMax * zMax` cells in which polycubes are grown."""

    _DIRECTIONS = ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1))

    def __init__(self, xMax, yMax, zMax):
        self._shape = (xMax, yMax, zMax)
        self._usedDict = {}
        self.grid = [[[False] * zMax for _ in range(yMax)] for _ in range(xMax)]

    def __getitem__(self, coordinate):
        return self.grid[coordinate[0]][coordinate[1]][coordinate[2]]

    def __setitem__(self, coordinate, isUsed):
        self.grid[coordinate[0]][coordinate[1]][coordinate[2]] = isUsed

    def IsOutOfRange(self, coordinate):
        return any(c < 0 or c >= top for c, top in zip(coordinate, self._shape))

    def GetRandomFeasible(self):
        import random
        xMax, yMax, zMax = self._shape
        return random.choice([(x, y, z) for z in range(zMax) for y in range(yMax) for x in range(xMax)
                              if not self[(x, y, z)]])

    def GetNextFeasibles(self, coordinate, isSuffle=True):
        import random
        around = [tuple(c + d for c, d in zip(coordinate, step)) for step in self._DIRECTIONS]
        free = [c for c in around if not self.IsOutOfRange(c) and not self[c]]
        if isSuffle:
            random.shuffle(free)
        return free

    def RandomGenerateCubes(self, numCube=None, method=GenerateMethod.DFS):
        import random
        xMax, yMax, zMax = self._shape
        if numCube is None:
            numCube = random.randint(1, xMax * yMax * zMax)
        self._usedDict.clear()
        cubes, pending = [], [self.GetRandomFeasible()]
        while len(cubes) < numCube and pending:
            if method == GenerateMethod.DFS:
                cell = pending.pop()
            elif method == GenerateMethod.BFS:
                cell = pending.pop(0)
            else:
                cell = pending.pop() if random.random() <= 0.5 else pending.pop(0)
            self[cell] = True
            pending.extend(c for c in self.GetNextFeasibles(cell) if c not in pending)
            cubes.append(CubeTruss(cell, self._usedDict))
        return cubes

    def ProcessPinSupport(self, isAddPinSupport, length):
        lowest = min((z for _, _, z in self._usedDict), default=0)
        length = [float(v) for v in length]
        joints = [None] * len(self._usedDict)
        for (x, y, z), jointID in self._usedDict.items():
            joints[jointID] = [[float(x * length[0]), float(y * length[1]), float((z - lowest) * length[2])],
                               "PIN" if (isAddPinSupport and z == lowest) else "NO"]
        return joints

    def CubesToTruss(self, cubes, length, isAddPinSupport=True, isAllowParallel=True, linkType=LinkType.Random,
                     memberType=(1., 1e7, 0.1)):
        memberType = list(memberType)
        hasLinked = None if isAllowParallel else set()
        members = [[link, memberType] for cube in cubes for link in cube.LinkMember(linkType, hasLinked)]
        return {"joint": self.ProcessPinSupport(isAddPinSupport, length), "force": {}, "member": members}


# ---- data augmentation (reference generate.py:13-148) ------------------------------------------------
# Callables on a truss JSON dict (or a `Truss`, which is re-loaded from its augmented serialisation);
# same names, arguments and `random` call order as the reference so that seeded pipelines carry over.

class TrussDataAugmenter:
    """Base: `_apply(data)` edits the JSON dict in place."""

    def _apply(self, data):
        return data

    @staticmethod
    def IsTrussClass(trussData):
        """(is it a `Truss` object, the JSON dict to work on) - `generate.py:14-20`."""
        if isinstance(trussData, Truss):
            return True, trussData.Serialize()
        return False, trussData

    def __call__(self, trussData):
        if isinstance(trussData, Truss):
            # The reference re-loads the augmented serialisation INTO the same object
            # (generate.py:55-56), which appends a second copy of every joint and member; here the
            # object is rebuilt from the augmented data instead (documented in INTEGRATION.md).
            fresh = Truss(trussData.dim).LoadFromJSON(data=self._apply(trussData.Serialize()),
                                                      isOutputFile=trussData.isSolved)
            trussData.__dict__.update(fresh.__dict__)
            return trussData
        return self._apply(trussData)

    @staticmethod
    def GetCentroid(jointList):
        pts = np.array([p for p, _ in jointList], dtype=float).reshape(len(jointList), -1)
        return pts.mean(axis=0).tolist()

    @staticmethod
    def GetStableMinNumPin(trussData):
        return -((len(trussData["member"]) - 3 * len(trussData["joint"])) // 3)   # ceil((3 nJ - nM) / 3)


class NoChange(TrussDataAugmenter):
    """Leave the truss as it is."""


class AddJointNoise(TrussDataAugmenter):
    """Gaussian noise on every joint coordinate."""

    def __init__(self, noiseMeans=(0., 0., 0.), noiseStds=(1., 1., 1.)):
        self.noiseMeans, self.noiseStds = noiseMeans, noiseStds

    def _apply(self, data):
        import random
        for joint in data["joint"]:
            joint[0] = [joint[0][i] + random.gauss(self.noiseMeans[i], self.noiseStds[i]) for i in range(3)]
        return data


class Translation(TrussDataAugmenter):
    """Shift every joint by `translation`."""

    def __init__(self, translation):
        self.translation = translation

    def _apply(self, data):
        for joint in data["joint"]:
            joint[0] = [joint[0][i] + self.translation[i] for i in range(3)]
        return data


class MoveToCentroid(TrussDataAugmenter):
    """Shift the truss so that the centroid of its joints is the origin."""

    def _apply(self, data):
        centre = self.GetCentroid(data["joint"])
        return Translation([-c for c in centre])._apply(data)


class RandomTranslation(TrussDataAugmenter):
    """Shift by a vector drawn uniformly from `translateRange` per axis."""

    def __init__(self, translateRange=(-1., 1.)):
        self.translateRange = translateRange

    def __call__(self, trussData):
        import random
        return Translation([random.uniform(*self.translateRange) for _ in range(3)])(trussData)


class RandomResetPin(TrussDataAugmenter):
    """Re-draw which joints are pinned: between max(minNumPin, the counting-test minimum) and
    maxNumPinRatio * nJoint (all joints when None) pins at random joints, every other joint free."""

    def __init__(self, minNumPin=3, maxNumPinRatio=None):
        if minNumPin is not None and minNumPin < 3:
            from .utils import PinNotEnoughError
            raise PinNotEnoughError("Number of pins must >= 3.")
        self.minNumPin, self.maxNumPinRatio = minNumPin, maxNumPinRatio

    def _apply(self, data):
        import random
        joints = data["joint"]
        need = self.GetStableMinNumPin(data)
        lo = need if self.minNumPin is None else max(self.minNumPin, need)
        hi = len(joints) if self.maxNumPinRatio is None else int(self.maxNumPinRatio * len(joints))
        count = random.choice(range(lo, hi + 1))
        pinned = set(random.sample(range(len(joints)), k=count))
        for j, joint in enumerate(joints):
            joint[-1] = "PIN" if j in pinned else "NO"
        return data


class TrussDataAugmenterList(TrussDataAugmenter):
    """Apply several augmenters in order."""

    def __init__(self, *augmenters):
        self.augmenters = augmenters

    def __call__(self, trussData):
        for augmenter in self.augmenters:
            trussData = augmenter(trussData)
        return trussData
