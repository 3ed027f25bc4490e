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


def _load():
    global _gen
    if _gen is None:
        if not os.path.exists(GENLIB_PATH):
            raise HipExtensionError(f"{GENLIB_PATH} is missing: run __graft_entry__.build()")
        lib = ctypes.CDLL(GENLIB_PATH)
        P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        lib.trs_cubegen.restype = I
        lib.trs_cubegen.argtypes = [I, ctypes.c_uint64, I, I, I, P, I, I, I, D, D, P, I, I, P, I, I, I,
                                    P, P, P, P, P, P, P, P, P, P]
        lib.trs_cubegen_bounds.restype = I
        lib.trs_cubegen_bounds.argtypes = [I, I, I, I, I, P, P]
        _gen = lib
    return _gen


def _type_table(memberTypes):
    return np.ascontiguousarray([t.Serialize() if isinstance(t, MemberType) else list(t)
                                 for t in memberTypes], dtype=np.float64).reshape(-1, 3)


def generate_cube_batch(num_cubes, gridRange=(5, 5, 5), lengthRange=(50, 150),
                        forceRange=((-30000, 30000), (-30000, 30000), (-30000, 30000)),
                        nForceRange=None, method=GenerateMethod.Random, linkType=LinkType.Random,
                        memberTypes=((1., 1e7, 0.1),), isAllowParallel=False, seed=0,
                        return_retries=False):
    """One cube truss per entry of `num_cubes` (polycube sizes), as a `PackedBatch`.

    Arguments as the reference's `GenerateRandomCubeTrusses` (`generate.py:314-316`); pins on the
    lowest layer are always added (the reference's default `isAddPinSupport=True`)."""
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
                             int(linkType), int(isAllowParallel), float(lengthRange[0]),
                             float(lengthRange[1]), ptr(frange), lo, hi, ptr(table), len(table), nJ_max,
                             nM_max, ptr(xyz), ptr(conn), ptr(E), ptr(A), ptr(rho), ptr(cbits), ptr(loads),
                             ptr(nJ), ptr(nM), ctypes.cast(ctypes.byref(retries), ctypes.c_void_p))
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


_SUPPORT_3D = {0: "NO", 7: "PIN", 1: "ROLLER_X", 2: "ROLLER_Y", 4: "ROLLER_Z"}
_SUPPORT_2D = {0: "NO", 3: "PIN", 1: "ROLLER_X", 2: "ROLLER_Y"}


def packed_to_json(packed: PackedBatch, b):
    """Truss `b` of a packed batch as the reference's JSON dict (`detail/combine_with_JSON.md:71-163`)."""
    dim, nJ, nM = int(packed.dim[b]), int(packed.nJ[b]), int(packed.nM[b])
    names = _SUPPORT_3D if dim == 3 else _SUPPORT_2D
    mask = 7 if dim == 3 else 3          # a 2D truss carries an extra z bit in the packed form
    return {
        "joint": [[packed.xyz[b, j, :dim].tolist(), names[int(packed.cbits[b, j]) & mask]] for j in range(nJ)],
        "force": [[j, packed.loads[b, j, :dim].tolist()] for j in range(nJ)
                  if np.any(np.abs(packed.loads[b, j, :dim]) >= 1e-10)],
        "member": [[packed.conn[b, m].tolist(), [float(packed.A[b, m]), float(packed.E[b, m]),
                                                 float(packed.rho[b, m])]] for m in range(nM)],
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
    if not isAddPinSupport:
        raise NotImplementedError("trusses without supports cannot be analysed; isAddPinSupport must be True")
    cases = [(num, i) for num in range(numCubeRange[0], numCubeRange[1] + 1)
             for i in range(numEachRange[0], numEachRange[1] + 1)]
    packed = generate_cube_batch([num for num, _ in cases], gridRange, lengthRange, forceRange, nForceRange,
                                 method, linkType, memberTypes, isAllowParallel,
                                 seed=0 if seed is None else seed)
    trusses = []
    for b in range(packed.B):
        data = packed_to_json(packed, b)
        if augmenter is not None:
            data = augmenter(data)
        trusses.append(Truss(3).LoadFromJSON(data=data))
    if isDoStructuralAnalysis:
        from .batch import solve_batch
        res = solve_batch(trusses)
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
