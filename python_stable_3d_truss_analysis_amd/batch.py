"""Batched `Truss.Solve()`: packing of many trusses into SoA tensors, the device pipeline
(dofmap -> assemble -> potrf -> potrs -> recover through the C ABI), and the dense results.

This is the layer the reference does not have: its `Solve()` (`slientruss3d/truss.py:329-364`)
handles one truss per call.  `Truss.Solve()` here is `solve_batch([truss])`.

All device work is enqueued on the current PyTorch-ROCm stream; tensors are only the memory
the kernels run on (plumbing).  No CPU fallback: without a GPU or the built library the calls
raise `HipExtensionError`.
"""
import os
from dataclasses import dataclass

import numpy as np

from . import _capi
from .type import SupportType
from .utils import HipExtensionError


@dataclass
class PackedBatch:
    """B trusses as padded host arrays (numpy), ready to be uploaded.

    xyz [B,nJ_max,3] f64 | conn [B,nM_max,2] i32 | E, A, rho [B,nM_max] f64 |
    cbits [B,nJ_max] u8 (constrained-axis bits x=1,y=2,z=4) | loads [B,nJ_max,3] f64 |
    nJ, nM [B] i32 | dim [B] (2 or 3; 2D trusses are embedded with z fixed) | n_free [B] i32.

    TABLE member form (`table()`; ABI 10, `include/trs_solver.h` "Member forms"): the members' sections as one uint8
    `type_idx` [B,nM_max] into a table `types` [T,3] = (a, e, density), T <= 256 - the reference's own description of
    a member, `[[j0, j1], [a, e, density]]` with the triple drawn from a short list (`MemberType`, type.py:5-27) -
    and `conn` as uint16 pairs; E, A, rho are then None.  5 instead of 32 bytes per member in host memory, over PCIe
    and in HBM; the solvers take either form and give the same bits."""
    xyz: np.ndarray
    conn: np.ndarray
    E: np.ndarray
    A: np.ndarray
    rho: np.ndarray
    cbits: np.ndarray
    loads: np.ndarray
    nJ: np.ndarray
    nM: np.ndarray
    dim: np.ndarray
    n_free: np.ndarray
    type_idx: np.ndarray = None
    types: np.ndarray = None

    #: fields that are per batch, not per truss (never indexed, tiled or trimmed)
    _SHARED = ("types",)

    @property
    def B(self):
        return int(self.nJ.shape[0])

    @property
    def nJ_max(self):
        return int(self.xyz.shape[1])

    @property
    def nM_max(self):
        return int(self.conn.shape[1])

    @property
    def n_max(self):
        return int(self.n_free.max()) if self.B else 0

    @property
    def is_table(self):
        return self.type_idx is not None

    def _map(self, fn):
        """A batch whose per-truss arrays are `fn(field name, array)` (absent fields stay absent, the type table is shared)."""
        vals = {}
        for f in self.__dataclass_fields__:
            a = getattr(self, f)
            vals[f] = a if a is None or f in self._SHARED else fn(f, a)
        return PackedBatch(**vals)

    def replicate(self, times):
        """The same trusses `times` times over, as independent problems (no sharing on device)."""
        return self._map(lambda f, a: np.ascontiguousarray(np.tile(a, (times,) + (1,) * (a.ndim - 1))))

    def take(self, index):
        """Sub-batch (rows `index`, any numpy index) - used to shard a batch over ranks."""
        return self._map(lambda f, a: np.ascontiguousarray(a[index]))

    def pinned(self):
        """The same batch in page-locked host memory (one copy; needs the GPU runtime): `solve_batch`
        then uploads it by DMA at the full PCIe rate instead of through the driver's bounce buffers.
        The arrays are numpy views of pinned torch tensors, which they keep alive."""
        import torch

        def pin(f, a):
            a = np.ascontiguousarray(a)
            t = torch.empty(a.shape, dtype=torch.from_numpy(a[:0].copy()).dtype, pin_memory=True)
            v = t.numpy()
            v[...] = a
            return v
        return self._map(pin)

    def trimmed(self):
        """The same batch with the joint / member padding cut to this batch's own maxima."""
        jm = max(1, int(self.nJ.max(initial=1)))
        mm = max(1, int(self.nM.max(initial=1)))
        cut = {"xyz": jm, "loads": jm, "cbits": jm, "conn": mm, "E": mm, "A": mm, "rho": mm, "type_idx": mm}
        return self._map(lambda f, a: np.ascontiguousarray(a[:, :cut[f]]) if f in cut else a)

    def table(self):
        """The same batch in the TABLE member form (see the class): raises ValueError when the members use more than
        256 distinct (a, e, density) triples.  A batch already in that form is returned as it is."""
        if self.is_table:
            return self
        if self.nJ_max > 65535:
            raise ValueError("table member form: end joints are uint16 (at most 65 535 joints per truss)")
        live = np.arange(self.nM_max)[None, :] < np.asarray(self.nM)[:, None]
        # distinct BIT patterns (the table must give back exactly the doubles the batch holds), found without sorting
        # tens of millions of members: a 64-bit hash per member, the distinct hashes block by block, the hash's index in
        # their sorted list by binary search, and then the check that every member's triple IS its table row's
        bits = [np.ascontiguousarray(a, dtype=np.float64).view(np.uint64) for a in (self.A, self.E, self.rho)]
        with np.errstate(over="ignore"):
            h = (bits[0] * np.uint64(0x9E3779B97F4A7C15)) ^ (bits[1] * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(0x165667B19E3779F9)) \
                ^ (bits[2] * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x27D4EB2F165667C5))
        h = np.where(live, h, h[0, 0] if h.size else 0).ravel()    # (padding members take the first member's type)
        seen = np.zeros([0], dtype=np.uint64)
        for k in range(0, h.size, 1 << 22):
            seen = np.union1d(seen, np.unique(h[k: k + (1 << 22)]))
            if len(seen) > 256:
                raise ValueError(f"table member form: more than 256 distinct (a, e, density) triples")
        idx = np.searchsorted(seen, h)
        rep = np.zeros([max(1, len(seen))], dtype=np.int64)
        rep[idx[::-1]] = np.arange(h.size - 1, -1, -1)                 # first member of every hash value
        flat = [b.ravel() for b in bits]
        types_bits = np.stack([f[rep] for f in flat], axis=-1)
        if not all(np.array_equal(np.where(live.ravel(), flat[c], types_bits[idx, c]), types_bits[idx, c]) for c in range(3)):
            raise ValueError("table member form: hash collision between member types (use the general form)")
        order = np.lexsort((types_bits[:, 2].view(np.float64), types_bits[:, 1].view(np.float64), types_bits[:, 0].view(np.float64)))
        rank = np.empty_like(order)
        rank[order] = np.arange(len(order))
        types = np.ascontiguousarray(types_bits[order].view(np.float64)) if h.size else np.zeros([1, 3])
        type_idx = (rank[idx].astype(np.uint8).reshape(self.B, self.nM_max) * live).astype(np.uint8)
        vals = {f: getattr(self, f) for f in self.__dataclass_fields__}
        vals.update(conn=np.ascontiguousarray(self.conn, dtype=np.uint16), E=None, A=None, rho=None, type_idx=type_idx,
                    types=types)
        return PackedBatch(**vals)

    def general(self):
        """The same batch in the GENERAL member form (int32 end joints, E / A / rho per member; padding members get
        the sections of type 0, which no kernel reads)."""
        if not self.is_table:
            return self
        t = np.asarray(self.types, dtype=np.float64)[self.type_idx.astype(np.int64)]
        vals = {f: getattr(self, f) for f in self.__dataclass_fields__}
        vals.update(conn=np.ascontiguousarray(self.conn, dtype=np.int32), A=np.ascontiguousarray(t[..., 0]),
                    E=np.ascontiguousarray(t[..., 1]), rho=np.ascontiguousarray(t[..., 2]), type_idx=None, types=None)
        return PackedBatch(**vals)


@dataclass
class BatchSizes:
    """What the host needs to know about a batch whose arrays live on the DEVICE (`generate.generate_cube_batch_device`):
    the per-truss counts (they decide the size buckets and the slab shapes) and the padded widths.  Accepted
    wherever a `PackedBatch` is only asked for its sizes (`RaggedSolver(..., tensors=)`, `solve_batch(...,
    device_inputs=)`, `size_buckets`); `to_packed(tensors)` downloads the arrays."""
    nJ: np.ndarray
    nM: np.ndarray
    n_free: np.ndarray
    nJ_max: int
    nM_max: int

    @property
    def B(self):
        return int(self.nJ.shape[0])

    @property
    def n_max(self):
        return int(self.n_free.max()) if self.B else 0

    @property
    def dim(self):
        return np.full([self.B], 3, dtype=np.int32)

    def to_packed(self, tensors):
        """The same batch as host arrays (`PackedBatch`), downloaded from its device tensors."""
        host = {f: tensors[f].cpu().numpy() for f in ("xyz", "conn", "E", "A", "rho", "cbits", "loads")}
        return PackedBatch(host["xyz"], host["conn"], host["E"], host["A"], host["rho"], host["cbits"], host["loads"],
                           self.nJ.copy(), self.nM.copy(), self.dim, self.n_free.copy())


def count_free(cbits, nJ):
    """n_free[b] from the constraint bits (host copy of what trs_dofmap computes)."""
    bits = np.asarray(cbits, dtype=np.uint8)
    valid = np.arange(bits.shape[1])[None, :] < np.asarray(nJ)[:, None]
    constrained = ((bits & 1) + ((bits >> 1) & 1) + ((bits >> 2) & 1)) * valid
    return (3 * np.asarray(nJ) - constrained.sum(axis=1)).astype(np.int32)


def pack_arrays(xyz_list, conn_list, mtype_list, support_list, loads_list, dims):
    """Pack per-truss arrays: xyz [nJ,dim], conn [nM,2], mtype [nM,3]=(a,e,density),
    support [nJ] SupportType values, loads [nJ,dim]."""
    B = len(xyz_list)
    nJ = np.array([len(x) for x in xyz_list], dtype=np.int32)
    nM = np.array([len(c) for c in conn_list], dtype=np.int32)
    nJ_max, nM_max = max(1, int(nJ.max(initial=1))), max(1, int(nM.max(initial=1)))
    xyz = np.zeros([B, nJ_max, 3])
    loads = np.zeros([B, nJ_max, 3])
    cbits = np.zeros([B, nJ_max], dtype=np.uint8)
    conn = np.zeros([B, nM_max, 2], dtype=np.int32)
    E = np.ones([B, nM_max])
    A = np.ones([B, nM_max])
    rho = np.zeros([B, nM_max])
    for b in range(B):
        dim = dims[b]
        x = np.asarray(xyz_list[b], dtype=float).reshape(-1, dim)
        xyz[b, :nJ[b], :dim] = x
        loads[b, :nJ[b], :dim] = np.asarray(loads_list[b], dtype=float).reshape(-1, dim)
        table = SupportType._BITS3 if dim == 3 else SupportType._BITS2
        try:    # (one dict look-up per joint; an unknown type takes the slow road to the reference's exception)
            bits = np.array([table[s] for s in support_list[b]], dtype=np.uint8)
        except (KeyError, TypeError):
            bits = np.array([SupportType.ConstraintBits(s, dim) for s in support_list[b]], dtype=np.uint8)
        cbits[b, :nJ[b]] = bits | (4 if dim == 2 else 0)  # a 2D truss never moves in z
        if nM[b]:
            conn[b, :nM[b]] = np.asarray(conn_list[b], dtype=np.int32).reshape(-1, 2)
            mt = np.asarray(mtype_list[b], dtype=float).reshape(-1, 3)
            A[b, :nM[b]], E[b, :nM[b]], rho[b, :nM[b]] = mt[:, 0], mt[:, 1], mt[:, 2]
    dims = np.asarray(dims, dtype=np.int32)
    return PackedBatch(xyz, conn, E, A, rho, cbits, loads, nJ, nM, dims, count_free(cbits, nJ))


def pack_trusses(trusses):
    """`list[Truss]` -> `PackedBatch` (DOF order joint*dim + axis, reference `truss.py:312-314`)."""
    xyz, conn, mtype, sup, loads, dims = [], [], [], [], [], []
    for t in trusses:
        x, c, sections, supports, f = t.PackedArrays()
        xyz.append(x); conn.append(c); mtype.append(sections); sup.append(supports); loads.append(f)
        dims.append(t.dim)
    return pack_arrays(xyz, conn, mtype, sup, loads, dims)


def _member_form(packed, members):
    """`members`: "general" (E, A, rho per member), "table" (uint16 end joints, uint8 type index, type table; an
    error beyond 256 distinct triples) or "auto" (the table form whenever the batch allows it)."""
    if members == "general":
        return packed
    if members == "table":
        return packed.table()
    if members == "auto":
        try:
            return packed.table()
        except ValueError:
            return packed
    raise ValueError(f"members must be 'general', 'table' or 'auto', not {members!r}")


def pack_json(data_list, members="general"):
    """List of reference-format JSON dicts (`detail/combine_with_JSON.md:71-163`) -> PackedBatch,
    without building `Truss` objects (bulk path).  `members` = "auto": the table member form (`PackedBatch.table`)
    when the documents' `[a, e, density]` triples are at most 256 distinct ones."""
    xyz, conn, mtype, sup, loads, dims = [], [], [], [], [], []
    for data in data_list:
        dim = len(data["joint"][0][0])
        nJ = len(data["joint"])
        xyz.append(np.array([p for p, _ in data["joint"]], dtype=float))
        sup.append([SupportType.GetFromString(s) for _, s in data["joint"]])
        f = np.zeros([nJ, dim])
        for j, v in data["force"]:
            if np.any(np.abs(np.asarray(v, dtype=float)) >= 1e-10):  # zero loads are dropped (truss.py:181)
                f[j] = v
        loads.append(f)
        conn.append(np.array([c for c, _ in data["member"]], dtype=np.int32).reshape(-1, 2))
        mtype.append(np.array([t for _, t in data["member"]], dtype=float).reshape(-1, 3))
        dims.append(dim)
    return _member_form(pack_arrays(xyz, conn, mtype, sup, loads, dims), members)


_JSON_ERRORS = {1: "JSON syntax", 2: "inconsistent coordinate / load dimension", 3: "unknown or invalid support type",
                4: "joint id out of range", 5: "does not fit the padding", 6: "file cannot be read"}


def _pack_json_native(count, call):
    """Two passes through the native reader (`csrc/jsonpack.c`): sizes, then the padded arrays."""
    import ctypes
    nJ = np.zeros([count], dtype=np.int32); nM = np.zeros([count], dtype=np.int32)
    dim = np.full([count], 3, dtype=np.int32)
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    def run(nJ_max, nM_max, arrays):
        rc = call(nJ_max, nM_max, *(ptr(a) for a in arrays), ptr(nJ), ptr(nM), ptr(dim))
        if rc != 0:
            index, code = (-rc) // 1000 - 1, (-rc) % 1000
            raise ValueError(f"truss JSON #{index}: {_JSON_ERRORS.get(code, code)}")

    run(0, 0, [None] * 7)
    jm, mm = max(1, int(nJ.max(initial=1))), max(1, int(nM.max(initial=1)))
    xyz = np.empty([count, jm, 3]); loads = np.empty([count, jm, 3])
    conn = np.empty([count, mm, 2], dtype=np.int32)
    E = np.empty([count, mm]); A = np.empty([count, mm]); rho = np.empty([count, mm])
    cbits = np.empty([count, jm], dtype=np.uint8)
    run(jm, mm, [xyz, conn, E, A, rho, cbits, loads])
    return PackedBatch(xyz, conn, E, A, rho, cbits, loads, nJ, nM, dim, count_free(cbits, nJ))


def pack_json_texts(texts, members="general"):
    """Reference-format JSON documents (`bytes` or `str`, one truss each) -> `PackedBatch` through the
    native bulk reader (`csrc/jsonpack.c`, OpenMP over the documents): no Python objects per joint or
    member.  Same arrays as `pack_json([json.loads(t) for t in texts])`; `members` as there."""
    import ctypes
    from .generate import _load
    lib = _load()
    raw = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
    B = len(raw)
    arr = (ctypes.c_char_p * B)(*raw)
    lens = np.array([len(r) for r in raw], dtype=np.int64)
    lib.trs_json_pack.restype = ctypes.c_int
    return _member_form(_pack_json_native(B, lambda jm, mm, *rest: lib.trs_json_pack(
        ctypes.c_int(B), arr, lens.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(jm), ctypes.c_int(mm), *rest)), members)


def pack_json_files(paths, members="general"):
    """Truss JSON FILES -> `PackedBatch`; the files are read (once) and parsed natively, in parallel
    (the bulk form of `Truss.LoadFromJSON`, reference `truss.py:401-421`)."""
    import ctypes
    from .generate import _load
    lib = _load()
    raw = [os.fsencode(p) for p in paths]
    B = len(raw)
    arr = (ctypes.c_char_p * B)(*raw)
    bufs = (ctypes.c_void_p * B)()
    lens = np.zeros([B], dtype=np.int64)
    lib.trs_json_read_files.restype = ctypes.c_int
    lib.trs_json_pack.restype = ctypes.c_int
    rc = lib.trs_json_read_files(ctypes.c_int(B), arr, bufs, lens.ctypes.data_as(ctypes.c_void_p))
    try:
        if rc != 0:
            raise ValueError(f"truss JSON #{(-rc) // 1000 - 1}: {_JSON_ERRORS[6]}")
        return _member_form(_pack_json_native(B, lambda jm, mm, *rest: lib.trs_json_pack(
            ctypes.c_int(B), bufs, lens.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(jm), ctypes.c_int(mm), *rest)),
                            members)
    finally:
        lib.trs_json_free_files(ctypes.c_int(B), bufs)


@dataclass
class BatchResult:
    """Dense host results of a batched solve: displace/external [B,nJ_max,3], internal [B,nM_max],
    info [B] (0 ok, k>0: pivot k of the reduced stiffness matrix is not positive)."""
    displace: np.ndarray
    external: np.ndarray
    internal: np.ndarray
    info: np.ndarray


def _require_gpu(device):
    try:
        import torch
    except ImportError as exc:  # pragma: no cover
        raise HipExtensionError("PyTorch-ROCm is required for device memory and streams") from exc
    if not torch.cuda.is_available():
        raise HipExtensionError("no GPU visible: the truss solver has no CPU fallback")
    return torch, torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


#: per-batch switches of the pipeline and their defaults.  None of them changes a result; they select
#: between kernels that compute the same thing (A/B runs, tests).  They travel as flags / hints of every C
#: call (include/trs_solver.h) - the library has no process-wide state - and are attributes of a
#: `DeviceBatch` (`dev.options["compact"] = True`, or `DeviceBatch(..., options={...})`).
#:   compact             K_ff leaves the assembly as compact entry lists, tiles formed in the factorisation
#:   fused_substitution  the wave that factors a narrow-envelope matrix substitutes it as well
#:   recover_unstaged    force trs_recover's path for trusses whose tables exceed a CU's LDS
#:   recover_scan        with recover_unstaged: also that path's fall-back without member-end lists (tests)
#:   all_wide            every matrix to the work-group factorisation (four waves per matrix), whatever its envelope:
#:                       for batches of few, large systems, which leave most SIMDs idle under the wave-per-matrix kernel
DEFAULT_OPTIONS = {"compact": False, "fused_substitution": True, "recover_unstaged": False, "recover_scan": False,
                   "all_wide": False}


def default_options():
    """`DEFAULT_OPTIONS`, overridden by the environment variable TRS_OPTIONS="name=value,name=value" (A/B runs
    of the tools without touching code); an unknown name is an error."""
    opts = dict(DEFAULT_OPTIONS)
    for item in filter(None, os.environ.get("TRS_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        if name.strip() not in opts:
            raise ValueError(f"TRS_OPTIONS: unknown option {name.strip()!r} (known: {sorted(opts)})")
        opts[name.strip()] = bool(int(value))
    return opts


def order_plan(reorder, nJ_max, nM_max):
    """How a `reorder=` argument is carried out: None (no renumbering), ("given", perm array), ("device", effort)
    - `trs_joint_order` on the GPU, no host pass - or ("host", name) - `joint_order` on the host.

    True / "auto" = every coordinate sweep, plus reverse Cuthill-McKee and its reverse for small trusses (fewer than
    128 free joints) or where no sweep is possible - effort 3: on larger lattice-like trusses a sweep wins, and
    Cuthill-McKee is 40 % of the device kernel's time; "profile" = every candidate for every truss (effort 2);
    "fast" = RCM, its reverse and one sweep: on the device whenever the batch shape fits its kernel
    (`trs_joint_order_fits`), else on the host.  "rcm" = plain reverse Cuthill-McKee (host).  "host-auto" /
    "host-profile" / "host-fast" force the host versions (comparisons, tests); "device" insists on the device
    version (effort 3) and raises when the shape does not fit."""
    if reorder is False or reorder is None:
        return None
    if isinstance(reorder, np.ndarray):
        return ("given", reorder)
    if reorder in ("host-auto", "host-profile", "host-fast", "rcm"):
        return ("host", {"host-auto": "auto", "host-profile": "profile", "host-fast": "fast"}.get(reorder, reorder))
    if reorder is True or reorder in ("auto", "profile", "fast", "device"):
        fits = bool(_capi.load().trs_joint_order_fits(int(nJ_max), int(nM_max)))
        if fits:
            return ("device", {"fast": 1, "profile": 2}.get(reorder, 3))
        if reorder == "device":
            raise ValueError(f"joint order on the device: batch shape ({nJ_max} joints, {nM_max} members) does not "
                             "fit trs_joint_order (see trs_joint_order_fits)")
        return ("host", reorder if reorder in ("fast", "profile") else "auto")
    raise ValueError(f"unknown joint order {reorder!r} (True, 'auto', 'profile', 'fast', 'rcm', 'device', 'host-auto', "
                     "'host-profile', 'host-fast' or a permutation array)")


def joint_order_device(torch, tensors, effort=2, apply=True, want_choice=False, out=None):
    """`trs_joint_order` on resident inputs (`tensors`: xyz, conn, cbits, loads, nJ, nM on one device, padded
    shapes of `PackedBatch`): the cheapest joint order of every truss found ON THE GPU, asynchronously on the
    current stream.  Returns a dict: `perm` int32 [B, nJ_max] (old id of the joint that becomes joint k = the
    `joint_out` of the recovery), `reach` int32 [B] (envelope reach of the chosen order, the launch hint),
    `choice` (if asked) and, with `apply`, the renumbered `xyz`, `conn`, `cbits`, `loads`.  `out` = the dict
    of an earlier call with the same shapes: its tensors are written again (no allocation)."""
    lib = _capi.load()
    xyz, conn, cbits, loads = (tensors[k].contiguous() for k in ("xyz", "conn", "cbits", "loads"))
    # (uint16 end joints = the table member form: the _tab twin reads and writes them)
    order_fn, what = (lib.trs_joint_order_tab, "trs_joint_order_tab") if conn.dtype == torch.uint16 \
        else (lib.trs_joint_order, "trs_joint_order")
    dev = xyz.device
    B, nJ_max, nM_max = int(xyz.shape[0]), int(xyz.shape[1]), int(conn.shape[1])
    if out is None:
        out = {"perm": torch.empty([B, nJ_max], dtype=torch.int32, device=dev),
               "reach": torch.empty([B], dtype=torch.int32, device=dev)}
        if want_choice:
            out["choice"] = torch.empty([B], dtype=torch.int32, device=dev)
        if apply:
            out.update(xyz=torch.empty_like(xyz), conn=torch.empty_like(conn), cbits=torch.empty_like(cbits),
                       loads=torch.empty_like(loads))
    ptr = lambda k: out[k].data_ptr() if k in out else None
    with torch.cuda.device(dev):
        _capi.check(order_fn(
            B, nJ_max, nM_max, xyz.data_ptr(), conn.data_ptr(), cbits.data_ptr(), loads.data_ptr(),
            tensors["nJ"].data_ptr(), tensors["nM"].data_ptr(), out["perm"].data_ptr(), ptr("choice"),
            out["reach"].data_ptr(), ptr("xyz"), ptr("conn"), ptr("cbits"), ptr("loads"), int(effort),
            torch.cuda.current_stream(dev).cuda_stream), what)
    return out


class DeviceBatch:
    """A packed batch resident in HBM plus the workspace of the pipeline.

    Build once, call `solve()` any number of times (e.g. after `set_sections`).  Every tensor
    lives on one device; kernels run on the current stream of that device.
    """
    INPUT_FIELDS = ("xyz", "conn", "E", "A", "rho", "cbits", "loads", "nJ", "nM")
    #: the inputs of a batch in the TABLE member form (`PackedBatch.table`): uint16 end joints, a uint8 type index per
    #: member and - shared by the batch - the type table `types` [T, 3] = (a, e, density)
    TABLE_FIELDS = ("xyz", "conn", "type_idx", "cbits", "loads", "nJ", "nM")

    def __init__(self, packed: PackedBatch, device=None, use_envelope=True, use_small=True, reorder=False,
                 options=None):
        """`use_envelope=False` treats every reduced stiffness matrix as dense (no tile skipping);
        `use_small=False` keeps a batch of small trusses off the fused single-kernel path
        (`trs_solve_small`) and sends it through the staged pipeline; `reorder` (see `joint_order`) uploads
        the trusses with their joints renumbered for a narrower envelope - the results still arrive in the
        caller's numbering (`trs_recover` writes them through `joint_out`), so nothing else changes."""
        torch, dev = _require_gpu(device)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.packed = packed
        plan = order_plan(reorder, packed.nJ_max, packed.nM_max)
        if plan is not None and use_small and _capi.load().trs_solve_small_fits(packed.nJ_max, packed.nM_max, packed.n_max):
            plan = None   # the fused small-system kernel keeps its matrix in LDS: nothing to gain from an order
        perm, resident, ordered = None, packed, None
        if plan is not None and plan[0] != "device":
            perm = joint_order(packed.general(), plan[1])   # found once, on the host
            resident = permute_joints(packed.general(), perm)
            if packed.is_table:   # (members keep their order: the type indices stay, the end joints are renumbered)
                import dataclasses
                resident = dataclasses.replace(packed, xyz=resident.xyz, cbits=resident.cbits, loads=resident.loads,
                                               conn=np.ascontiguousarray(resident.conn, dtype=np.uint16))
        tensors = {f: up(getattr(resident, f)) for f in (self.TABLE_FIELDS if packed.is_table else self.INPUT_FIELDS)}
        if packed.is_table:
            tensors["types"] = up(np.asarray(packed.types, dtype=np.float64))
        if plan is not None and plan[0] == "device" and packed.B:
            # found, applied and priced on the GPU (trs_joint_order): no host pass over the batch
            ordered = joint_order_device(torch, tensors, effort=plan[1])
            tensors.update({k: ordered[k] for k in ("xyz", "conn", "cbits", "loads")})
        self._setup(torch, dev, tensors, packed.B, packed.nJ_max, packed.nM_max, packed.n_max, use_envelope, use_small)
        self.options.update(options or {})
        if perm is not None:
            self.joint_out = up(perm)
        # what the launches may be told about this batch (see `all_narrow`): the reach of its envelopes - from
        # the device order's own pricing, else from the host's envelope analysis
        if use_envelope and not self.small and packed.B:
            if ordered is not None:
                self.joint_out = ordered["perm"]
                self.all_narrow = bool(int(ordered["reach"].max().item()) <= NARROW_MAX_BELOW)
            else:
                self.all_narrow = bool(envelope_reach(resident.general()).max() <= NARROW_MAX_BELOW)
        elif ordered is not None:
            self.joint_out = ordered["perm"]

    @classmethod
    def from_device(cls, tensors, n_max, use_envelope=True, use_small=True, joint_out=None, all_narrow=False):
        """A batch whose inputs already live on the device: `tensors` maps INPUT_FIELDS (or, table member form,
        TABLE_FIELDS + "types") to contiguous
        device tensors of the padded shapes (see PackedBatch); `n_max` bounds the free DOFs per truss
        (host-known, it sizes the slab); `joint_out` (int32 [B, nJ_max] on the device, or None): the
        results of joint j go to row joint_out[b, j] (the joint order the caller applied to the inputs)."""
        torch, dev = _require_gpu(tensors["xyz"].device)
        self = cls.__new__(cls)
        self.packed = None
        self._setup(torch, dev, tensors, int(tensors["xyz"].shape[0]), int(tensors["xyz"].shape[1]),
                    int(tensors["conn"].shape[1]), int(n_max), use_envelope, use_small)
        if joint_out is not None:
            if self.small:
                raise ValueError("a batch on the fused small-system path takes no joint order")
            if tuple(joint_out.shape) != (self.B, self.nJ_max) or joint_out.dtype != torch.int32:
                raise ValueError("joint_out must be int32 [B, nJ_max]")
            self.joint_out = joint_out.contiguous()
        self.all_narrow = bool(all_narrow and use_envelope and not self.small)
        return self

    def _setup(self, torch, dev, tensors, B, nJ_max, nM_max, n_max, use_envelope, use_small=True):
        self.torch, self.device = torch, dev
        self.lib = _capi.load()
        self.B, self.nJ_max, self.nM_max, self.n_max = B, nJ_max, nM_max, n_max
        #: table member form: conn uint16, `type_idx` + `types` in place of E / A / rho (which are then None)
        self.table = tensors.get("type_idx") is not None
        for f in self.INPUT_FIELDS + ("type_idx", "types"):
            setattr(self, f, tensors.get(f))
        if self.table and (self.conn.dtype != torch.uint16 or self.type_idx.dtype != torch.uint8 or self.types is None):
            raise ValueError("table member form: conn uint16 [B, nM_max, 2], type_idx uint8 [B, nM_max], types float64 [T, 3]")
        self.ld = self.lib.trs_slab_ld(self.n_max)
        self.rows = self.lib.trs_slab_rows(self.n_max)
        #: the whole batch goes through the fused small-system kernel (n_free <= 128, tables fit the LDS)
        self.small = bool(use_small and self.lib.trs_solve_small_fits(nJ_max, nM_max, n_max))
        self.use_envelope = use_envelope
        self.free_index = torch.empty([B, self.nJ_max * 3], dtype=torch.int32, device=dev)
        self.n_free = torch.empty([B], dtype=torch.int32, device=dev)
        self.u = torch.empty([B, self.nJ_max, 3], dtype=torch.float64, device=dev)
        self.f_ext = torch.empty([B, self.nJ_max, 3], dtype=torch.float64, device=dev)
        self.N = torch.empty([B, self.nM_max], dtype=torch.float64, device=dev)
        self.info = torch.empty([B], dtype=torch.int32, device=dev)
        self._slab = None   # (S, uf, work, env): the staged pipeline's workspace, allocated on first use
        self.joint_out = None   # int32 [B, nJ_max]: where the results of (resident) joint j go, or None
        #: the host knows that no envelope of this batch reaches further than NARROW_MAX_BELOW chunks below its
        #: diagonal blocks (`envelope_reach`): every matrix is then routed to the wave-per-matrix kernels
        #: (TRS_ASM_ALL_NARROW) and the work-group kernels, which would find nothing, are not launched.  Safe
        #: either way - with the flag the device routes so regardless - it only must not outlive the topology.
        self.all_narrow = False
        #: the host has SEEN (`adopt_tile_hint`) that no matrix of this resident batch leaves tiles of its envelope
        #: unwritten (trs_common.h `kmask`: a tower-like truss has too few tiles without an entry of K_ff for the
        #: skipping to pay): the assembly is then told not to form the masks again (TRS_ASM_ALL_TILES).  Like
        #: `all_narrow` it must not outlive the topology (`upload` resets it); a wrong value costs time, never a result.
        self.all_tiles = False
        self._potrf_fused = False
        self.options = default_options()

    def _workspace(self):
        """Stiffness slab, reduced solution, assembly workspace and envelope metadata of the staged
        pipeline; a batch on the fused small path never allocates them."""
        if self._slab is None:
            torch, dev, B = self.torch, self.device, self.B
            S = torch.empty([B, self.rows, self.ld], dtype=torch.float64, device=dev)
            if os.environ.get("TRS_DEBUG_POISON"):   # tests: any read of a never-written slab entry shows
                S.fill_(float("nan"))
            uf = torch.empty([B, self.rows], dtype=torch.float64, device=dev)
            work_bytes = self.lib.trs_assemble_work_bytes(self.nJ_max, self.nM_max, self.n_max)
            work = torch.empty([B, work_bytes], dtype=torch.uint8, device=dev)
            env = torch.zeros([B, self.lib.trs_env_ints(self.n_max)], dtype=torch.int32, device=dev) \
                if self.use_envelope else None
            self._slab = (S, uf, work, env)
        return self._slab

    S = property(lambda self: self._workspace()[0])
    uf = property(lambda self: self._workspace()[1])
    work = property(lambda self: self._workspace()[2])
    env = property(lambda self: self._workspace()[3])

    # -- individual stages (used by the parity tests and the benchmark) ------------------
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _env_ptr(self):
        return self.env.data_ptr() if self.env is not None else None

    def _members(self, types=None):
        """The three member arguments of a C call: (conn, E, A), or - table form - (conn16, type_idx, types)."""
        if self.table:
            return self.conn.data_ptr(), self.type_idx.data_ptr(), (self.types if types is None else types).data_ptr()
        return self.conn.data_ptr(), self.E.data_ptr(), self.A.data_ptr()

    def _fn(self, name):
        """The entry point `name`, or its `_tab` twin for a batch in the table member form."""
        return getattr(self.lib, name + "_tab" if self.table else name), name + ("_tab" if self.table else "")

    def dofmap(self):
        _capi.check(self.lib.trs_dofmap(self.B, self.nJ_max, self.cbits.data_ptr(), self.nJ.data_ptr(),
                                        self.free_index.data_ptr(), self.n_free.data_ptr(),
                                        self._stream()), "trs_dofmap")

    def _hints(self, substituted=False):
        if not self.all_narrow or self.env is None or self.options["all_wide"]:
            return 0
        # (whether THIS batch's last factorisation ran with the substitution fused in, not the option's value now)
        fused = substituted and self.rows <= 1024 and self._potrf_fused
        return HINT_NO_WIDE | (HINT_SUBSTITUTED if fused else 0)

    def _stage_hints(self):
        return (HINT_COMPACT if self.options["compact"] and self.env is not None and not self.options["all_wide"] else 0) | \
               (0 if self.options["fused_substitution"] else HINT_SEPARATE_STAGES) | \
               (HINT_RECOVER_UNSTAGED if self.options["recover_unstaged"] else 0) | \
               (HINT_RECOVER_SCAN if self.options["recover_scan"] else 0)

    def assemble(self, flags=0):
        wide = self.options["all_wide"] and self.env is not None
        if self.all_narrow and self.env is not None and not wide:
            flags |= ASM_ALL_NARROW
        if self.options["compact"] and self.env is not None and not wide:
            flags |= ASM_COMPACT
        if wide:
            flags |= ASM_ALL_WIDE
        if self.all_tiles and self.env is not None:
            flags |= ASM_ALL_TILES
        fn, what = self._fn("trs_assemble")
        _capi.check(fn(
            self.B, self.nJ_max, self.nM_max, self.xyz.data_ptr(), *self._members(),
            self.loads.data_ptr(), self.free_index.data_ptr(),
            self.n_free.data_ptr(), self.nJ.data_ptr(), self.nM.data_ptr(), self.ld, self.rows,
            self.S.data_ptr(), flags, self.work.data_ptr(), self._env_ptr(), self.uf.data_ptr(), self.rows,
            self._stream()), what)

    def potrf(self):
        self._potrf_fused = bool(self.options["fused_substitution"])
        _capi.check(self.lib.trs_potrf_batched(self.B, self.n_free.data_ptr(), self.ld, self.rows,
                                               self.S.data_ptr(), self.info.data_ptr(), self._env_ptr(),
                                               self.work.data_ptr(), self.uf.data_ptr(), self.rows,
                                               self._hints() | (self._stage_hints() & (HINT_COMPACT | HINT_SEPARATE_STAGES)),
                                               self._stream()), "trs_potrf_batched")

    def potrs(self):
        _capi.check(self.lib.trs_potrs_batched(self.B, self.n_free.data_ptr(), self.ld, self.rows,
                                               self.S.data_ptr(), self.uf.data_ptr(), self.rows,
                                               self._env_ptr(), self._hints(substituted=True), self._stream()),
                    "trs_potrs_batched")

    def recover(self):
        fn, what = self._fn("trs_recover")
        _capi.check(fn(
            self.B, self.nJ_max, self.nM_max, self.xyz.data_ptr(), *self._members(),
            self.loads.data_ptr(), self.free_index.data_ptr(),
            self.nJ.data_ptr(), self.nM.data_ptr(), self.uf.data_ptr(), self.rows, self.u.data_ptr(),
            self.f_ext.data_ptr(), self.N.data_ptr(),
            self.joint_out.data_ptr() if self.joint_out is not None else None,
            self._stage_hints() & (HINT_RECOVER_UNSTAGED | HINT_RECOVER_SCAN), self._stream()), what)

    def recover_rows(self, rows, out, nJ_out_max, nM_out_max):
        """`trs_recover_rows`: the recovery with a ragged batch's bucket scatter folded in - the results of truss b go
        to row rows[b] (int64 device tensor) of `out["u"]`, `out["f_ext"]` [*, nJ_out_max, 3], `out["N"]`
        [*, nM_out_max] and `out["info"]`."""
        fn, what = self._fn("trs_recover_rows")
        _capi.check(fn(
            self.B, self.nJ_max, self.nM_max, self.xyz.data_ptr(), *self._members(),
            self.loads.data_ptr(), self.free_index.data_ptr(),
            self.nJ.data_ptr(), self.nM.data_ptr(), self.uf.data_ptr(), self.rows,
            self.joint_out.data_ptr() if self.joint_out is not None else None, self.info.data_ptr(),
            rows.data_ptr(), int(nJ_out_max), int(nM_out_max), out["u"].data_ptr(), out["f_ext"].data_ptr(),
            out["N"].data_ptr(), out["info"].data_ptr(), self._stage_hints() & (HINT_RECOVER_UNSTAGED | HINT_RECOVER_SCAN), self._stream()),
            what)

    def solve_rows(self, rows, out, nJ_out_max, nM_out_max, types=None):
        """`trs_solve_rows`: the staged pipeline in one C call with `recover_rows` as its last stage.  `types` (table
        member form only): another type table for this solve (e.g. every row the same fixed section)."""
        fn, what = self._fn("trs_solve_rows")
        with self.torch.cuda.device(self.device):
            _capi.check(fn(
                self.B, self.nJ_max, self.nM_max, self.n_max, self.xyz.data_ptr(), *self._members(types),
                self.cbits.data_ptr(), self.loads.data_ptr(),
                self.nJ.data_ptr(), self.nM.data_ptr(), self.free_index.data_ptr(),
                self.n_free.data_ptr(), self.ld, self.rows, self.S.data_ptr(), self.uf.data_ptr(),
                self.rows, out["u"].data_ptr(), out["f_ext"].data_ptr(), out["N"].data_ptr(),
                self.info.data_ptr(), self.work.data_ptr(), self._env_ptr(),
                self.joint_out.data_ptr() if self.joint_out is not None else None, rows.data_ptr(),
                int(nJ_out_max), int(nM_out_max), out["info"].data_ptr(),
                (HINT_NO_WIDE if self.all_narrow and self.env is not None else 0) | self._stage_hints() |
                (HINT_ALL_TILES if self.all_tiles and self.env is not None else 0) | self._wide_hint(),
                self._stream()), what)

    def _solve_small(self, fitness=None, out=None, types=None):
        """`trs_solve_small`: the whole of `Truss.Solve()` in one kernel (optionally with the GA
        reductions); returns the three reduction tensors (`out`, three float64 [B] device tensors, or new ones) or
        None."""
        t = self.torch
        if fitness is None:
            out = [None, None, None]
        elif out is None:
            out = [t.empty([self.B], dtype=t.float64, device=self.device) for _ in range(3)]
        ptr = lambda x: None if x is None else x.data_ptr()
        fn, what = self._fn("trs_solve_small")
        # (general form: the densities of the fitness reductions are an argument; table form: they are in the table)
        rho = () if self.table else (self.rho.data_ptr() if fitness is not None else None,)
        with t.cuda.device(self.device):
            _capi.check(fn(
                self.B, self.nJ_max, self.nM_max, self.n_max, self.xyz.data_ptr(), *self._members(types),
                self.cbits.data_ptr(), self.loads.data_ptr(),
                self.nJ.data_ptr(), self.nM.data_ptr(), self.u.data_ptr(), self.f_ext.data_ptr(),
                self.N.data_ptr(), self.info.data_ptr(), self.free_index.data_ptr(), self.n_free.data_ptr(), *rho,
                float(fitness[0]) if fitness else 0.0, float(fitness[1]) if fitness else 0.0,
                ptr(out[0]), ptr(out[1]), ptr(out[2]), self._stream()), what)
        return out if fitness is not None else None

    def solve_fitness(self, allow_stress, allow_displace, out=None):
        """Solve and reduce to (weight, stress_violation, displacement_violation) per truss - one kernel
        on the fused small path, `solve()` + `fitness()` otherwise (GA generation, ga.py:139-160).  `out`: three
        float64 [B] device tensors to write into (e.g. the rows of one [3, B] tensor: one download)."""
        if self.small:
            return self._solve_small((allow_stress, allow_displace), out)
        self.solve()
        return self.fitness(allow_stress, allow_displace, out)

    def set_sections_from_genes(self, genes, count, n_member, type_table):
        """`trs_ga_sections`: A, E, rho of the resident batch from a GA population's gene matrix (uint8 device tensor
        [count, n_member]) and its type table (float64 device tensor [n_type, 3] = a, e, density)."""
        if self.table:
            raise ValueError("set_sections_from_genes needs the general member form")
        _capi.check(self.lib.trs_ga_sections(self.B, self.nM_max, int(count), int(n_member), int(type_table.shape[0]),
                                             genes.data_ptr(), type_table.data_ptr(), self.A.data_ptr(),
                                             self.E.data_ptr(), self.rho.data_ptr(), self._stream()), "trs_ga_sections")

    def _wide_hint(self):
        return HINT_ALL_WIDE if self.options["all_wide"] and self.env is not None else 0

    def solve(self, types=None):
        """The whole pipeline, asynchronous on the current stream: one kernel for a batch of small
        trusses (`trs_solve_small`), otherwise one C call that enqueues the five stages.  `types` as `solve_rows`."""
        if self.small:
            self._solve_small(types=types)
            return
        fn, what = self._fn("trs_solve")
        with self.torch.cuda.device(self.device):
            _capi.check(fn(
                self.B, self.nJ_max, self.nM_max, self.n_max, self.xyz.data_ptr(), *self._members(types),
                self.cbits.data_ptr(), self.loads.data_ptr(),
                self.nJ.data_ptr(), self.nM.data_ptr(), self.free_index.data_ptr(),
                self.n_free.data_ptr(), self.ld, self.rows, self.S.data_ptr(), self.uf.data_ptr(),
                self.rows, self.u.data_ptr(), self.f_ext.data_ptr(), self.N.data_ptr(),
                self.info.data_ptr(), self.work.data_ptr(), self._env_ptr(),
                self.joint_out.data_ptr() if self.joint_out is not None else None,
                # (not on the fused small path here, by shape or by request: the staged pipeline in any case)
                (HINT_NO_WIDE if self.all_narrow and self.env is not None else 0) | self._stage_hints() | HINT_NO_SMALL |
                (HINT_ALL_TILES if self.all_tiles and self.env is not None else 0) | self._wide_hint(),
                self._stream()), what)

    def fitness(self, allow_stress, allow_displace, out=None):
        """(weight, stress_violation, displacement_violation) per truss, on device."""
        t = self.torch
        if self.table:
            raise ValueError("trs_fitness exists in the general member form only (the fused small-system kernel takes either)")
        if out is None:
            out = [t.empty([self.B], dtype=t.float64, device=self.device) for _ in range(3)]
        _capi.check(self.lib.trs_fitness(
            self.B, self.nJ_max, self.nM_max, self.xyz.data_ptr(), self.conn.data_ptr(),
            self.A.data_ptr(), self.rho.data_ptr(), self.nJ.data_ptr(), self.nM.data_ptr(),
            self.u.data_ptr(), self.N.data_ptr(), float(allow_stress), float(allow_displace),
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), self._stream()), "trs_fitness")
        return out

    def adopt_tile_hint(self):
        """After a solve (or an assembly) of the resident batch: read back whether ANY matrix left tiles of its
        envelope unwritten (one int per truss of the envelope metadata) and, if none did, tell the later assemblies
        of this topology not to form the tile masks again (`all_tiles`).  Synchronises.  Returns `all_tiles`."""
        if self.small or self.env is None or self.B == 0:
            return False
        off = 2 * (self.rows // 16) + self.rows // 64 + 8       # kmask[0] of every truss (csrc/trs_common.h)
        self.all_tiles = bool((self.env[:, off] == -1).all().item())
        return self.all_tiles

    @property
    def fields(self):
        """Names of this batch's per-truss input tensors (its member form's)."""
        return self.TABLE_FIELDS if self.table else self.INPUT_FIELDS

    def pinned_inputs(self, packed: PackedBatch):
        """Page-locked host copies of a batch's inputs (same padded shapes and member form as this device batch)."""
        t = self.torch
        if packed.is_table != self.table:
            raise ValueError("pinned_inputs: the batch's member form differs from the resident batch's")
        return {f: t.from_numpy(np.ascontiguousarray(getattr(packed, f))).pin_memory() for f in self.fields}

    def upload(self, host_inputs):
        """Replace the resident inputs by another batch of the same padded shapes (asynchronous on
        the current stream when `host_inputs` come from `pinned_inputs`)."""
        self.all_narrow = False   # another topology: the host's knowledge of the envelopes is gone
        self.all_tiles = False
        for f in self.fields:
            getattr(self, f).copy_(host_inputs[f], non_blocking=True)

    def download(self, out=None):
        """Copy the dense results to (pinned) host tensors, asynchronously; returns the dict."""
        t = self.torch
        if out is None:
            out = {k: t.empty(v.shape, dtype=v.dtype).pin_memory()
                   for k, v in (("u", self.u), ("f_ext", self.f_ext), ("N", self.N), ("info", self.info))}
        for k in out:
            out[k].copy_(getattr(self, k), non_blocking=True)
        return out

    def set_sections(self, A, E, rho):
        """Replace the member sections (host arrays [B,nM_max]); geometry stays resident."""
        if self.table:
            raise ValueError("set_sections: a batch in the table member form changes its sections through `types` / `type_idx`")
        for dst, src in ((self.A, A), (self.E, E), (self.rho, rho)):
            dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src, dtype=np.float64)))

    def result(self):
        """Synchronise and download the dense results."""
        self.torch.cuda.synchronize(self.device)
        return BatchResult(self.u.cpu().numpy(), self.f_ext.cpu().numpy(), self.N.cpu().numpy(),
                           self.info.cpu().numpy())


class StreamedSolver:
    """Host -> device -> host pipeline for a stream of equally shaped batches (the feed of a resident
    solver over PCIe): the upload of batch k+1, the solve of batch k and the download of batch k-1 overlap
    on three HIP streams, with `slots` resident `DeviceBatch`es and page-locked staging buffers.

        pipe = StreamedSolver(template_packed)
        for packed in batches:                     # same padded shapes as the template
            done = pipe.submit(pipe.stage(packed)) # the results of the batch submitted `slots` calls ago, or None
            ...consume `done` here...
        for done in pipe.drain(): ...

    Results are views of page-locked output buffers (a ring of slots + 1): what `submit` returns stays
    valid until the NEXT call of `submit`; `drain` results until the pipe is used again.  Only what the
    solve reads is uploaded (`fields`; the densities are not part of `Truss.Solve()`)."""

    def __init__(self, template: PackedBatch, device=None, slots=2, use_envelope=True,
                 fields=("xyz", "conn", "E", "A", "cbits", "loads", "nJ", "nM"), same_topology=False):
        """`same_topology=True`: every batch of the stream has the template's connectivity and supports (only
        coordinates, sections and loads vary - a parameter study, a GA, a replicated benchmark batch), so the
        launch hints the host derived from the template's envelopes hold for all of them."""
        torch, dev = _require_gpu(device)
        self.torch, self.device, self.fields = torch, dev, tuple(fields)
        self.dev = [DeviceBatch(template, dev, use_envelope=use_envelope) for _ in range(slots)]
        if not same_topology:   # the batches that follow need not share the template's envelopes: no launch hints
            for d in self.dev:
                d.all_narrow = False

        # Every slot's uploaded fields live in ONE device buffer and ONE page-locked host buffer of the same
        # layout (likewise the four result fields), so that a batch crosses PCIe as one DMA per direction
        # instead of eight + four copies; the DeviceBatch tensors and the host dicts are views of them.
        def flat(like, pinned):
            offs, total = {}, 0
            for k, v in like.items():
                offs[k] = total
                total += (v.numel() * v.element_size() + 255) // 256 * 256
            buf = torch.empty([total], dtype=torch.uint8, pin_memory=True) if pinned else \
                torch.empty([total], dtype=torch.uint8, device=dev)
            views = {k: buf[offs[k]: offs[k] + v.numel() * v.element_size()].view(v.dtype).view(v.shape)
                     for k, v in like.items()}
            return buf, views

        self.host_in, self._host_in_flat, self._dev_in_flat = [], [], []
        self.host_out, self._host_out_flat, self._dev_out_flat = [], [], []
        for d in self.dev:
            ins = {f: getattr(d, f) for f in self.fields}
            dbuf, dviews = flat(ins, pinned=False)
            hbuf, hviews = flat(ins, pinned=True)
            for f in self.fields:
                dviews[f].copy_(ins[f])
                hviews[f].copy_(ins[f])
                setattr(d, f, dviews[f])
            outs = {k: getattr(d, k) for k in ("u", "f_ext", "N", "info")}
            obuf, oviews = flat(outs, pinned=False)
            for k in outs:
                setattr(d, k, oviews[k])
            self._dev_in_flat.append(dbuf); self._host_in_flat.append(hbuf); self.host_in.append(hviews)
            self._dev_out_flat.append(obuf)
        torch.cuda.synchronize(dev)
        like_out = {k: getattr(self.dev[0], k) for k in ("u", "f_ext", "N", "info")}
        for _ in range(slots + 1):
            hbuf, hviews = flat(like_out, pinned=True)
            self._host_out_flat.append(hbuf); self.host_out.append(hviews)
        self.s_up, self.s_run, self.s_down = (torch.cuda.Stream(dev) for _ in range(3))
        self.ev_up = [torch.cuda.Event() for _ in range(slots)]       # upload into device slot finished
        self.ev_run = [torch.cuda.Event() for _ in range(slots)]      # solve on device slot finished
        self.ev_down = [torch.cuda.Event() for _ in range(slots)]     # device slot's results downloaded
        self.count = 0
        self.pending = []   # (device slot, host buffer) of the downloads in flight, oldest first

    def stage(self, packed: PackedBatch):
        """Copy a batch's host arrays into the next slot's page-locked staging buffers."""
        slot = self.count % len(self.dev)
        self.ev_up[slot].synchronize()   # the previous upload out of this staging buffer has finished
        for f in self.fields:
            self.host_in[slot][f].numpy()[...] = getattr(packed, f)
        return self.host_in[slot]

    def submit(self, host_inputs=None):
        """Enqueue upload -> solve -> download of one batch (`host_inputs`: pinned tensors by field name,
        default: the slot's staging buffers).  Returns the `BatchResult` of the batch submitted `slots`
        calls ago (its download has finished), else None."""
        t = self.torch
        n = len(self.dev)
        slot, hbuf = self.count % n, self.count % (n + 1)
        self.count += 1
        done = self._take(self.pending.pop(0)) if len(self.pending) == n else None
        dev, src = self.dev[slot], host_inputs if host_inputs is not None else self.host_in[slot]
        own = [i for i, h in enumerate(self.host_in) if h is src]   # one of this pipe's own staging buffers?
        with t.cuda.stream(self.s_up):
            self.s_up.wait_event(self.ev_run[slot])     # the slot's previous solve no longer reads its inputs
            if own:   # one DMA for all fields
                self._dev_in_flat[slot].copy_(self._host_in_flat[own[0]], non_blocking=True)
            else:
                for f in self.fields:
                    getattr(dev, f).copy_(src[f], non_blocking=True)
            self.ev_up[slot].record(self.s_up)
        with t.cuda.stream(self.s_run):
            self.s_run.wait_event(self.ev_up[slot])
            self.s_run.wait_event(self.ev_down[slot])   # the slot's previous results have been downloaded
            dev.solve()
            self.ev_run[slot].record(self.s_run)
        with t.cuda.stream(self.s_down):
            self.s_down.wait_event(self.ev_run[slot])
            self._host_out_flat[hbuf].copy_(self._dev_out_flat[slot], non_blocking=True)   # one DMA for the four results
            self.ev_down[slot].record(self.s_down)
        self.pending.append((slot, hbuf))
        return done

    def _take(self, entry):
        slot, hbuf = entry
        self.ev_down[slot].synchronize()
        o = self.host_out[hbuf]
        return BatchResult(o["u"].numpy(), o["f_ext"].numpy(), o["N"].numpy(), o["info"].numpy())

    def drain(self):
        """Wait for everything in flight; the results of the batches not yet handed out, oldest first."""
        out = [self._take(e) for e in self.pending]
        self.pending = []
        return out


def rcm_permutation(packed: PackedBatch):
    """Reverse Cuthill-McKee joint order of every truss (native, `csrc/reorder.c`):
    perm[b, k] = old id of the joint that becomes joint k.  Shrinks the envelope of the reduced
    stiffness matrix of trusses that are not numbered along their long axis (cube trusses)."""
    import ctypes
    from .generate import _load
    lib = _load()
    lib.trs_rcm_order.restype = ctypes.c_int
    perm = np.empty([packed.B, packed.nJ_max], dtype=np.int32)
    ptr = lambda a: np.ascontiguousarray(a).ctypes.data_as(ctypes.c_void_p)
    conn, cbits, nJ, nM = (np.ascontiguousarray(a) for a in (packed.conn, packed.cbits, packed.nJ, packed.nM))
    rc = lib.trs_rcm_order(ctypes.c_int(packed.B), ctypes.c_int(packed.nJ_max), ctypes.c_int(packed.nM_max),
                           ptr(conn), ptr(cbits), ptr(nJ), ptr(nM), perm.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"trs_rcm_order failed ({rc})")
    return perm


def profile_permutation(packed: PackedBatch, return_choice=False, effort=2):
    """The cheapest of several candidate joint orders per truss (native, `csrc/reorder.c`
    `trs_profile_order`): reverse Cuthill-McKee, its reverse and twelve binned coordinate sweeps, priced
    by the 16x16-tile envelope the factorisation works in.  Never worse than `rcm_permutation` (efforts 0-2); on
    the reference's cube trusses 25-35 % less factorisation work.  perm[b, k] = old id of the joint that
    becomes joint k; `return_choice` adds the winning candidate's id per truss (0 = RCM).  `effort`: 0 RCM and
    its reverse, 1 + one sweep, 2 everything, 3 every sweep and the RCM pair only for trusses below 128 free joints."""
    import ctypes
    from .generate import _load
    lib = _load()
    lib.trs_profile_order.restype = ctypes.c_int
    perm = np.empty([packed.B, packed.nJ_max], dtype=np.int32)
    choice = np.empty([packed.B], dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    xyz = np.ascontiguousarray(packed.xyz, dtype=np.float64)
    conn, nJ, nM = (np.ascontiguousarray(a, dtype=np.int32) for a in (packed.conn, packed.nJ, packed.nM))
    cbits = np.ascontiguousarray(packed.cbits, dtype=np.uint8)
    rc = lib.trs_profile_order(ctypes.c_int(packed.B), ctypes.c_int(packed.nJ_max), ctypes.c_int(packed.nM_max),
                               ptr(xyz), ptr(conn), ptr(cbits), ptr(nJ), ptr(nM), ptr(perm), ptr(choice),
                               ctypes.c_int(effort))
    if rc != 0:
        raise RuntimeError(f"trs_profile_order failed ({rc})")
    return (perm, choice) if return_choice else perm


NARROW_MAX_BELOW = 24   # csrc/trs_common.h TRS_NARROW_MAX_BELOW: reach up to which a matrix goes to a wave of its own
# include/trs_solver.h
HINT_NO_WIDE, HINT_SUBSTITUTED, HINT_COMPACT, HINT_SEPARATE_STAGES, HINT_NO_SMALL, HINT_RECOVER_UNSTAGED = 1, 2, 4, 8, 16, 32
HINT_RECOVER_SCAN = 128
HINT_ALL_WIDE = 256
HINT_ALL_TILES = 64
ORDER_RCM_BELOW = 128           # csrc/order.hip RCM_BELOW: effort 3 prices Cuthill-McKee below this many free joints
ASM_FULL_SYMMETRIC, ASM_COMPACT, ASM_ALL_NARROW, ASM_ALL_TILES, ASM_ALL_WIDE = 1, 2, 4, 8, 16


def envelope_reach(packed: PackedBatch, perm=None):
    """Per truss, how many 16-row chunks the row envelope of K_ff reaches below its 64 x 64 diagonal blocks in
    the numbering given, or after the renumbering `perm` (native, `csrc/reorder.c`; the metadata `trs_assemble`
    derives on the device).  A batch that stays at or below `NARROW_MAX_BELOW` holds no matrix for the
    work-group kernels."""
    import ctypes
    from .generate import _load
    lib = _load()
    lib.trs_envelope_reach.restype = ctypes.c_int
    reach = np.empty([packed.B], dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    conn, nJ, nM = (np.ascontiguousarray(a, dtype=np.int32) for a in (packed.conn, packed.nJ, packed.nM))
    cbits = np.ascontiguousarray(packed.cbits, dtype=np.uint8)
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    rc = lib.trs_envelope_reach(ctypes.c_int(packed.B), ctypes.c_int(packed.nJ_max), ctypes.c_int(packed.nM_max),
                                ptr(conn), ptr(cbits), ptr(nJ), ptr(nM), None if pm is None else ptr(pm), ptr(reach))
    if rc != 0:
        raise RuntimeError(f"trs_envelope_reach failed ({rc})")
    return reach


def joint_order(packed: PackedBatch, reorder):
    """The HOST-side permutation for a `reorder=` argument, with the same meaning of the names as `order_plan`:
    True / "auto" = `profile_permutation(effort=3)` (every sweep; Cuthill-McKee for small trusses), "profile" = every
    candidate for every truss, "fast" = one coordinate sweep instead of six (80 % of the gain for half the host
    time: what a host-in / host-out call wants when the order has to be found on the host, where it is the longest
    step), "rcm" = `rcm_permutation`; an int32 array [B, nJ_max] found earlier (e.g. on another thread) passes
    through."""
    if isinstance(reorder, np.ndarray):
        if reorder.shape != (packed.B, packed.nJ_max):
            raise ValueError(f"joint order of shape {reorder.shape}, expected {(packed.B, packed.nJ_max)}")
        return np.ascontiguousarray(reorder, dtype=np.int32)
    if reorder == "fast":
        return profile_permutation(packed, effort=1)
    if reorder == "rcm":
        return rcm_permutation(packed)
    if reorder is True or reorder == "auto":
        return profile_permutation(packed, effort=3)
    if reorder == "profile":
        return profile_permutation(packed)
    raise ValueError(f"unknown joint order {reorder!r} (True, 'auto', 'profile', 'fast', 'rcm' or a permutation array)")


def permute_joints(packed: PackedBatch, perm):
    """The same trusses with joint k := old joint perm[b, k] (members keep their order).
    Native (`csrc/reorder.c`, OpenMP over the batch)."""
    import ctypes
    from .generate import _load
    lib = _load()
    lib.trs_apply_joint_order.restype = ctypes.c_int
    B, nJ_max, nM_max = packed.B, packed.nJ_max, packed.nM_max
    src = [np.ascontiguousarray(a, dtype=t) for a, t in (
        (perm, np.int32), (packed.nM, np.int32), (packed.xyz, np.float64), (packed.conn, np.int32),
        (packed.cbits, np.uint8), (packed.loads, np.float64))]
    xyz, conn = np.empty_like(src[2]), np.empty_like(src[3])
    cbits, loads = np.empty_like(src[4]), np.empty_like(src[5])
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.trs_apply_joint_order(ctypes.c_int(B), ctypes.c_int(nJ_max), ctypes.c_int(nM_max),
                                   *(ptr(a) for a in src), ptr(xyz), ptr(conn), ptr(cbits), ptr(loads))
    if rc != 0:
        raise RuntimeError(f"trs_apply_joint_order failed ({rc})")
    return PackedBatch(xyz, conn, packed.E, packed.A, packed.rho, cbits, loads, packed.nJ, packed.nM,
                       packed.dim, packed.n_free)


def global_stiffness(trusses_or_packed, device=None):
    """The FULL global stiffness matrix of every truss, `Truss.GetKMatrix()` of the reference
    (`truss.py:307-316`: zero-initialised nDOF x nDOF, four dim x dim blocks added per member, DOF = joint * dim
    + axis, parallel members accumulate), assembled on the GPU by the same kernel as the solve: `trs_dofmap`
    on a batch with NO constraint frees every DOF (reduced index = DOF index) and `trs_assemble` with
    TRS_ASM_FULL_SYMMETRIC and no envelope writes the whole symmetric matrix.  Returns a list of B dense
    `np.ndarray`s of shape `[nJoint * dim, nJoint * dim]` (a 2D truss drops the embedded z axis)."""
    if not isinstance(trusses_or_packed, PackedBatch) and len(trusses_or_packed) == 0:
        return []
    packed = trusses_or_packed if isinstance(trusses_or_packed, PackedBatch) \
        else pack_trusses(list(trusses_or_packed))
    if packed.B == 0:
        return []
    torch, dev = _require_gpu(device)
    import dataclasses
    free = dataclasses.replace(packed, cbits=np.zeros_like(packed.cbits), n_free=(3 * packed.nJ).astype(np.int32))
    out = []
    # one slab per truss is nDOF^2 doubles: a few trusses at a time keep the workspace below ~1 GiB
    per = max(1, int((1 << 30) // max(1, (3 * packed.nJ_max + 80) ** 2 * 8)))
    with torch.cuda.device(dev):
        for lo in range(0, packed.B, per):
            part = free.take(np.arange(lo, min(packed.B, lo + per)))
            db = DeviceBatch(part, dev, use_envelope=False, use_small=False)
            db.dofmap()
            db.assemble(flags=ASM_FULL_SYMMETRIC)
            torch.cuda.synchronize(dev)
            S = db.S.cpu().numpy()
            for b in range(part.B):
                nJ, dim = int(part.nJ[b]), int(part.dim[b])
                K = S[b, :3 * nJ, :3 * nJ]
                if dim == 2:
                    keep = (np.arange(3 * nJ) % 3) < 2
                    K = K[keep][:, keep]
                out.append(np.ascontiguousarray(K))
            del db
    return out


SMALL_N = 128  # largest reduced system of the fused small-system kernel (csrc/small.hip)


def size_buckets(packed: PackedBatch, max_slab_bytes=64 << 30, granularity=64, quantum=0, span=2):
    """Group the trusses of a ragged batch for launching.  Returns a list of index arrays (their union
    is range(B)).

    * Every truss with at most SMALL_N free DOFs goes into ONE group when that group qualifies for the
      fused small-system kernel (`trs_solve_small_fits` on the group's own maxima): that kernel sizes
      its work per truss, so nothing is gained by splitting it.
    * The others are grouped by system size rounded up to `granularity` (a multiple of 64, the padded
      size n_pad: the slab and every work-group of a launch are then uniform), at most `max_slab_bytes`
      of stiffness slab per launch.
    * `quantum` > 0 (resident solvers: the matrices the factorisation kernel holds in flight, 12 per CU): the
      groups are cut at WHOLE ROUNDS of that kernel instead - the trusses in descending size, a group takes
      trusses of up to `span` further size classes below its largest and, where it can, a multiple of `quantum`
      of them.  A class of 3176 matrices runs 104 of them in a second round of their own; 65 536 cube trusses
      take 47.4 instead of 48.6 ms per step in ten groups instead of fourteen (`tools/round_buckets.py`).  The
      kernels work on every truss by its own size, so the results do not depend on the grouping."""
    g = max(64, int(granularity) // 64 * 64)
    n_pad = (packed.n_free.astype(np.int64) + g - 1) // g * g
    groups = []
    small = np.flatnonzero(packed.n_free <= SMALL_N)
    taken = np.zeros(packed.B, dtype=bool)
    # (with `quantum` also when they are of one size class: a bucket that spans further classes would take them through
    # the staged kernels, whose rounding differs from the small-system kernel's in the last bit - which kernel solves a
    # truss must not depend on how the batch is grouped, `tools/fuzz_streamed.py`)
    if len(small) and (quantum > 0 or len(np.unique(n_pad[small])) > 1):
        try:
            fits = _capi.load().trs_solve_small_fits(int(packed.nJ[small].max()), int(packed.nM[small].max()),
                                                     int(packed.n_free[small].max()))
        except HipExtensionError:
            fits = 0
        if fits:
            groups.append(small)
            taken[small] = True
    slab_bytes = lambda size: max(1, int(size) * (int(size) + 16) * 8)
    if quantum > 0:
        rest = np.flatnonzero(~taken)
        rest = rest[np.argsort(-n_pad[rest], kind="stable")]
        i = 0
        while i < len(rest):
            top = int(n_pad[rest[i]])
            cap = max(1, max_slab_bytes // slab_bytes(top))
            in_span = int(np.searchsorted(-n_pad[rest[i:]], -(top - g * max(0, int(span))), side="right"))
            take = min(cap, in_span)
            if take >= quantum:
                take = take // quantum * quantum
            # inside a group by ascending size: the factorisation takes its matrices from the last to the first, so
            # the largest start first and the chip empties over the smallest (47.2 against 48.5 ms per cube step)
            sl = np.sort(rest[i:i + take])
            sl = sl[np.argsort(packed.n_free[sl], kind="stable")]
            groups.append(sl)
            i += take
        return groups
    for size in np.unique(n_pad[~taken]):
        idx = np.flatnonzero((n_pad == size) & ~taken)
        step = max(1, max_slab_bytes // slab_bytes(size))
        groups.extend(idx[i: i + step] for i in range(0, len(idx), step))
    return groups


class SolverWorkspace:
    """Grow-only device buffers for the bucket pipelines of `RaggedSolver`s that run one after the other on one
    stream (slab, reduced vectors, assembly tables, envelope metadata, gathered un-ordered inputs): several
    solvers - the pieces of `solve_batch_streamed` - share one instead of allocating their own."""

    KINDS = {"S": "float64", "uf": "float64", "work": "uint8", "env": "int32", "raw_xyz": "float64",
             "raw_loads": "float64", "raw_cbits": "uint8", "raw_conn": "int32"}

    def __init__(self, torch, device):
        self.torch, self.device, self.buf = torch, device, {}
        self._lanes = {}

    def lane(self, index):
        """The workspace of lane `index` of a `RaggedSolver` that deals its buckets onto several streams (lane 0 is
        this workspace; the others are created on first use and live as long as it does)."""
        if index == 0:
            return self
        if index not in self._lanes:
            self._lanes[index] = SolverWorkspace(self.torch, self.device)
        return self._lanes[index]

    def nbytes(self):
        """Bytes this workspace and its lanes hold now."""
        own = sum(int(b.numel() * b.element_size()) for b in self.buf.values() if b is not None)
        return own + sum(ws.nbytes() for ws in self._lanes.values())

    def get(self, need):
        """Flat tensors of at least `need[kind]` elements each (zero-filled for the envelope metadata)."""
        t = self.torch
        grow = {kind: int(count) for kind, count in need.items()
                if self.buf.get(kind) is None or self.buf[kind].numel() < count}
        regrown = [kind for kind in grow if self.buf.get(kind) is not None]
        if regrown:
            # buffers that have to grow are dropped first and come back with headroom: a stream of batches of
            # slightly different shapes settles after a few steps.  The caching allocator keeps the dropped
            # blocks for other tensors; they go back to the driver only when the device is short of memory
            # (`empty_cache` costs a second with a hundred GB cached - never on the ordinary path)
            for kind in regrown:
                self.buf[kind] = None
                grow[kind] += grow[kind] // 8
            wanted = sum(grow[k] * t.empty((), dtype=getattr(t, self.KINDS[k])).element_size() for k in grow)
            free, _ = t.cuda.mem_get_info(self.device)
            reusable = t.cuda.memory_reserved(self.device) - t.cuda.memory_allocated(self.device)
            if free + reusable // 2 < wanted:
                t.cuda.empty_cache()
        for kind, count in grow.items():
            make = t.zeros if kind == "env" else t.empty
            self.buf[kind] = make([max(1, count)], dtype=getattr(t, self.KINDS[kind]), device=self.device)
            if kind == "S" and os.environ.get("TRS_DEBUG_POISON"):
                self.buf[kind].fill_(float("nan"))
        return self.buf


def default_slab_budget(torch, device, n_lanes, workspace=None):
    """Slab bytes a resident `RaggedSolver` may spread over its lanes when the caller names no budget:
    `LANE_SLAB_BYTES` per lane, but never more than 60 % of the device's memory and never more than 85 % of what is
    FREE now plus what the solver's own workspace (and this process's cached, unused blocks) already holds - a
    co-tenant of the device, e.g. the network being trained on the samples, keeps its memory; at least 2 GiB, so that a
    crowded device gives small buckets rather than none."""
    total = int(torch.cuda.get_device_properties(device).total_memory)
    free, _ = torch.cuda.mem_get_info(device)
    reusable = int(torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    held = workspace.nbytes() if workspace is not None else 0
    roomy = int(0.85 * (int(free) + reusable + held))
    return max(2 << 30, min(n_lanes * LANE_SLAB_BYTES, int(0.6 * total), roomy))


_SHARED_WORKSPACES = {}
_LANE_STREAMS = {}
#: streams a resident `RaggedSolver` deals its buckets onto unless told otherwise (`lanes=`), and the slab budget per lane
DEFAULT_LANES = 4
LANE_SLAB_BYTES = 36 << 30
#: run streams of the host-fed pipeline (`RaggedSolver(host_io=...)`, `solve_batch_streamed`).  ONE: with two to four the
#: call is no shorter (67.2 / 67.9 / 67.8 ms per 65 536 cube trusses, EXPERIMENTS R6.4) - the run stream is busy from
#: the first pull to the last solve either way, and what bounds the call is that device work on the 232 CUs the copy
#: kernels leave it; `lanes=` stays for other shapes of batch
HOSTFED_LANES = int(os.environ.get("TRS_HOSTFED_LANES", "1"))


def lane_streams(torch, device, count):
    """`count` side streams of `device` for the extra lanes of `RaggedSolver`s (made once per process and device:
    every solver's step forks from and joins the caller's stream, so sharing them only serialises what two callers
    put on the same lane)."""
    have = _LANE_STREAMS.setdefault(str(device), [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device=device))
    return have[:count]


def shared_workspace(torch, device):
    """The process's `SolverWorkspace` of (`device`, CURRENT STREAM) for bucket pipelines that run one after the
    other on that stream (the chunks of `data.dataset_chunks`): one slab for all of them instead of one per call.
    The buffers are re-used without any synchronisation of their own - stream order is the only thing that keeps
    one call's factorisation from the next call's assembly -, hence one workspace per stream: calls issued on
    different streams of a device (or from threads with different current streams) get different buffers.  Two
    threads that drive the SAME stream must serialise their calls themselves, as for any other stream-ordered
    resource.

    The cache is BOUNDED: at most `MAX_SHARED_WORKSPACES` (4; a workspace is up to tens of GB) live at a time PER DEVICE,
    the device's least recently used one is dropped when a further stream of it asks (its buffers go back to the caching allocator once
    the solvers built on it are gone; kernels still queued on them keep them alive through the allocator's
    stream-ordered reuse, as for any freed tensor).  A stream handle value that comes back after its stream was
    destroyed inherits the old entry - harmless: a destroyed stream has no work left, and the buffers carry no state
    between calls."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    ws = _SHARED_WORKSPACES.pop(key, None)
    if ws is None:
        ws = SolverWorkspace(torch, device)
        # The bound is PER DEVICE (a process that drives eight devices in turn keeps every device's workspaces).
        # Dropping a workspace also drops its lanes' buffers, which were allocated on the caller's stream and used on
        # the side streams of the lanes: that is safe only because `RaggedSolver.step` always JOINS its lanes back into
        # the caller's stream before it returns - every use of a lane buffer is ordered before whatever the caller's
        # stream does next, the allocator's stream-ordered reuse included.
        mine = [k for k in _SHARED_WORKSPACES if k[0] == key[0]]       # (dicts keep insertion order: oldest use first)
        for old_key in mine[:max(0, len(mine) - MAX_SHARED_WORKSPACES + 1)]:
            _SHARED_WORKSPACES.pop(old_key)
    _SHARED_WORKSPACES[key] = ws                                       # most recently used last
    return ws


MAX_SHARED_WORKSPACES = 4


def release_workspaces():
    """Drop the shared workspaces of every device and stream (they hold the largest slabs a call needed - up to
    `default_slab_budget`, i.e. up to 60 % of a device's memory over the lanes - for the life of the process) and hand
    the cached blocks back to the driver."""
    import torch
    _SHARED_WORKSPACES.clear()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


class _MaskedStreams:
    """The three streams of the host-fed pipeline with their compute units set apart (`trs_stream_create_masked`):
    pull and push get CUs of their own - consecutive mask bits go round the XCDs, so eight are one CU of every
    XCD - and the solver kernels the rest."""

    def __init__(self, torch, dev, lib, pull_cus, push_cus):
        import ctypes
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        if not (pull_cus > 0 and push_cus > 0 and pull_cus + push_cus < n_cu):
            raise ValueError(f"cannot set {pull_cus} + {push_cus} copy CUs apart on a device with {n_cu}")
        words = (n_cu + 31) // 32
        sets = (range(0, pull_cus), range(pull_cus + push_cus, n_cu), range(pull_cus, pull_cus + push_cus))   # pull, run, push
        self.lib, self.handles, self.streams = lib, [], []
        self.torch, self.dev = torch, dev
        with torch.cuda.device(dev):
            try:
                for k, cus in enumerate(sets):
                    mask = (ctypes.c_uint32 * words)()
                    for c in cus:
                        mask[c // 32] |= 1 << (c % 32)
                    if k == 1:
                        self.run_mask = mask
                    handle = ctypes.c_void_p()
                    _capi.check(lib.trs_stream_create_masked(mask, words, ctypes.byref(handle)), "trs_stream_create_masked")
                    self.handles.append(handle)
                    self.streams.append(torch.cuda.ExternalStream(handle.value, device=dev))
                self.runs = [self.streams[1]]
            except Exception:
                self.close()   # (a runtime that refuses the 2nd or 3rd mask must not leak the streams made so far)
                raise

    def __iter__(self):
        return iter(self.streams[:3])

    def run_streams(self, count):
        """`count` run streams on the solver kernels' CU set (the first is the pipeline's own; more are made on demand
        and kept): the run lanes of a host-fed `RaggedSolver`."""
        import ctypes
        while len(self.runs) < count:
            handle = ctypes.c_void_p()
            with self.torch.cuda.device(self.dev):
                _capi.check(self.lib.trs_stream_create_masked(self.run_mask, len(self.run_mask), ctypes.byref(handle)),
                            "trs_stream_create_masked")
            self.handles.append(handle)
            self.runs.append(self.torch.cuda.ExternalStream(handle.value, device=self.dev))
        return self.runs[:count]

    def close(self):
        """Destroy the streams (after the work queued on them has finished).  The pipeline's own set is created
        once per process and device and lives as long as the process."""
        self.torch.cuda.synchronize(self.dev)
        for handle in self.handles:
            _capi.check(self.lib.trs_stream_destroy(handle), "trs_stream_destroy")
        self.handles, self.streams, self.runs = [], [], []


def _flow_shop_order(groups, packed, n_pad_of, rule="johnson"):
    """Order of the buckets in the host-fed pipeline.  Every bucket is pulled over PCIe, then solved; the link and
    the chip each take the buckets one after the other - a two-machine flow shop, for which Johnson's rule gives the
    shortest makespan: first the buckets that take longer to solve than to pull, by increasing pull time (the chip
    starts early and never runs dry), then the others by decreasing solve time (what is left exposed at the end
    is the smallest solve).  The two times are estimates - live input bytes at the rate a pull reaches beside the
    solves, and count x n_pad^2 at the rate the factorisation of cube trusses runs at - and only their ratio
    matters.  Small buckets are link-bound, large ones chip-bound; smallest-first (`rule="small"`) leaves the chip
    waiting in the first third of the step and the link idle in the last."""
    if rule == "small":
        return sorted(groups, key=lambda idx: len(idx) * n_pad_of(idx) ** 2), 0
    if rule == "large":
        return sorted(groups, key=lambda idx: -len(idx) * n_pad_of(idx) ** 2), len(groups)
    nJ, nM = packed.nJ.astype(np.int64), packed.nM.astype(np.int64)
    pull = lambda idx: float((nJ[idx] * 49 + nM[idx] * 24).sum()) / 45e9
    solve = lambda idx: len(idx) * float(n_pad_of(idx)) ** 2 * 2.0e-12
    chip_bound = sorted((g for g in groups if pull(g) < solve(g)), key=pull)
    link_bound = sorted((g for g in groups if pull(g) >= solve(g)), key=lambda g: -solve(g))
    return chip_bound + link_bound, len(chip_bound)


def _pipeline_streams(torch, dev, lib):
    """(pull, run, push) streams of `RaggedSolver`'s host-fed pipeline.  By default the copy kernels get compute units
    of their own through CU-masked streams (`TRS_PCIE_CUS="pull,push"`, default "16,8" = two CUs per XCD for the pull,
    one for the push: the row copies reach the link's rate on those, `tools/masked_pipeline_check.py`) and the solver
    kernels the rest - 73 instead of 91 ms per host-fed call of the 65 536-truss cube batch, because the copies no
    longer compete with the factorisation for registers on every CU.  `TRS_PCIE_CUS=0` = three ordinary streams.
    (Round 4 had to make the masks opt-in: processes using them died of memory access faults or hung once in five
    runs of the host-fed tests.  That was the joint-order kernel's race, which any concurrent kernel could bring out -
    EXPERIMENTS R5.1; with it fixed, 36 of 36 runs of those tests with the masks passed.)"""
    spec = os.environ.get("TRS_PCIE_CUS", "16,8")
    key = (str(dev), spec)
    if key not in _PIPELINE_STREAMS:   # (a queue with a CU mask takes ~20 ms to create: once per process and device)
        pull_cus, push_cus = int(spec.split(",")[0]), int(spec.split(",")[-1])
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        if pull_cus <= 0 or 2 * (pull_cus + push_cus) > n_cu:   # (a partitioned or small device: no CUs to spare)
            _PIPELINE_STREAMS[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        else:
            try:
                _PIPELINE_STREAMS[key] = _MaskedStreams(torch, dev, lib, pull_cus, push_cus)
            except HipExtensionError as exc:   # (a runtime that refuses CU masks: the pipeline still works, slower)
                import warnings
                warnings.warn(f"CU-masked streams are not available ({exc}); the host-fed pipeline uses ordinary streams")
                _PIPELINE_STREAMS[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    return _PIPELINE_STREAMS[key]


_PIPELINE_STREAMS = {}


class RaggedSolver:
    """A RAGGED batch (trusses of very different sizes: the reference's `GenerateRandomCubeTrusses` loop,
    `generate.py:342-374`, BASELINE config 3) resident on one device in the caller's order and numbering, set
    up once and solved any number of times with everything on the GPU.  Per size bucket (`size_buckets`):

        gather the bucket's rows, trimmed to its own maxima (`trs_copy_rows`)
        -> joint order of the bucket (`trs_joint_order`: found, applied and priced on the device; its LDS tables
           are sized by the BUCKET's maxima, so buckets of small trusses run more work-groups per CU)
        -> `trs_solve`
        -> scatter u / f_ext / N / info back to the caller's rows (`trs_copy_rows`)

    The buckets share ONE workspace (slab, reduced vectors, assembly tables, envelope metadata, the gathered
    un-ordered inputs) sized for the largest of them; their launches follow each other on the current stream.
    Results stay resident (`u`, `f_ext`, `N`, `info`: full-batch tensors in the caller's numbering) until
    `result()` downloads them.  `reorder` as `order_plan`; a host-side plan is carried out once, at set-up."""

    GATHER = ("xyz", "conn", "E", "A", "cbits", "loads", "nJ", "nM")
    GATHER_TABLE = ("xyz", "conn", "type_idx", "cbits", "loads", "nJ", "nM")   # table member form (`PackedBatch.table`)
    JOINT_ORDERED = ("xyz", "conn", "cbits", "loads")

    def __init__(self, packed, device=None, reorder=True, max_slab_bytes=None, granularity=64,
                 options=None, tensors=None, workspace=None, host_io=None, n_variants=1, lanes=None):
        """`packed`: a `PackedBatch` (uploaded here) or, with `tensors` = the batch's device tensors by field name
        (e.g. from `generate.generate_cube_batch_device`), just its `BatchSizes`.  `workspace`: a
        `SolverWorkspace` shared with other solvers that run on the same stream one after the other.

        `lanes` (default `DEFAULT_LANES` = 4; resident batches only): the buckets are dealt onto that many streams -
        lane 0 is the caller's stream, the others fork from it at the start of `step()` and join it at the end, so
        a step still is ONE stream-ordered operation for the caller -, each lane with a workspace of its own, so that
        one bucket's kernels fill the emptying-chip tails and the latency-bound phases of another's: 46.7 -> 43.8 ms
        per step of the 65 536-truss cube batch.  Results are bit for bit those of one lane (asserted step by step:
        `tests/test_gpu_streams.py`).  Rounds 3-4 had to keep this switched off - stalls and bursts of corrupted
        trusses once in a few dozen steps; the cause was a race in `trs_joint_order`'s kernel that only concurrent
        kernels brought out, found with the torch-free reproducer `tools/repro_streams.cpp` (EXPERIMENTS R5.1).
        `max_slab_bytes` (default `default_slab_budget`: `LANE_SLAB_BYTES` = 36 GiB per lane, at most 60 % of the
        device's memory and 85 % of what is free on it now) is the budget of all lanes together.

        `host_io=(inputs, outputs)`: the batch STAYS in page-locked host memory - `inputs` / `outputs` are dicts of
        pinned CPU tensors (the padded arrays of a `PackedBatch.pinned()`; `u`, `f_ext`, `N`, `info` of a
        `ResultPool`).  Page-locked memory is mapped into the device's address space, so every bucket's gather
        PULLS its rows straight out of the host batch and its scatter PUSHES the results into the host arrays
        (full rows, zero padding included); `step()` then runs the buckets as a three-stream pipeline - pull of
        bucket k + 1, device work of bucket k, push of bucket k - 1 at the same time (`solve_batch_streamed`).

        `n_variants` > 1: every step solves the batch that many times with different member sections
        (`step(sections=[...])`, as `solve_batch(sections=...)`): gather and joint order once per bucket, one result
        set per variant in `outs` (`u` / `f_ext` / `N` / `info` are the first variant's)."""
        torch, dev = _require_gpu(device if tensors is None else tensors["xyz"].device)
        self.torch, self.device, self.packed, self.lib = torch, dev, packed, _capi.load()
        B, nJ_max, nM_max = packed.B, packed.nJ_max, packed.nM_max
        self.B = B
        self.host_io = host_io is not None
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        #: table member form (uint16 end joints, a uint8 type index per member, one type table for the batch): the
        #: batch's own form - a `PackedBatch.table()`, host_io inputs or device tensors that carry `type_idx`
        source = host_io[0] if self.host_io else tensors
        self.table = (source.get("type_idx") is not None) if source is not None else bool(getattr(packed, "is_table", False))
        gather = self.GATHER_TABLE if self.table else self.GATHER
        self.types = None
        self._fixed_types = {}   # type tables of fixed-section variants (`_section_types`), by (a, e, density)
        if self.table:
            types = source.get("types") if source is not None and source.get("types") is not None else getattr(packed, "types", None)
            if types is None:
                raise ValueError("table member form: the type table `types` [T, 3] is missing")
            self.types = (types if torch.is_tensor(types) else torch.from_numpy(np.ascontiguousarray(types, dtype=np.float64))).to(dev)
        if self.host_io:
            # (the sizes nJ / nM come from `packed` and go up once, with the bucket index lists: they tell the copy
            # kernels how much of a row is live, so that nothing but live bytes crosses the link)
            self.gather_fields = tuple(f for f in gather if f not in ("nJ", "nM"))
            self.inputs = {f: host_io[0][f] for f in self.gather_fields}
            if not all(t.is_pinned() and t.is_contiguous() for t in self.inputs.values()):
                raise ValueError("host_io inputs must be contiguous page-locked tensors (PackedBatch.pinned())")
        else:
            self.gather_fields = gather
            self.inputs = {f: (tensors[f].contiguous() if tensors is not None else up(getattr(packed, f))) for f in gather}
        plan = order_plan(reorder, nJ_max, nM_max) if B else None
        self.plan = plan
        # A batch whose LARGEST truss does not fit `trs_joint_order` gets a host plan - but its buckets are ordered
        # one by one with tables sized by the bucket, so every bucket that fits is still ordered on the device
        # (unless the caller forced a host order or gave a permutation).
        self.device_effort = plan[1] if plan is not None and plan[0] == "device" else None
        if plan is not None and plan[0] == "host" and (reorder is True or reorder in ("auto", "profile", "fast")):
            self.device_effort = {"fast": 1, "profile": 2}.get(reorder, 3)
        self.ordered = None        # host plan: renumbered xyz / conn / cbits / loads + perm of the FULL batch
        if plan is not None and plan[0] != "device":
            # (found for the whole batch: the buckets that do not fit the device kernel take their rows from it)
            host = (packed if isinstance(packed, PackedBatch) else packed.to_packed(tensors)).general()
            perm = joint_order(host, plan[1])
            renum = permute_joints(host, perm)
            self.ordered = {k: up(getattr(renum, k)) for k in self.JOINT_ORDERED}
            if self.table:
                self.ordered["conn"] = up(np.ascontiguousarray(renum.conn, dtype=np.uint16))
            self.ordered["perm"] = up(perm)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)   # padding beyond a bucket's width stays 0
        if self.host_io:
            out = host_io[1]
            self.u, self.f_ext, self.N, self.info = out["u"], out["f_ext"], out["N"], out["info"]
            want = {"u": (B, nJ_max, 3), "f_ext": (B, nJ_max, 3), "N": (B, nM_max), "info": (B,)}
            if not all(out[k].is_pinned() and out[k].is_contiguous() and tuple(out[k].shape) == want[k] for k in want):
                raise ValueError("host_io outputs must be contiguous page-locked tensors of the padded result shapes")
            # `live` (optional): per result array a device int32 [B] "row r is zero behind byte live[r]"
            # (`ResultPool.take_tracked`); without it every push zero-fills its rows to their full width
            self.live = dict(out.get("live") or {})
            if any(t.device != dev or t.dtype != torch.int32 or tuple(t.shape) != (B,) for t in self.live.values()):
                raise ValueError("host_io live extents must be int32 [B] tensors on the solver's device")
            if n_variants != 1:
                raise ValueError("the host-fed pipeline solves one set of sections per step")
            self.outs = [{"u": self.u, "f_ext": self.f_ext, "N": self.N, "info": self.info}]
        else:
            self.outs = [{"u": z([B, nJ_max, 3], torch.float64), "f_ext": z([B, nJ_max, 3], torch.float64),
                          "N": z([B, nM_max], torch.float64), "info": z([B], torch.int32)}
                         for _ in range(max(1, int(n_variants)))]
            first = self.outs[0]
            self.u, self.f_ext, self.N, self.info = first["u"], first["f_ext"], first["N"], first["info"]
        # (host-fed: `lanes` = RUN streams of the pull / run / push pipeline - bucket k is ordered and solved on run
        # stream k mod lanes with that lane's workspace, so that one bucket's latency-bound phases and emptying-chip
        # tails are filled by the next bucket's kernels, as in the resident step)
        n_lanes = max(1, int(lanes if lanes is not None else (HOSTFED_LANES if self.host_io else DEFAULT_LANES)))
        if max_slab_bytes is None:
            max_slab_bytes = default_slab_budget(torch, dev, n_lanes, workspace)
        # resident batches: groups cut at whole rounds of the factorisation kernel (3 waves x 4 SIMDs per CU in flight)
        quantum = 0 if self.host_io else 12 * int(torch.cuda.get_device_properties(dev).multi_processor_count)
        groups = size_buckets(packed, max_slab_bytes // n_lanes, granularity, quantum=quantum) if B else []
        n_pad_of = lambda idx: (int(packed.n_free[idx].max()) + 63) // 64 * 64
        slab_of = lambda idx: len(idx) * n_pad_of(idx) * (n_pad_of(idx) + 16)
        # largest slab first (the shared workspace is sized once) - in the host-fed pipeline SMALLEST first: the
        # device starts after a short pull, and what is exposed at the end is a small bucket's push
        groups.sort(key=lambda idx: -slab_of(idx))
        if self.host_io:
            groups, self._chip_bound_buckets = _flow_shop_order(groups, packed, n_pad_of,
                                                                os.environ.get("TRS_HOSTFED_ORDER", "johnson"))
        self.buckets = []
        self.lanes = max(1, min(n_lanes, len(groups)))
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        for idx in groups:
            need = {"S": 0, "uf": 0, "work": 0, "env": 0, "raw_j": 0, "raw_m": 0}
            nJ_b, nM_b = max(1, int(packed.nJ[idx].max())), max(1, int(packed.nM[idx].max()))
            n_b, Bb = int(packed.n_free[idx].max()), len(idx)
            sub = {"xyz": e([Bb, nJ_b, 3], torch.float64), "loads": e([Bb, nJ_b, 3], torch.float64),
                   "cbits": e([Bb, nJ_b], torch.uint8), "nJ": e([Bb], torch.int32), "nM": e([Bb], torch.int32)}
            if self.table:
                sub.update(conn=e([Bb, nM_b, 2], torch.uint16), type_idx=e([Bb, nM_b], torch.uint8), types=self.types)
            else:
                sub.update(conn=e([Bb, nM_b, 2], torch.int32), E=e([Bb, nM_b], torch.float64), A=e([Bb, nM_b], torch.float64))
                sub["rho"] = sub["A"]   # placeholder of the right shape: no kernel of the solve reads the densities
            if self.host_io:
                sub["nJ"], sub["nM"] = up(packed.nJ[idx].astype(np.int32)), up(packed.nM[idx].astype(np.int32))
            small = bool(self.lib.trs_solve_small_fits(nJ_b, nM_b, n_b))
            renumbered = plan is not None and not small   # the fused small-system kernel gains nothing from an order
            jout = e([Bb, nJ_b], torch.int32) if renumbered else None
            db = DeviceBatch.from_device(sub, n_b, joint_out=jout)
            opts = dict(options or {})
            compact_rows = opts.pop("compact_rows", None)   # (lo, hi): the compact form for the buckets of lo <= rows < hi
            db.options.update(opts)
            if compact_rows is not None:
                db.options["compact"] = bool(compact_rows[0] <= db.rows < compact_rows[1])
            if not db.small:
                need["S"], need["uf"] = Bb * db.rows * db.ld, Bb * db.rows
                need["work"] = Bb * self.lib.trs_assemble_work_bytes(nJ_b, nM_b, n_b)
                need["env"] = Bb * self.lib.trs_env_ints(n_b)
            on_device = renumbered and self.device_effort is not None and \
                (plan[0] == "device" or bool(self.lib.trs_joint_order_fits(nJ_b, nM_b)))
            # A resident bucket that is ordered on the device needs neither a gather nor a scatter launch: its
            # trs_joint_order_rows reads the trusses' rows straight out of the full batch and its trs_recover_rows
            # writes the results straight into the caller's rows (`fused_io`).
            fused = on_device and not self.host_io and os.environ.get("TRS_RAGGED_FUSED_IO", "1") != "0"
            if on_device and not self.host_io and not fused:   # the bucket's rows in the caller's numbering: input of its trs_joint_order
                need["raw_j"], need["raw_m"] = Bb * nJ_b, Bb * nM_b
            self.buckets.append({"rows": up(np.ascontiguousarray(idx, dtype=np.int64)), "dev": db, "count": Bb,
                                 "renumbered": renumbered, "order_on_device": on_device, "idx": idx, "fused_io": fused,
                                 "reach": e([Bb], torch.int32) if on_device else None, "lane": 0, "need": need,
                                 # microseconds, roughly: what a bucket of the cube batch takes alone on the chip, per
                                 # truss 0.1 + 1.3e-6 rows^2 (EXPERIMENTS R5.6) - the lanes are dealt by it
                                 "cost": Bb * (0.1 + 1.3e-6 * float(db.rows) ** 2)})
        self.workspace = workspace if workspace is not None else SolverWorkspace(torch, dev)
        self._side_streams = lane_streams(torch, dev, self.lanes - 1) if self.lanes > 1 and not self.host_io else []
        self._deal_lanes()
        if self.host_io:
            self._streams = _pipeline_streams(torch, dev, self.lib)
            if isinstance(self._streams, _MaskedStreams):
                self._run_streams = self._streams.run_streams(self.lanes)
            else:
                self._run_streams = [tuple(self._streams)[1]] + list(lane_streams(torch, dev, self.lanes - 1))

    def _deal_lanes(self):
        """Deal the buckets onto the lanes - longest processing time first on `bk["cost"]` (a size model; dealing by MEASURED
        times, and evening out the lanes' ends move by move, were tried: the lanes then end together and the step is
        no shorter - the chip is busy either way, EXPERIMENTS R5.6) -, give every lane a workspace that holds the largest of ITS
        buckets (they run one after the other on its stream) and point the buckets at it.  The buckets are launched in
        the order of `self.buckets`; resident batches keep them by descending cost, so every lane starts with its
        longest bucket and the step ends over the short ones.  Nothing of this solver may be in flight."""
        if self.host_io:   # (the host-fed pipeline has its own order - the flow shop's -: the run lanes take turns)
            for k, bk in enumerate(self.buckets):
                bk["lane"] = k % self.lanes
            self._bind_lanes()
            return
        self.buckets.sort(key=lambda bk: -bk["cost"])
        load = [0.0] * self.lanes
        for bk in self.buckets:
            lane = min(range(self.lanes), key=lambda l: load[l])
            load[lane] += bk["cost"]
            bk["lane"] = lane
        self._bind_lanes()

    def _bind_lanes(self):
        """Workspaces for the lanes as the buckets are dealt now (`bk["lane"]`), the buckets pointed at them."""
        torch, dev = self.torch, self.device
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        needs = [{"S": 0, "uf": 0, "work": 0, "env": 0, "raw_j": 0, "raw_m": 0} for _ in range(self.lanes)]
        for bk in self.buckets:
            for k, v in bk["need"].items():
                needs[bk["lane"]][k] = max(needs[bk["lane"]][k], v)
        lane_bufs = []
        for lane, need in enumerate(needs):
            lane_bufs.append(self.workspace.lane(lane).get(
                {"S": need["S"], "uf": need["uf"], "work": need["work"], "env": need["env"],
                 "raw_xyz": need["raw_j"] * 3, "raw_loads": need["raw_j"] * 3, "raw_cbits": need["raw_j"],
                 "raw_conn": need["raw_m"] * 2}))
        for bk in self.buckets:
            db, Bb = bk["dev"], bk["count"]
            bufs = lane_bufs[bk["lane"]]
            raw = {"xyz": bufs["raw_xyz"], "loads": bufs["raw_loads"], "cbits": bufs["raw_cbits"], "conn": bufs["raw_conn"]}
            if not db.small:
                wb = self.lib.trs_assemble_work_bytes(db.nJ_max, db.nM_max, db.n_max)
                ei = self.lib.trs_env_ints(db.n_max)
                db._slab = (bufs["S"][:Bb * db.rows * db.ld].view(Bb, db.rows, db.ld),
                            bufs["uf"][:Bb * db.rows].view(Bb, db.rows), bufs["work"][:Bb * wb].view(Bb, wb),
                            bufs["env"][:Bb * ei].view(Bb, ei))
            if bk["fused_io"]:
                pass
            elif bk["order_on_device"] and self.host_io:
                # (the pull of bucket k + 1 runs while bucket k is being ordered: every bucket its own buffers)
                if "raw" not in bk:
                    nJ_b, nM_b = db.nJ_max, db.nM_max
                    bk["raw"] = {"xyz": e([Bb, nJ_b, 3], torch.float64), "loads": e([Bb, nJ_b, 3], torch.float64),
                                 "cbits": e([Bb, nJ_b], torch.uint8),
                                 "conn": e([Bb, nM_b, 2], torch.uint16 if self.table else torch.int32),
                                 "nJ": db.nJ, "nM": db.nM}
            elif bk["order_on_device"]:
                nJ_b, nM_b = db.nJ_max, db.nM_max
                bk["raw"] = {"xyz": raw["xyz"][:Bb * nJ_b * 3].view(Bb, nJ_b, 3),
                             "loads": raw["loads"][:Bb * nJ_b * 3].view(Bb, nJ_b, 3),
                             "cbits": raw["cbits"][:Bb * nJ_b].view(Bb, nJ_b),
                             "conn": (raw["conn"][:Bb * nM_b * 2].view(Bb, nM_b, 2) if not self.table else
                                      raw["conn"][:Bb * nM_b].view(torch.uint16).view(Bb, nM_b, 2)),
                             "nJ": db.nJ, "nM": db.nM}
            if bk["order_on_device"]:
                # trs_joint_order writes the renumbered bucket straight into the solver's input tensors
                bk["ordered"] = {"perm": db.joint_out, "reach": bk["reach"], "xyz": db.xyz, "conn": db.conn,
                                 "cbits": db.cbits, "loads": db.loads}
        self._tables = self._copy_tables()

    def _copy_tables(self):
        """ctypes argument arrays of the gather / scatter launches of every bucket (all device pointers are
        fixed at set-up)."""
        import ctypes
        P, Z = ctypes.c_void_p, ctypes.c_size_t
        row_bytes = lambda t: int(t[0].numel() * t.element_size()) if t.dim() > 1 else int(t.element_size())
        tables = []
        for bk in self.buckets:
            db = bk["dev"]
            pairs = []
            for f in (() if bk["fused_io"] else self.gather_fields):
                if bk["order_on_device"] and f in self.JOINT_ORDERED:
                    pairs.append((f, self.inputs[f], bk["raw"][f]))       # caller's numbering -> input of the order
                elif bk["renumbered"] and f in self.JOINT_ORDERED:
                    pairs.append((f, self.ordered[f], getattr(db, f)))     # host plan: renumbered at set-up
                else:
                    pairs.append((f, self.inputs[f], getattr(db, f)))
            if bk["renumbered"] and not bk["order_on_device"]:
                pairs.append(("perm", self.ordered["perm"], db.joint_out))
            outs = [[("u", db.u, o["u"]), ("f_ext", db.f_ext, o["f_ext"]), ("N", db.N, o["N"]), ("info", db.info, o["info"])]
                    for o in self.outs]
            # host-fed: only the live part of a row crosses the link - (count array, bytes per element) by field
            per_joint = {"xyz": 24, "loads": 24, "cbits": 1, "u": 24, "f_ext": 24}
            per_member = {"conn": 4 if self.table else 8, "E": 8, "A": 8, "type_idx": 1, "N": 8}

            def pack(pairs, trimmed_is_dst):
                n = len(pairs)
                src, dst = (P * n)(*[a.data_ptr() for _, a, _ in pairs]), (P * n)(*[b.data_ptr() for _, _, b in pairs])
                sp, dp = (Z * n)(*[row_bytes(a) for _, a, _ in pairs]), (Z * n)(*[row_bytes(b) for _, _, b in pairs])
                width = (Z * n)(*[row_bytes(b if trimmed_is_dst else a) for _, a, b in pairs])
                if not self.host_io:
                    return n, src, sp, dst, dp, width, None, None, None, None
                counts = (P * n)(*[db.nJ.data_ptr() if f in per_joint else (db.nM.data_ptr() if f in per_member else None)
                                   for f, _, _ in pairs])
                elem = (Z * n)(*[per_joint.get(f, per_member.get(f, 0)) for f, _, _ in pairs])
                # pulled rows are zeroed behind their live part on the device; pushed rows up to the full row of the
                # host array, or - where the array's live extents are tracked - only over what the last writer left
                fill = (Z * n)(*[row_bytes(b) for _, _, b in pairs])
                live = None if trimmed_is_dst else (P * n)(*[self.live[f].data_ptr() if f in self.live else None
                                                             for f, _, _ in pairs])
                return n, src, sp, dst, dp, width, fill, counts, elem, live
            tables.append((pack(pairs, True) if pairs else None, [pack(o, False) for o in outs]))
        return tables

    def step(self, record=None, sections=None):
        """One pass of the whole path over the batch, asynchronous on the current stream.  `record` (a list):
        instrumented step - (stage name, start event, end event) of every bucket's gather, order, solve and
        scatter are appended.  `sections`: one entry per variant of the solver (`n_variants`) - None = the
        members' own sections, (A, E, density) = every member set to that type; results in `outs[slot]`."""
        torch = self.torch
        if self.B == 0:
            return
        sections = [None] * len(self.outs) if sections is None else list(sections)
        if len(sections) != len(self.outs):
            raise ValueError(f"{len(sections)} section variants for a solver built for {len(self.outs)}")
        # the variants with the members' own sections first: they need what the gather brought
        slots = sorted(range(len(sections)), key=lambda k: sections[k] is not None)

        def timed(name, call):   # (events go onto the CURRENT stream: the bucket's lane)
            if record is None:
                return call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = call()
            e1.record()
            record.append((name, e0, e1))
            return out

        # work-groups of a copy kernel whose other side is host memory (it needs bytes in flight, not CUs)
        # (the pushes are kept to a handful of work-groups: stores to host memory are fire-and-forget, and beyond
        # ~35 GB/s they back up in the queues the solver's own stores stand in - its step beside a saturating
        # push takes 89 instead of 60 ms, beside one of four work-groups 64, `tools/masked_pipeline_check.py`; three
        # work-groups while the chip-bound buckets are being solved, eight for the small ones at the end, is where
        # the pushes still keep up with the solves of the cube batch)
        masked = self.host_io and isinstance(self._streams, _MaskedStreams)
        PCIE_BLOCKS = int(os.environ.get("TRS_PCIE_BLOCKS", "64" if masked else "32"))
        push_spec = os.environ.get("TRS_PCIE_PUSH_BLOCKS", "3,8" if masked else str(PCIE_BLOCKS)).split(",")
        n_chip = getattr(self, "_chip_bound_buckets", 0)   # "a,b": a for the chip-bound buckets, b for the others
        push_blocks = lambda k: int(push_spec[0] if (k < n_chip or len(push_spec) == 1) else push_spec[1])
        with torch.cuda.device(self.device):
            if not self.host_io:
                caller = torch.cuda.current_stream(self.device)
                lanes = [caller] + list(self._side_streams)
                self._lane_events = []   # (kept until the next step: none is destroyed while a stream may still wait on it)
                if len(lanes) > 1:   # fork: the side lanes start behind everything queued on the caller's stream
                    fork = torch.cuda.Event()
                    fork.record(caller)
                    self._lane_events.append(fork)
                    for side in lanes[1:]:
                        side.wait_event(fork)
                try:
                    self._step_resident(lanes, sections, slots, timed)
                finally:
                    for side in lanes[1:]:   # join: what follows on the caller's stream sees every lane's results
                        done = torch.cuda.Event()
                        done.record(side)
                        caller.wait_event(done)
                        self._lane_events.append(done)
                return
            if sections[0] is not None:
                raise ValueError("the host-fed pipeline solves the members' own sections")
            # host-fed pipeline: pull of bucket k + 1 | order + solve of bucket k | push of bucket k - 1
            s_up, _, s_down = tuple(self._streams)
            runs = self._run_streams
            caller = torch.cuda.current_stream(self.device)
            timing = record is not None
            begin = torch.cuda.Event(enable_timing=timing)
            begin.record(caller)
            for st in [s_up, s_down] + list(runs):
                st.wait_event(begin)
            mark = lambda name, stream: record.append((name, begin, self._marked(stream))) if timing else None
            for k, (bk, (gather, scatters)) in enumerate(zip(self.buckets, self._tables)):
                scatter = scatters[0]
                with torch.cuda.stream(s_up):
                    mark(f"bucket {k} pull begins", s_up)
                    _capi.check(self.lib.trs_copy_rows(*gather, bk["count"], bk["rows"].data_ptr(), 0, PCIE_BLOCKS,
                                                       s_up.cuda_stream), "trs_copy_rows (pull)")
                    pulled = torch.cuda.Event()
                    pulled.record(s_up)
                    mark(f"bucket {k} pulled", s_up)
                s_run = runs[bk["lane"]]
                with torch.cuda.stream(s_run):
                    s_run.wait_event(pulled)
                    mark(f"bucket {k} run begins", s_run)
                    if bk["order_on_device"]:
                        joint_order_device(torch, bk["raw"], effort=self.device_effort, out=bk["ordered"])
                    bk["dev"].solve()
                    solved = torch.cuda.Event()
                    solved.record(s_run)
                    mark(f"bucket {k} solved", s_run)
                with torch.cuda.stream(s_down):
                    s_down.wait_event(solved)
                    mark(f"bucket {k} push begins", s_down)
                    _capi.check(self.lib.trs_copy_rows(*scatter, bk["count"], bk["rows"].data_ptr(), 1, push_blocks(k),
                                                       s_down.cuda_stream), "trs_copy_rows (push)")
                    mark(f"bucket {k} pushed", s_down)
            done = torch.cuda.Event()
            done.record(s_down)
            caller.wait_event(done)   # work queued on the caller's stream after step() sees the results

    def _step_resident(self, lanes, sections, slots, timed):
        """The buckets of a resident batch, each on the stream of its lane (`lanes[0]` = the caller's)."""
        torch = self.torch
        nJ_full, nM_full = int(self.u.shape[1]), int(self.N.shape[1])
        for bk, (gather, scatters) in zip(self.buckets, self._tables):
            with torch.cuda.stream(lanes[bk["lane"]]):
                stream = lanes[bk["lane"]].cuda_stream
                if bk["fused_io"]:
                    db, inp, ordr = bk["dev"], self.inputs, bk["ordered"]
                    if self.table:
                        sec_in, sec_out = (inp["type_idx"].data_ptr(),), (db.type_idx.data_ptr(),)
                        order_rows, what = self.lib.trs_joint_order_rows_tab, "trs_joint_order_rows_tab"
                    else:
                        sec_in, sec_out = (inp["E"].data_ptr(), inp["A"].data_ptr()), (db.E.data_ptr(), db.A.data_ptr())
                        order_rows, what = self.lib.trs_joint_order_rows, "trs_joint_order_rows"
                    timed("order", lambda: _capi.check(order_rows(
                        bk["count"], db.nJ_max, db.nM_max, bk["rows"].data_ptr(), int(inp["xyz"].shape[1]),
                        int(inp["conn"].shape[1]), inp["xyz"].data_ptr(), inp["conn"].data_ptr(),
                        inp["cbits"].data_ptr(), inp["loads"].data_ptr(), *sec_in,
                        inp["nJ"].data_ptr(), inp["nM"].data_ptr(), ordr["perm"].data_ptr(), ordr["reach"].data_ptr(),
                        db.xyz.data_ptr(), db.conn.data_ptr(), db.cbits.data_ptr(), db.loads.data_ptr(),
                        *sec_out, db.nJ.data_ptr(), db.nM.data_ptr(),
                        int(self.device_effort), stream), what))
                    for slot in slots:
                        types = self._section_types(db, sections[slot])
                        timed("solve", lambda: db.solve_rows(bk["rows"], self.outs[slot], nJ_full, nM_full, types=types))
                    continue
                timed("gather", lambda: _capi.check(self.lib.trs_copy_rows(
                    *gather, bk["count"], bk["rows"].data_ptr(), 0, 0, stream), "trs_copy_rows (gather)"))
                if bk["order_on_device"]:
                    timed("order", lambda: joint_order_device(torch, bk["raw"], effort=self.device_effort, out=bk["ordered"]))
                for slot in slots:
                    types = self._section_types(bk["dev"], sections[slot])
                    timed("solve", lambda: bk["dev"].solve(types=types))
                    timed("scatter", lambda: _capi.check(self.lib.trs_copy_rows(
                        *scatters[slot], bk["count"], bk["rows"].data_ptr(), 1, 0, stream), "trs_copy_rows (scatter)"))

    def _section_types(self, db, section):
        """A section variant of a step (`section` = None: the members' own, or (a, e, density) for every member): the
        general form overwrites the bucket's A and E, the table form solves with another type table - every row the
        fixed triple - and leaves the bucket's type indices alone.  Returns the table to pass on, or None."""
        if section is None:
            return None
        if not db.table:
            db.A.fill_(float(section[0]))
            db.E.fill_(float(section[1]))
            return None
        key = tuple(float(v) for v in section[:3]) + (0.0,) * (3 - len(section[:3]))
        cache = self._fixed_types
        if key not in cache:
            row = self.torch.tensor(key, dtype=self.torch.float64, device=self.device)
            cache[key] = row.expand(max(1, int(self.types.shape[0])), 3).contiguous()
        return cache[key]

    def _marked(self, stream):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def adopt_launch_hints(self):
        """After a step: read back the largest envelope reach the device order reported per bucket (one scalar
        each) and tell every bucket whose envelopes all stay within the wave-per-matrix kernels' range to skip the
        launches that would find no matrix (`DeviceBatch.all_narrow`).  The order is a function of the resident
        inputs, so the hint holds for every later step; it is safe in any case (a hinted batch is routed narrow
        on the device regardless).  Returns the number of buckets hinted."""
        hinted = 0
        for bk in self.buckets:
            db = bk["dev"]
            if bk["order_on_device"] and not db.small and db.use_envelope:
                db.all_narrow = bool(int(bk["reach"].max().item()) <= NARROW_MAX_BELOW)
                hinted += int(db.all_narrow)
        return hinted

    def result(self):
        """Synchronise and download the dense results (caller's order and numbering); in the host-fed mode they
        are in the caller's page-locked arrays already."""
        self.torch.cuda.synchronize(self.device)
        if self.host_io:
            return BatchResult(self.u.numpy(), self.f_ext.numpy(), self.N.numpy(), self.info.numpy())
        return BatchResult(self.u.cpu().numpy(), self.f_ext.cpu().numpy(), self.N.cpu().numpy(),
                           self.info.cpu().numpy())


@dataclass
class DeviceResult:
    """Dense results of a batched solve left on the device (torch tensors, caller's joint order):
    displace / external [B,nJ_max,3] f64, internal [B,nM_max] f64, info [B] i32; `inputs` maps
    DeviceBatch.INPUT_FIELDS to the resident input tensors (shared by the results of one call)."""
    displace: object
    external: object
    internal: object
    info: object
    inputs: dict


def _solve_small_host(packed: PackedBatch, torch, dev, variants, on_device=False):
    """Host arrays in -> host results out for a batch that qualifies for the fused small-system kernel
    as a whole: ONE upload (every input packed into one byte buffer), one `trs_solve_small` launch per
    section variant, ONE download.  This is what `Truss.Solve()` of a small truss costs: two copies and
    a kernel.  Returns a list of `BatchResult`, one per variant (`on_device`: of `DeviceResult`, nothing
    is downloaded)."""
    lib = _capi.load()
    B, nJm, nMm = packed.B, packed.nJ_max, packed.nM_max
    ins = [("xyz", packed.xyz, np.float64), ("loads", packed.loads, np.float64), ("E", packed.E, np.float64),
           ("A", packed.A, np.float64), ("rho", packed.rho, np.float64), ("conn", packed.conn, np.int32),
           ("nJ", packed.nJ, np.int32), ("nM", packed.nM, np.int32), ("cbits", packed.cbits, np.uint8)]
    off, total = {}, 0
    for name, arr, dt in ins:
        off[name] = total
        total += (arr.size * np.dtype(dt).itemsize + 15) // 16 * 16
    own = len(variants) > 1 or variants[0] is not None
    if own:   # the solves read A_var / E_var; the batch's own sections stay in A / E
        off["A_var"], off["E_var"] = total, total + (packed.A.size * 8 + 15) // 16 * 16
        total = off["E_var"] + (packed.A.size * 8 + 15) // 16 * 16
    host = np.empty([total], dtype=np.uint8)
    for name, arr, dt in ins:
        host[off[name]: off[name] + arr.size * np.dtype(dt).itemsize].view(dt)[:] = arr.reshape(-1)
    din = torch.from_numpy(host).to(dev)
    nu, nn = B * nJm * 3 * 8, B * nMm * 8
    per = 2 * nu + nn + (B * 4 + 15) // 16 * 16
    dout = torch.empty([len(variants) * per], dtype=torch.uint8, device=dev)
    pin, pout = din.data_ptr(), dout.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    nsec = B * nMm * 8
    with torch.cuda.device(dev):
        for slot, sec in enumerate(variants):
            a_ptr, e_ptr = pin + off["A"], pin + off["E"]
            if own:
                va = din[off["A_var"]: off["A_var"] + nsec].view(torch.float64)
                ve = din[off["E_var"]: off["E_var"] + nsec].view(torch.float64)
                if sec is None:
                    va.copy_(din[off["A"]: off["A"] + nsec].view(torch.float64))
                    ve.copy_(din[off["E"]: off["E"] + nsec].view(torch.float64))
                else:
                    va.fill_(float(sec[0])); ve.fill_(float(sec[1]))
                a_ptr, e_ptr = pin + off["A_var"], pin + off["E_var"]
            o = pout + slot * per
            _capi.check(lib.trs_solve_small(
                B, nJm, nMm, packed.n_max, pin + off["xyz"], pin + off["conn"], e_ptr, a_ptr,
                pin + off["cbits"], pin + off["loads"], pin + off["nJ"], pin + off["nM"], o, o + nu,
                o + 2 * nu, o + 2 * nu + nn, None, None, None, 0.0, 0.0, None, None, None, stream),
                "trs_solve_small")
    if on_device:
        tdt = {np.float64: torch.float64, np.int32: torch.int32, np.uint8: torch.uint8}
        inputs = {name: din[off[name]: off[name] + arr.size * np.dtype(dt).itemsize].view(tdt[dt]).view(arr.shape)
                  for name, arr, dt in ins}
        out = []
        for slot in range(len(variants)):
            d = dout[slot * per: (slot + 1) * per]
            out.append(DeviceResult(d[:nu].view(torch.float64).view(B, nJm, 3),
                                    d[nu: 2 * nu].view(torch.float64).view(B, nJm, 3),
                                    d[2 * nu: 2 * nu + nn].view(torch.float64).view(B, nMm),
                                    d[2 * nu + nn: 2 * nu + nn + 4 * B].view(torch.int32), inputs))
        return out
    hout = dout.cpu().numpy()
    results = []
    for slot in range(len(variants)):
        h = hout[slot * per: (slot + 1) * per]
        results.append(BatchResult(h[:nu].view(np.float64).reshape(B, nJm, 3),
                                   h[nu: 2 * nu].view(np.float64).reshape(B, nJm, 3),
                                   h[2 * nu: 2 * nu + nn].view(np.float64).reshape(B, nMm),
                                   h[2 * nu + nn: 2 * nu + nn + 4 * B].view(np.int32).copy()))
    return results


class ResultPool:
    """Page-locked host buffers for the results of `solve_batch(..., pool=...)`, reused from call to call:
    the download is one DMA per array instead of a copy into freshly page-faulted memory.  The arrays of a
    returned `BatchResult` are views of the pool and stay valid until the pool serves another call.

    `tracked=True` (opt-in): the host-fed pipeline (`solve_batch_streamed`) keeps LIVE EXTENTS of the result arrays
    (`take_tracked`) and pushes only live bytes - the zero padding crosses the link once, when the arrays are made.
    The contract that comes with it: the returned arrays are READ-ONLY views; a caller who writes into them (say
    `res.displace += x`) must call `invalidate()` before the pool's next use, or later results carry stale bytes in
    their padding.  Without it every push zero-fills its rows to their full width (more bytes over PCIe, no
    contract)."""

    def __init__(self, tracked=False):
        self._bufs = {}
        self._live = {}
        self.tracked = bool(tracked)

    def take(self, torch, key, shape, dtype):
        self._live.pop(key, None)   # whoever takes the plain buffer may write anything anywhere in it
        buf = self._bufs.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            self._bufs[key] = buf
        return buf

    def take_tracked(self, torch, key, shape, dtype, device):
        """The buffer together with its LIVE EXTENTS: a device int32 tensor, one entry per row, that states "row r
        is zero behind byte live[r]".  `trs_copy_rows` keeps the statement true while it pushes results into the
        rows (it zeroes only what the previous writer of a row left and records the new extent), so the zero
        padding of the padded result arrays is paid for once - the buffer is created zeroed - instead of crossing
        the link with every call.  Anyone else who writes into the buffer must take it with `take` (or call
        `invalidate`): the next tracked use then starts from "nothing known" and zero-fills whole rows once."""
        shape = tuple(shape)
        row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
        buf, live = self._bufs.get(key), self._live.get(key)
        if buf is None or tuple(buf.shape) != shape or buf.dtype != dtype:
            buf = torch.zeros(shape, dtype=dtype, pin_memory=True)
            self._bufs[key] = buf
            live = torch.zeros([shape[0]], dtype=torch.int32, device=device)
        elif live is None or live.device != torch.device(device) or live.shape[0] != shape[0]:
            live = torch.full([shape[0]], row_bytes, dtype=torch.int32, device=device)
        self._live[key] = live
        return buf, live

    def invalidate(self):
        """Forget the live extents (after writing into pool buffers by hand)."""
        self._live.clear()


def host_result_arrays(torch, pool, B, nJ_max, nM_max, device):
    """The page-locked result arrays of a host-fed `RaggedSolver` (`host_io=(inputs, THIS)`) out of a
    `ResultPool`, with their live extents."""
    out, live = {}, {}
    for name, shape in (("u", [B, nJ_max, 3]), ("f_ext", [B, nJ_max, 3]), ("N", [B, nM_max])):
        if pool.tracked:   # (the caller opted into the read-only contract of tracked extents)
            out[name], live[name] = pool.take_tracked(torch, (0, name), shape, torch.float64, device)
        else:
            out[name] = pool.take(torch, (0, name), shape, torch.float64)
    out["info"] = pool.take(torch, (0, "info"), [B], torch.int32)
    out["live"] = live
    return out


def solve_batch_streamed(packed: PackedBatch, device=None, reorder=True, pool=None, max_slab_bytes=48 << 30, lanes=None):
    """Host arrays in -> host results out for a LARGE ragged batch held in page-locked memory
    (`PackedBatch.pinned()`), as a pipeline over PCIe with NO staging copy: the batch stays where it is, every
    size bucket's gather pulls its rows straight out of the host arrays (page-locked memory is mapped into the
    device's address space; only the bucket-trimmed prefix of every row crosses the link - for cube trusses
    about 60 % of the padded bytes), the device orders and solves the bucket, and its scatter pushes the results
    into the (page-locked) result arrays - pull of bucket k + 1, device work of bucket k and push of bucket k - 1
    at the same time on three streams (`RaggedSolver(host_io=...)`; `lanes` = run streams, default `HOSTFED_LANES`: the
    buckets' device work alternates between them).  Same results, bit for bit, as `solve_batch(packed, reorder=...)`.  `solve_batch(..., pool=...)` routes big pinned batches here by itself."""
    torch, dev = _require_gpu(device)
    B, nJ_max, nM_max = packed.B, packed.nJ_max, packed.nM_max
    if B == 0:
        return BatchResult(np.zeros([0, nJ_max, 3]), np.zeros([0, nJ_max, 3]), np.zeros([0, nM_max]),
                           np.zeros([0], dtype=np.int32))
    pool = pool if pool is not None else ResultPool()
    host_out = host_result_arrays(torch, pool, B, nJ_max, nM_max, dev)
    host_in = {f: torch.from_numpy(getattr(packed, f)) for f in (RaggedSolver.GATHER_TABLE if packed.is_table else RaggedSolver.GATHER)}
    if packed.is_table:   # (5 instead of 24 bytes per member cross the link: conn uint16 + one type index)
        host_in["types"] = torch.from_numpy(np.ascontiguousarray(packed.types, dtype=np.float64))
    solver = RaggedSolver(packed, dev, reorder=reorder, max_slab_bytes=max_slab_bytes, host_io=(host_in, host_out),
                          lanes=lanes)
    solver.step()
    return solver.result()


def _is_pinned(packed):
    """Every solver input of a `PackedBatch` is a contiguous array in page-locked memory."""
    import torch
    try:
        return all(getattr(packed, f).flags["C_CONTIGUOUS"] and torch.from_numpy(getattr(packed, f)).is_pinned()
                   for f in (RaggedSolver.GATHER_TABLE if packed.is_table else RaggedSolver.GATHER))
    except (RuntimeError, TypeError, ValueError):
        return False


#: `solve_batch(..., pool=...)` hands batches of at least this many trusses to `solve_batch_streamed`
STREAMED_FROM = 16384


def solve_batch(trusses_or_packed, device=None, max_slab_bytes=64 << 30, reorder=False, sections=None,
                on_device=False, pool=None, options=None, device_inputs=None):
    """Solve many trusses in device pipelines.  Accepts `list[Truss]` or a `PackedBatch`.

    The packed inputs go up once; a ragged batch is bucketed by padded system size
    (`size_buckets`) and every bucket is gathered, solved and scattered back ON THE DEVICE
    (`index_select` / `index_copy_`), inputs trimmed to the bucket's own maxima; the dense results come
    down once.  `reorder=True` renumbers the joints of every truss first, by the cheapest of reverse
    Cuthill-McKee and coordinate sweeps (`order_plan`: True / "profile", "fast", "rcm", ...): found, applied
    and undone (`trs_recover`'s `joint_out`) ON THE DEVICE by `trs_joint_order` whenever the batch shape fits
    that kernel, otherwise found on the host (natively, while the inputs go up).  Worth it when the trusses
    are not numbered along their long axis, e.g. generated cube trusses.

    `sections=[None, (a, e, density), ...]` solves the same trusses several times - `None` with their
    own member sections, a triple with every member set to it (the "fixed member type" prior of the
    reference's dataset path, `data.py:107-114`) - and returns a list of results: the geometry is
    uploaded, reordered and bucketed once, only A and E change between the solves.

    `on_device=True` leaves the results on the GPU (`DeviceResult`, torch tensors in the caller's joint
    order, plus the resident inputs) for device-side consumers such as the graph-feature kernel.  With
    `device_inputs` such a call runs in the `shared_workspace` of (device, current stream): its kernels - and
    the tensors it returns - are ordered by THAT stream only; a consumer on another stream must wait for it
    (`stream.wait_stream`), and callers that drive one stream from several threads must serialise their calls.

    Host side of a large batch: the joint order is found on a worker thread while the inputs go up; a
    `PackedBatch.pinned()` uploads by DMA; `pool=ResultPool()` downloads into reused page-locked buffers
    (the returned arrays are then views of the pool, valid until its next use).  `options`: per-call switches
    of the pipeline (`DEFAULT_OPTIONS`), for A/B runs and tests.  `device_inputs`: the batch's arrays already
    live on the device (dict by field name, e.g. `generate.generate_cube_batch_device`) - nothing is uploaded and
    the first argument only carries the sizes (`BatchSizes` or a `PackedBatch`).

    A `PackedBatch` in the TABLE member form (`PackedBatch.table()`, `pack_json(..., members="auto")`: uint16 end joints,
    a uint8 type index per member, one type table) goes up as it is - 5 instead of 24 bytes per member - and is solved
    by the resident bucket pipeline (`RaggedSolver`) through the `_tab` entry points: the same bits as the general form."""
    packed = trusses_or_packed if isinstance(trusses_or_packed, (PackedBatch, BatchSizes)) \
        else pack_trusses(list(trusses_or_packed))
    torch, dev = _require_gpu(device if device_inputs is None else device_inputs["xyz"].device)
    B, nJ_max, nM_max = packed.B, packed.nJ_max, packed.nM_max
    variants = [None] if sections is None else list(sections)
    if isinstance(packed, PackedBatch) and packed.is_table:
        if on_device or device_inputs is not None:
            packed = packed.general()   # (device-side consumers - graph features, fitness - read the general form)
        elif not (pool is not None and B >= STREAMED_FROM and sections is None and options is None and _is_pinned(packed)
                  and not _capi.load().trs_solve_small_fits(nJ_max, nM_max, packed.n_max)):
            # the table member form goes through the resident bucket pipeline as it is: 5 bytes per member up,
            # every kernel reads the table (same bits as the general form)
            if B == 0:
                empty = BatchResult(np.zeros([0, nJ_max, 3]), np.zeros([0, nJ_max, 3]), np.zeros([0, nM_max]),
                                    np.zeros([0], dtype=np.int32))
                return empty if sections is None else [empty for _ in variants]
            solver = RaggedSolver(packed, dev, reorder=reorder if reorder is not False else None,
                                  max_slab_bytes=None if max_slab_bytes >= 64 << 30 else max_slab_bytes,
                                  options=options, n_variants=len(variants))
            solver.step(sections=variants)
            torch.cuda.synchronize(dev)
            results = [BatchResult(o["u"].cpu().numpy(), o["f_ext"].cpu().numpy(), o["N"].cpu().numpy(),
                                   o["info"].cpu().numpy()) for o in solver.outs]
            return results[0] if sections is None else results
    if (pool is not None and B >= STREAMED_FROM and sections is None and not on_device and device_inputs is None
            and isinstance(packed, PackedBatch) and options is None and _is_pinned(packed)
            and not _capi.load().trs_solve_small_fits(nJ_max, nM_max, packed.n_max)):
        return solve_batch_streamed(packed, dev, reorder=reorder, pool=pool, max_slab_bytes=min(max_slab_bytes, 48 << 30))
    if B and device_inputs is None and _capi.load().trs_solve_small_fits(nJ_max, nM_max, packed.n_max):
        # every truss is small: the fused kernel, no bucketing, no reordering (nothing to gain from it)
        out = _solve_small_host(packed, torch, dev, variants, on_device)
        return out[0] if sections is None else out
    if device_inputs is not None and on_device and B and options is None:
        groups = size_buckets(packed, min(max_slab_bytes, 48 << 30))
        if len(groups) > 1:
            # a ragged batch that is on the device already and stays there: the resident bucket pipeline
            # (one-launch gathers and scatters, joint order per bucket, one workspace shared by all buckets and by
            # all calls on this device), one result set per section variant
            solver = RaggedSolver(packed, dev, reorder=reorder, tensors=device_inputs,
                                  max_slab_bytes=None if max_slab_bytes >= 64 << 30 else max_slab_bytes,
                                  workspace=shared_workspace(torch, dev), n_variants=len(variants))
            solver.step(sections=variants)
            inputs = {f: device_inputs[f] for f in DeviceBatch.INPUT_FIELDS if f in device_inputs}
            results = [DeviceResult(o["u"], o["f_ext"], o["N"], o["info"], inputs) for o in solver.outs]
            return results[0] if sections is None else results
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
    plan = order_plan(reorder, nJ_max, nM_max) if B else None
    ordering = None
    if plan is not None and plan[0] == "host" and B >= 1024 and isinstance(packed, PackedBatch) and device_inputs is None:
        # native code (the GIL is released): the order is found while the inputs go up
        from concurrent.futures import ThreadPoolExecutor
        worker = ThreadPoolExecutor(max_workers=1)
        ordering = worker.submit(joint_order, packed, plan[1])
    # the densities play no part in the solve (weights and graph features only): they go up for on-device
    # consumers; otherwise the field is not transferred at all (a fifth of the upload of a cube-truss batch)
    needs_rho = on_device
    if device_inputs is not None:
        full = {f: device_inputs[f].contiguous() for f in DeviceBatch.INPUT_FIELDS if f in device_inputs}
        if plan is not None and plan[0] == "host" and not isinstance(packed, PackedBatch):
            packed_host = packed.to_packed(device_inputs)   # a host-side order needs the arrays on the host
            ordering = None
            plan = ("given", joint_order(packed_host, plan[1]))
    else:
        full = {f: up(getattr(packed, f)) for f in DeviceBatch.INPUT_FIELDS if f != "rho" or needs_rho}
    original = dict(full)   # the caller's joint order (the reordering below makes new tensors)
    perm32 = None
    if plan is not None and plan[0] == "device":
        # found and applied on the GPU, behind the upload on the same stream: no host pass over the batch
        ordered = joint_order_device(torch, full, effort=plan[1])
        perm32 = ordered["perm"]                                             # [B, nJ_max] int32, joint k := old perm[k]
        full.update({k: ordered[k] for k in ("xyz", "conn", "cbits", "loads")})
    elif plan is not None:
        if ordering is not None:
            host_perm = ordering.result()
            worker.shutdown(wait=False)
        else:
            host_perm = joint_order(packed, plan[1])
        perm32 = up(host_perm)
        perm = perm32.long()
        inverse = torch.empty_like(perm)
        inverse.scatter_(1, perm, torch.arange(nJ_max, device=dev).expand(B, -1))
        by_joint = perm[:, :, None].expand(-1, -1, 3)
        full["xyz"] = full["xyz"].gather(1, by_joint)
        full["loads"] = full["loads"].gather(1, by_joint)
        full["cbits"] = full["cbits"].gather(1, perm)
        conn = inverse.gather(1, full["conn"].long().reshape(B, -1)).reshape(B, nM_max, 2)
        live = torch.arange(nM_max, device=dev)[None, :, None] < full["nM"].long()[:, None, None]
        full["conn"] = (conn * live).to(torch.int32)
    groups = size_buckets(packed, max_slab_bytes)
    # largest slab first: the caching allocator then serves every later bucket from the first
    # bucket's block instead of growing the pool bucket by bucket
    n_pad_of = lambda idx: (int(packed.n_free[idx].max()) + 63) // 64 * 64
    groups.sort(key=lambda idx: -len(idx) * n_pad_of(idx) * (n_pad_of(idx) + 16))
    whole = len(groups) == 1 and len(groups[0]) == B
    outs = []
    for _ in variants:
        z = torch.zeros if not whole else torch.empty
        outs.append({"u": z([B, nJ_max, 3], dtype=torch.float64, device=dev),
                     "f_ext": z([B, nJ_max, 3], dtype=torch.float64, device=dev),
                     "N": z([B, nM_max], dtype=torch.float64, device=dev),
                     "info": z([B], dtype=torch.int32, device=dev)})
    joint_fields = ("xyz", "cbits", "loads")
    lib = _capi.load()
    for idx in groups:
        n_b = int(packed.n_free[idx].max()) if not whole else packed.n_max
        if whole:
            rows, nJ_b, nM_b = None, nJ_max, nM_max
        else:
            rows = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(dev)
            nJ_b = max(1, int(packed.nJ[idx].max()))
            nM_b = max(1, int(packed.nM[idx].max()))
        # a bucket of small trusses runs on the fused kernel, which takes no joint order (and gains
        # nothing from one): it reads the caller's numbering
        renumbered = perm32 is not None and not lib.trs_solve_small_fits(nJ_b, nM_b, n_b)
        source = full if renumbered else original
        if whole:
            sub = dict(source)
        else:
            sub = {}
            for f in DeviceBatch.INPUT_FIELDS:
                if f not in source:
                    continue
                t = source[f].index_select(0, rows)
                if f in ("nJ", "nM"):
                    sub[f] = t
                else:  # trimmed to the bucket's own maxima
                    sub[f] = t[:, :(nJ_b if f in joint_fields else nM_b)].contiguous()
        if "rho" not in sub:
            sub["rho"] = sub["A"]   # placeholder of the right shape: no kernel of the solve reads it
        own = (sub["A"], sub["E"]) if len(variants) > 1 else None
        if own is not None:   # the bucket's own sections survive the fixed-section solves
            sub["A"], sub["E"] = own[0].clone(), own[1].clone()
        # the bucket's kernels see the renumbered joints; trs_recover writes the results of joint k to the
        # caller's row perm[k] (a real joint's target is a real joint, so the trimmed width holds it)
        jout = None
        if renumbered:
            jout = perm32 if whole else perm32.index_select(0, rows)[:, :nJ_b].contiguous()
        # (no launch hints here: finding the envelopes' reach on the host costs a one-shot call more than
        # the handful of empty launches it would save; a resident DeviceBatch does it once and keeps it)
        bucket = DeviceBatch.from_device(sub, n_b, joint_out=jout)
        bucket.options.update(options or {})
        for slot, sec in enumerate(variants):
            if sec is not None:
                bucket.A.fill_(float(sec[0]))
                bucket.E.fill_(float(sec[1]))
            elif own is not None:
                bucket.A.copy_(own[0])
                bucket.E.copy_(own[1])
            bucket.solve()
            o = outs[slot]
            if whole:
                o["u"].copy_(bucket.u); o["f_ext"].copy_(bucket.f_ext); o["N"].copy_(bucket.N)
                o["info"].copy_(bucket.info)
            else:
                o["u"][:, :nJ_b].index_copy_(0, rows, bucket.u)
                o["f_ext"][:, :nJ_b].index_copy_(0, rows, bucket.f_ext)
                o["N"][:, :nM_b].index_copy_(0, rows, bucket.N)
                o["info"].index_copy_(0, rows, bucket.info)
        del bucket
    results = []
    if on_device:
        results = [DeviceResult(o["u"], o["f_ext"], o["N"], o["info"], original) for o in outs]
        return results[0] if sections is None else results
    if pool is not None:   # one DMA per array into the pool's page-locked buffers
        host = []
        for slot, o in enumerate(outs):
            h = {k: pool.take(torch, (slot, k), v.shape, v.dtype) for k, v in o.items()}
            for k, v in o.items():
                h[k].copy_(v, non_blocking=True)
            host.append(h)
        torch.cuda.synchronize(dev)
        for h in host:
            results.append(BatchResult(h["u"].numpy(), h["f_ext"].numpy(), h["N"].numpy(), h["info"].numpy()))
        return results[0] if sections is None else results
    torch.cuda.synchronize(dev)
    for o in outs:
        results.append(BatchResult(o["u"].cpu().numpy(), o["f_ext"].cpu().numpy(), o["N"].cpu().numpy(),
                                   o["info"].cpu().numpy()))
    return results[0] if sections is None else results
