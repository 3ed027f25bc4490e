"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

A plain numpy restatement of the reference's `Truss.Solve()` path
(`/root/reference/slientruss3d/truss.py:329-364` and the helpers it calls), one truss
at a time, with the same algorithm and the same order of floating-point operations:
Python member loop -> dense global K -> boolean-mask elimination -> dense
`np.linalg.solve` (LAPACK dgesv, as the reference) -> reactions by K[~mask,:] @ u ->
per-member force loop.

Who may use it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` - as the checker / reported baseline, never as the thing shipped.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks this file against
 (a) the reference's own known answers `data/*_output_*.json` and
     `generate/cube-7_case_*.json` (copied as data fixtures under `tests/golden/`), and
 (b) dense, un-sparsified outputs captured by importing the real reference in the
     build container (`tests/golden/make_golden.py`, vectors in `tests/golden/*.npz`).

Input: the reference's JSON dict (`detail/combine_with_JSON.md:71-163`):
  {"joint": [[[x,y(,z)], "PIN"|"NO"|"ROLLER_X|Y|Z"], ...],
   "force": [[jointID, [fx,fy(,fz)]], ...],
   "member": [[[j0,j1],[a,e,density]], ...]}
"""
import numpy as np

ZERO_EPS = 1e-10  # reference utils.py:79-84

# reference type.py:48-74: True = constrained axis
_RESIST_3D = {"PIN": (True, True, True), "ROLLER_X": (True, False, False),
              "ROLLER_Y": (False, True, False), "ROLLER_Z": (False, False, True),
              "NO": (False, False, False)}
_RESIST_2D = {"PIN": (True, True), "ROLLER_X": (True, False),
              "ROLLER_Y": (False, True), "NO": (False, False)}


class OracleNotStable(Exception):
    """Counting test failed (reference raises TrussNotStableError, truss.py:332-333)."""


def truss_dim(data):
    return len(data["joint"][0][0])


def applied_loads(data):
    """Load dict as the reference keeps it: zero vectors dropped on insert
    (truss.py:177-182), later entries for the same joint overwrite earlier ones."""
    loads = {}
    for joint_id, vec in data["force"]:
        if not np.all(np.abs(np.array(vec, dtype=float)) < ZERO_EPS):
            loads[joint_id] = tuple(float(v) for v in vec)
    return loads


def resistance_mask(support, dim):
    """type.py:48-74."""
    table = _RESIST_3D if dim == 3 else _RESIST_2D
    return np.array(table[support])


def resistance_number(support, dim):
    """type.py:37-46."""
    if support == "PIN":
        return dim
    return 0 if support == "NO" else 1


def is_stable(data):
    """truss.py:158-164 (necessary-only counting test)."""
    dim = truss_dim(data)
    n_res = sum(resistance_number(s, dim) for _, s in data["joint"])
    enough = len(data["member"]) + n_res >= len(data["joint"]) * dim
    return enough if dim == 2 else (n_res >= 6 and enough)


def member_length(p0, p1):
    """truss.py:19,98: pure-Python float ops in this order."""
    return sum((p1[i] - p0[i]) ** 2. for i in range(len(p0))) ** 0.5


def member_cosines(p0, p1, length):
    """truss.py:60-63."""
    return [(p1[i] - p0[i]) / length for i in range(len(p0))]


def member_matK(p0, p1, a, e, length=None):
    """truss.py:65-86: k * [[cc^T, -cc^T], [-cc^T, cc^T]] from the direction cosines, one nested list ->
    one np.array -> one scalar multiplication, as the reference builds it (the products l*m etc. are formed
    once and negated by the unary minus, so the entries carry the same bits).  `length` is the member's cached
    length (the reference computes it once per member, truss.py:19,98, not once per matK)."""
    if length is None:
        length = member_length(p0, p1)
    k = e * a / length                                   # truss.py:56-58
    cs = member_cosines(p0, p1, length)
    if len(cs) == 3:
        x, y, z = cs
        xx, yy, zz, xy, xz, yz = x ** 2., y ** 2., z ** 2., x * y, x * z, y * z
        r0, r1, r2 = [xx, xy, xz], [xy, yy, yz], [xz, yz, zz]
    else:
        x, y = cs
        xx, yy, xy = x ** 2., y ** 2., x * y
        r0, r1, r2 = [xx, xy], [xy, yy], None
    rows = [r for r in (r0, r1, r2) if r is not None]
    top = [r + [-v for v in r] for r in rows]
    bot = [[-v for v in r] + r for r in rows]
    return k * np.array(top + bot)


def joint_positions(data):
    dim = truss_dim(data)
    return [tuple(float(vec[i]) for i in range(dim)) for vec, _ in data["joint"]]


class Prepared:
    """A truss as the reference holds it between `LoadFromJSON` and `Solve()` (truss.py:110-121, 401-421):
    positions as float tuples, members with their CACHED length (Member.__init__, truss.py:19), the load dict
    and the supports.  `solve(prepare(data))` then costs what the reference's `Truss.Solve()` costs - the
    reference's published protocol (example.py:1-25) loads the truss once and times only `Solve()`."""
    __slots__ = ("data", "dim", "pos", "members", "loads", "supports", "lengths")

    def __init__(self, data):
        self.data = data
        self.dim = truss_dim(data)
        self.pos = joint_positions(data)
        self.members = [(int(j0), int(j1), float(a), float(e), float(rho)) for (j0, j1), (a, e, rho) in data["member"]]
        self.lengths = [member_length(self.pos[j0], self.pos[j1]) for j0, j1, _a, _e, _r in self.members]
        self.loads = applied_loads(data)
        self.supports = [s for _, s in data["joint"]]


def prepare(data):
    return data if isinstance(data, Prepared) else Prepared(data)


def global_K(data):
    """truss.py:307-316: dense zero-initialised K, four dim x dim block += per member,
    members in ID order."""
    p = prepare(data)
    dim, pos = p.dim, p.pos
    K = np.zeros([len(pos) * dim, len(pos) * dim])
    for (j0, j1, a, e, _rho), length in zip(p.members, p.lengths):
        Ke = member_matK(pos[j0], pos[j1], a, e, length)
        for i, x in ((0, j0 * dim), (dim, j1 * dim)):
            for j, y in ((0, j0 * dim), (dim, j1 * dim)):
                K[x: x + dim, y: y + dim] += Ke[i: i + dim, j: j + dim]
    return K


def force_vector(data):
    """truss.py:303-304."""
    p = prepare(data)
    f = np.zeros([len(p.pos), p.dim])
    for joint_id, vec in p.loads.items():
        f[joint_id] = vec
    return f.ravel()


def free_mask(data):
    """truss.py:319-326: True where the displacement is unknown."""
    p = prepare(data)
    dim = p.dim
    mask = np.ones([len(p.pos) * dim], dtype=np.bool_)
    for joint_id, support in enumerate(p.supports):
        mask[joint_id * dim: (joint_id + 1) * dim] = np.logical_not(resistance_mask(support, dim))
    return mask


def truss_weight(data):
    """truss.py:52-54,166-168: sum of a * L * density."""
    p = prepare(data)
    return sum(a * length * rho for (_j0, _j1, a, _e, rho), length in zip(p.members, p.lengths))


def solve(data, check_stable=True):
    """truss.py:329-364 with dense (un-sparsified) outputs.  `data`: the JSON dict, or `prepare(data)` (the
    truss as the reference holds it once loaded - what a timing of `Solve()` alone must start from).

    Returns dict: u [nJ,dim], f_ext [nJ,dim] (applied load at free DOFs, K@u at
    constrained DOFs), N [nM] (axial force, tension positive), weight, K_ff, mask.
    Raises OracleNotStable (counting test) or numpy.linalg.LinAlgError (singular K_ff).
    """
    p = prepare(data)
    if check_stable and not is_stable(p.data):
        raise OracleNotStable("The truss is not stable !")
    dim, pos = p.dim, p.pos
    K = global_K(p)
    f = force_vector(p)
    mask = free_mask(p)

    u = np.zeros([len(pos) * dim])
    K_ff = K[mask, :][:, mask]
    u[mask] = np.linalg.solve(K_ff, f[mask])             # truss.py:343

    not_mask = np.logical_not(mask)
    f[not_mask] = (K[not_mask, :] @ u.reshape(-1, 1)).ravel()   # truss.py:348-349

    N = np.zeros([len(p.members)])
    for m, ((j0, j1, a, e, _rho), length) in enumerate(zip(p.members, p.lengths)):
        idx = list(range(j0 * dim, (j0 + 1) * dim)) + list(range(j1 * dim, (j1 + 1) * dim))
        Ke = member_matK(pos[j0], pos[j1], a, e, length)
        v = (Ke[dim:] @ u[idx].reshape(-1, 1)).ravel()   # force on joint1, truss.py:357
        axis = np.array(pos[j1]) - np.array(pos[j0])     # truss.py:89-91
        sign = 1. if np.dot(axis, v) > 0 else -1.
        N[m] = sign * (v ** 2).sum() ** 0.5              # utils.py:87-88
    return {"u": u.reshape(-1, dim), "f_ext": f.reshape(-1, dim), "N": N,
            "weight": truss_weight(p), "K_ff": K_ff, "mask": mask}


def sparsify(result):
    """The reference's result dicts (truss.py:344-345,350-351,358-359): drop joints whose
    every component is below 1e-10, and members with |N| < 1e-10."""
    u = {j: row for j, row in enumerate(result["u"]) if not np.all(np.abs(row) < ZERO_EPS)}
    f = {j: row for j, row in enumerate(result["f_ext"]) if not np.all(np.abs(row) < ZERO_EPS)}
    n = {m: float(v) for m, v in enumerate(result["N"]) if not abs(v) < ZERO_EPS}
    return u, f, n


def fitness_terms(data, result, allow_stress, allow_displace):
    """The GA fitness of ga.py:139-149 from a solved truss, using the sparse dict semantics
    of truss.py:429-462 with isGetSumViolation=True.
    Returns (fitness, isInternalAllowed, isDisplaceAllowed)."""
    u, _f, n = sparsify(result)
    areas = [float(a) for _, (a, _e, _rho) in data["member"]]
    stress_vio = sum(s - allow_stress for m, force in n.items()
                     if (s := abs(force) / areas[m]) > allow_stress)
    disp_vio = sum(l - allow_displace for d in u.values()
                   if (l := (d ** 2).sum() ** 0.5) > allow_displace)
    ok_s, ok_d = abs(stress_vio) < ZERO_EPS, abs(disp_vio) < ZERO_EPS
    fitness = result["weight"]
    if not ok_s:
        fitness += stress_vio / allow_stress * 1e5
    if not ok_d:
        fitness += disp_vio / allow_displace * 1e5
    return fitness, ok_s, ok_d


def densify(sparse_pairs, count, width=None):
    """JSON list [[id, value], ...] -> dense array with zeros for missing ids
    (the comparator of SURVEY.md section 4: missing key == 0)."""
    out = np.zeros([count] if width is None else [count, width])
    for key, value in sparse_pairs:
        out[key] = value
    return out
