"""`Member` and `Truss`: the Python model whose `Truss.Solve()` is the drop-in boundary.

Public names, argument meaning, result shapes (sparse dicts with the 1e-10 threshold)
and exceptions follow the reference's `slientruss3d/truss.py:10-466`.  What differs is
what happens inside `Solve()`: the reference assembles and solves one truss in Python
and numpy (`truss.py:329-364`); here `Solve()` is a batch-of-one call into the batched
HIP solver (`batch.solve_batch`), which needs a GPU and the in-tree C-ABI library and
raises `HipExtensionError` otherwise - there is no CPU fallback.
"""
import copy
import json
import math
from pprint import pformat

import numpy as np

from .type import MemberType, SupportType
from .utils import (CheckDim, DimensionError, GetLength, InvaildJointError, IsZero,
                    IsZeroVector, NotAllBeSetError, TrussNotSolvedError,
                    TrussNotStableError, ZERO_EPS)


def _distance(p, q):
    """sqrt of the sum of the squared differences - the squares as `x ** 2.0`, summed left to right, exactly as
    the generic form `sqrt(sum((b - a) ** 2.0 ...))` rounds (unrolled: this runs once per member of every truss
    object that is built)."""
    if len(p) == 3:
        return math.sqrt((q[0] - p[0]) ** 2.0 + (q[1] - p[1]) ** 2.0 + (q[2] - p[2]) ** 2.0)
    if len(p) == 2:
        return math.sqrt((q[0] - p[0]) ** 2.0 + (q[1] - p[1]) ** 2.0)
    return math.sqrt(sum((b - a) ** 2.0 for a, b in zip(p, q)))


class Member:
    """One bar between two joint positions (reference `truss.py:10-106`)."""

    def __init__(self, joint0, joint1, dim=3, memberType=None):
        self._dim = CheckDim(dim)
        if len(joint0) != dim or len(joint1) != dim:
            raise DimensionError(
                f"Dimension of each joint must be {dim}, but got dim(joint0) = {len(joint0)} "
                f"and dim(joint1) = {len(joint1)}.")
        self._ends = [joint0, joint1]
        # The instance is kept, not copied: two members given the same MemberType object
        # stay aliased through `memberType = ...` exactly as in the reference
        # (`truss.py:16-18,44-46`).
        self._type = MemberType() if memberType is None else memberType
        self._length = _distance(joint0, joint1)

    def __repr__(self):
        return f"Member[{self._ends[0]}, {self._ends[1]}, k={self.k :.4f}]"

    dim = property(lambda self: self._dim)
    e = property(lambda self: self._type.e)
    a = property(lambda self: self._type.a)
    density = property(lambda self: self._type.density)
    length = property(lambda self: self._length)

    @property
    def memberType(self):
        return self._type.Copy()

    @memberType.setter
    def memberType(self, other):
        self._type.Set(other)

    @property
    def weight(self):
        return self.a * self._length * self.density

    @property
    def k(self):
        """Axial stiffness E*A/L (`truss.py:56-58`)."""
        return self.e * self.a / self._length

    @property
    def cosines(self):
        p, q = self._ends
        return [(q[i] - p[i]) / self._length for i in range(self._dim)]

    @property
    def matK(self):
        """2*dim x 2*dim local stiffness k*[[cc^T, -cc^T], [-cc^T, cc^T]] (`truss.py:65-86`)."""
        c = np.asarray(self.cosines, dtype=float)
        block = np.outer(c, c)
        return self.k * np.block([[block, -block], [-block, block]])

    def IsTension(self, forceVec):
        """True when the force on joint1 points away from joint0 (`truss.py:89-91`)."""
        axis = np.asarray(self._ends[1], dtype=float) - np.asarray(self._ends[0], dtype=float)
        return bool(np.dot(axis, forceVec) > 0)

    def SetPosition(self, jointID_0or1, position):
        if jointID_0or1 not in (0, 1):
            raise KeyError("[jointID_0or1] must be 0 or 1.")
        self._ends[jointID_0or1] = position
        self._length = _distance(self._ends[0], self._ends[1])

    def Serialize(self):
        return {"joint0": list(self._ends[0]), "joint1": list(self._ends[1]),
                "memberType": self._type.Serialize()}

    def Copy(self):
        return Member(tuple(self._ends[0]), tuple(self._ends[1]), self._dim, self._type.Copy())


class Truss:
    """A 2D/3D pin-jointed truss (reference `truss.py:109-466`).

    Joint and member IDs are consecutive integers in insertion order (`truss.py:175,185`);
    the DOF of (joint j, axis a) is j*dim + a (`truss.py:312-314,324`).
    """

    def __init__(self, dim):
        self._dim = CheckDim(dim)
        self._pos = []        # jointID -> tuple of dim floats
        self._sup = []        # jointID -> SupportType value
        self._loads = {}      # jointID -> tuple of dim floats, insertion ordered
        self._ends = []       # memberID -> (jointID0, jointID1)
        self._bars = []       # memberID -> Member
        self._clear_results()

    def _clear_results(self):
        self._displace = None
        self._external = None
        self._internal = None
        self._solved = False

    def __repr__(self):
        bar = "-" * 30

        def section(title, body):
            return f"{bar}\n{title}\n{bar}\n{body}\n\n"

        solved = self._solved
        return (object.__repr__(self) + "\n"
                + section("Joints :", pformat(self.GetJoints()))
                + section("Forces :", pformat(self._loads))
                + section("Members :", pformat(self.GetMembers(False)))
                + section("Displaces:", pformat(self._displace) if solved else "(Not Solved)")
                + section("Internals:", pformat(self._internal) if solved else "(Not Solved)")
                + section("Externals:", pformat(self._external) if solved else "(Not Solved)"))

    # ------------------------------------------------------------------ counts
    dim = property(lambda self: self._dim)
    nJoint = property(lambda self: len(self._pos))
    nMember = property(lambda self: len(self._bars))
    nForce = property(lambda self: len(self._loads))
    isSolved = property(lambda self: self._solved)

    @property
    def nSupport(self):
        return sum(1 for s in self._sup if s != SupportType.NO)

    @property
    def nResistance(self):
        return sum(SupportType.GetResistanceNumber(s, self._dim) for s in self._sup)

    @property
    def isStable(self):
        """Necessary-only counting test of the reference (`truss.py:158-164`)."""
        nRes = self.nResistance
        enough = self.nMember + nRes >= self.nJoint * self._dim
        return enough if self._dim == 2 else (nRes >= 6 and enough)

    @property
    def weight(self):
        return sum(bar.weight for bar in self._bars)

    # ---------------------------------------------------------------- builders
    def AddNewJoint(self, vector, supportType=SupportType.NO):
        self._pos.append(tuple(float(vector[i]) for i in range(self._dim)))
        self._sup.append(supportType)

    def AddExternalForce(self, jointID, vector):
        if not (isinstance(jointID, (int, np.integer)) and 0 <= jointID < len(self._pos)):
            raise InvaildJointError(f"No such joint [{jointID}], can't add force on it.")
        if not IsZeroVector(vector):  # zero loads are dropped (`truss.py:181-182`)
            self._loads[int(jointID)] = tuple(float(vector[i]) for i in range(self._dim))

    def AddNewMember(self, jointID0, jointID1, memberType):
        self._ends.append((jointID0, jointID1))
        self._bars.append(Member(self._pos[jointID0], self._pos[jointID1], self._dim, memberType))

    # ----------------------------------------------------------------- setters
    def SetJointPosition(self, jointID, position):
        self._pos[jointID] = position
        for (j0, j1), bar in zip(self._ends, self._bars):
            if j0 == jointID:
                bar.SetPosition(0, position)
            if j1 == jointID:
                bar.SetPosition(1, position)

    def SetJointPositions(self, jointPositionDict):
        for jointID, position in jointPositionDict.items():
            self.SetJointPosition(jointID, position)

    def SetSupportType(self, jointID, supportType):
        # The reference assigns into a tuple here and raises TypeError (`truss.py:198-203`);
        # this implementation performs the documented intent.  See INTEGRATION.md.
        self._sup[jointID] = supportType

    def SetSupportTypes(self, supportTypeDict):
        for jointID, supportType in supportTypeDict.items():
            self.SetSupportType(jointID, supportType)

    def SetMemberType(self, memberID, memberType):
        self._bars[memberID].memberType = memberType

    def SetMemberTypes(self, memberTypeDict, isCheckAllSet=False):
        if isCheckAllSet and set(range(len(self._bars))) - set(memberTypeDict):
            raise NotAllBeSetError("Didn't set member types to all members.")
        for memberID, memberType in memberTypeDict.items():
            self._bars[memberID].memberType = memberType

    def SetMemberConnect(self, memberID, connect):
        bar = self._bars[memberID]
        bar.SetPosition(0, self._pos[connect[0]])
        bar.SetPosition(1, self._pos[connect[1]])
        self._ends[memberID] = (connect[0], connect[1])

    def SetMemberConnects(self, memberConnectDict):
        for memberID, connect in memberConnectDict.items():
            self.SetMemberConnect(memberID, connect)

    # ----------------------------------------------------------------- getters
    def GetJointPosition(self, jointID):
        return self._pos[jointID]

    def GetJointPositions(self):
        return dict(enumerate(self._pos))

    def GetSupportType(self, jointID):
        return self._sup[jointID]

    def GetSupportTypes(self):
        return dict(enumerate(self._sup))

    def GetMemberType(self, memberID):
        return self._bars[memberID].memberType

    def GetMemberTypes(self):
        return {i: bar.memberType for i, bar in enumerate(self._bars)}

    def GetMemberConnect(self, memberID):
        return self._ends[memberID]

    def GetMemberFromConnect(self, connect):
        for ends, bar in zip(self._ends, self._bars):
            if ends[0] == connect[0] and ends[1] == connect[1]:
                return bar
        return None

    def GetForce(self, jointID):
        return self._loads[jointID]

    def GetJoints(self, isProtect=True):
        return {i: (p, s) for i, (p, s) in enumerate(zip(self._pos, self._sup))}

    def GetMembers(self, isProtect=True):
        bars = [bar.Copy() for bar in self._bars] if isProtect else self._bars
        return {i: (j0, j1, bar) for i, ((j0, j1), bar) in enumerate(zip(self._ends, bars))}

    def GetForces(self, isProtect=True):
        return dict(self._loads) if isProtect else self._loads

    def GetDisplacements(self, isProtect=True):
        return copy.deepcopy(self._displace) if isProtect else self._displace

    def GetExternalForces(self, isProtect=True):
        return copy.deepcopy(self._external) if isProtect else self._external

    def GetInternalForces(self, isProtect=True):
        return copy.deepcopy(self._internal) if isProtect else self._internal

    def GetInternalStresses(self):
        if self._internal is None:
            return None
        return {m: force / self._bars[m].a for m, force in self._internal.items()}

    def GetResistances(self):
        """External force minus applied load at every supported joint (`truss.py:279-291`)."""
        if not self._solved:
            return None
        out = {}
        for jointID, sup in enumerate(self._sup):
            if sup == SupportType.NO:
                continue
            total = self._external.get(jointID, np.zeros([self._dim]))
            out[jointID] = total - self._loads[jointID] if jointID in self._loads else total
        return out

    def GetJointIDs(self):
        return list(range(len(self._pos)))

    def GetMemberIDs(self):
        return list(range(len(self._bars)))

    def GetUsedMemberTypes(self):
        return {bar.memberType for bar in self._bars}

    # ------------------------------------------- dense views used by the packer
    def GetExternalForceVector(self):
        """Dense load vector of length nJoint*dim (`truss.py:303-304`)."""
        f = np.zeros([len(self._pos), self._dim])
        for jointID, vec in self._loads.items():
            f[jointID] = vec
        return f.ravel()

    def PackedArrays(self):
        """This truss as the arrays the batched solver packs (`batch.pack_trusses`): positions [nJ, dim], member
        end joints [nM, 2] int32, sections [nM, 3] = (a, e, density), support types (list), loads [nJ, dim] -
        straight from the model's own lists, no per-joint getter calls."""
        nJ, dim = len(self._pos), self._dim
        xyz = np.array(self._pos, dtype=float).reshape(nJ, dim)
        conn = np.array(self._ends, dtype=np.int32).reshape(len(self._bars), 2)
        sections = np.array([(t.a, t.e, t.density) for t in (bar._type for bar in self._bars)],
                            dtype=float).reshape(len(self._bars), 3)
        return xyz, conn, sections, self._sup, self.GetExternalForceVector().reshape(nJ, dim)

    def GetKMatrix(self):
        """The dense global stiffness matrix, `[nJoint * dim, nJoint * dim]` (`truss.py:307-316`): every
        member's four dim x dim blocks `+- k c c^T` added at its joints' DOFs (DOF = joint * dim + axis),
        supports NOT eliminated.  Assembled on the GPU by the solve's own assembly kernel
        (`batch.global_stiffness`); no CPU fallback."""
        from .batch import global_stiffness  # late import: keeps the model importable without torch
        return global_stiffness([self])[0]

    def GetDisplacementUnknownMask(self):
        """True where the DOF is free (`truss.py:319-326`)."""
        mask = np.ones([len(self._pos) * self._dim], dtype=np.bool_)
        for jointID, sup in enumerate(self._sup):
            lo = jointID * self._dim
            mask[lo: lo + self._dim] = ~SupportType.GetResistanceMask(sup, self._dim)
        return mask

    # -------------------------------------------------------------------- solve
    def Solve(self):
        """Direct-stiffness analysis of this truss on the GPU (reference `truss.py:329-364`).

        Raises `TrussNotStableError` before any arithmetic when the counting test fails
        and `numpy.linalg.LinAlgError` when the reduced stiffness matrix is not positive
        definite (the reference's LU raises it for an exactly singular matrix).
        """
        if not self.isStable:
            raise TrussNotStableError("The truss is not stable !")
        from .batch import solve_batch  # late import: keeps the model importable without torch
        result = solve_batch([self])
        if int(result.info[0]) != 0:
            raise np.linalg.LinAlgError("Singular matrix")
        self.AdoptDenseResults(result.displace[0], result.external[0], result.internal[0])

    def AdoptDenseResults(self, displace, external, internal):
        """Install dense results (`[nJoint, dim]`, `[nJoint, dim]`, `[nMember]`) as the
        sparse result dicts of the reference: entries below 1e-10 in every component are
        dropped (`truss.py:344-345,350-351,358-359`).  Used by `Solve()` and by the
        batched callers, which solve many trusses in one launch."""
        nJ, nM, dim = len(self._pos), len(self._bars), self._dim
        u = np.asarray(displace, dtype=float)[:nJ, :dim]
        f = np.asarray(external, dtype=float)[:nJ, :dim]
        n = np.asarray(internal, dtype=float)[:nM]
        keep_u = (np.abs(u) >= ZERO_EPS).any(axis=1)
        keep_f = (np.abs(f) >= ZERO_EPS).any(axis=1)
        ju, jf, jm = np.flatnonzero(keep_u), np.flatnonzero(keep_f), np.flatnonzero(np.abs(n) >= ZERO_EPS)
        # (one copy per array, its rows handed out as the dict values: they do not alias the caller's arrays)
        self._displace = dict(zip(ju.tolist(), u[ju]))
        self._external = dict(zip(jf.tolist(), f[jf]))
        self._internal = dict(zip(jm.tolist(), n[jm].tolist()))
        self._solved = True

    # --------------------------------------------------------------------- JSON
    def Serialize(self):
        """The JSON-ready dict of `detail/combine_with_JSON.md:71-163` (`truss.py:367-398`)."""
        data = {
            "joint": [[list(p), SupportType.GetFromType(s)] for p, s in zip(self._pos, self._sup)],
            "force": [[j, list(v)] for j, v in self._loads.items()],
            "member": [[[j0, j1], bar._type.Serialize()]
                       for (j0, j1), bar in zip(self._ends, self._bars)],
        }
        if self._solved:
            data["displace"] = [[j, list(v)] for j, v in self._displace.items()]
            data["external"] = [[j, list(v)] for j, v in self._external.items()]
            data["internal"] = [[m, float(v)] for m, v in self._internal.items()]
            data["weight"] = self.weight
        return data

    def LoadFromJSON(self, path=None, isOutputFile=False, data=None):
        if data is None:
            with open(path, "r", encoding="utf-8") as f:
                data = json.load(f)
        for vector, supportName in data["joint"]:
            self.AddNewJoint(vector, SupportType.GetFromString(supportName))
        for jointID, vector in data["force"]:
            self.AddExternalForce(jointID, vector)
        for (jointID0, jointID1), typeList in data["member"]:
            self.AddNewMember(jointID0, jointID1, MemberType(*typeList))
        if isOutputFile:
            self._displace = {j: np.array(v) for j, v in data["displace"]}
            self._external = {j: np.array(v) for j, v in data["external"]}
            self._internal = {m: float(v) for m, v in data["internal"]}
            self._solved = True
        return self

    def DumpIntoJSON(self, path):
        with open(path, "w", encoding="utf-8") as f:
            json.dump(self.Serialize(), f, ensure_ascii=False)

    def Copy(self):
        return Truss(self._dim).LoadFromJSON(data=self.Serialize(), isOutputFile=self._solved)

    # ------------------------------------------------------- constraint checks
    def _excess(self, values, limit, isGetSumViolation, isGetSumNonViolation):
        """Shared body of the two `Is...Allowed` checks (`truss.py:429-462`).
        `values` is an iterable of (key, magnitude)."""
        values = list(values)
        if isGetSumViolation:
            violation = sum(v - limit for _, v in values if v > limit)
            ok = bool(IsZero(violation))
        else:
            violation = {key: v - limit for key, v in values if v > limit}
            ok = len(violation) == 0
        if isGetSumNonViolation:
            return ok, violation, sum(limit - v for _, v in values if v <= limit)
        return ok, violation

    def IsInternalStressAllowed(self, limit, isGetSumViolation=False, isGetSumNonViolation=False):
        if not self._solved:
            raise TrussNotSolvedError("Haven't done structural analysis yet.")
        stresses = ((m, abs(force) / self._bars[m].a) for m, force in self._internal.items())
        return self._excess(stresses, limit, isGetSumViolation, isGetSumNonViolation)

    def IsDisplacementAllowed(self, limit, isGetSumViolation=False, isGetSumNonViolation=False):
        if not self._solved:
            raise TrussNotSolvedError("Haven't done structural analysis yet.")
        lengths = ((j, GetLength(d)) for j, d in self._displace.items())
        return self._excess(lengths, limit, isGetSumViolation, isGetSumNonViolation)
