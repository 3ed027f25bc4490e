"""Exceptions, tolerances and small numeric helpers of the Truss.Solve() boundary.

Mirrors the observable behaviour of the reference's `slientruss3d/utils.py:52-88`
(exception class names, `CheckDim`, the 1e-10 sparsification threshold used by
`IsZero` / `IsZeroVector`, and `GetLength`).  The plotting helpers of that file
(`utils.py:12-48`) are out of scope.
"""
import math

import numpy as np

INF = float("inf")

#: absolute threshold below which a result component is dropped from the sparse
#: result dicts (reference `utils.py:79-84`).
ZERO_EPS = 1e-10


# Same class names as the reference (including the spelling "Invaild"), `utils.py:52-67`.
class InvalidSupportTypeError(Exception):
    pass


class InvalidMetapathTypeError(Exception):
    pass


class InvalidTaskTypeError(Exception):
    pass


class InvalidLinkTypeError(Exception):
    pass


class InvalidGenerateMethodError(Exception):
    pass


class TrussNotStableError(Exception):
    pass


class TrussNotSolvedError(Exception):
    pass


class DimensionError(Exception):
    pass


class InvaildJointError(Exception):
    pass


class EliteNumberTooMuchError(Exception):
    pass


class ProbabilityGreaterThanOneError(Exception):
    pass


class OnlyOneMemberTypeError(Exception):
    pass


class MinStressTooLargeError(Exception):
    pass


class MinDisplaceTooLargeError(Exception):
    pass


class NotAllBeSetError(Exception):
    pass


class PinNotEnoughError(Exception):
    pass


class HipExtensionError(RuntimeError):
    """The HIP solver library or a GPU is missing: the product path has no CPU fallback."""


def CheckDim(dim):
    """Accept only 2 or 3 (reference `utils.py:71-75`)."""
    if dim != 2 and dim != 3:
        raise DimensionError(f"Dimension of truss and member must be 2 or 3, but got [{dim}].")
    return dim


def IsZero(num, eps=ZERO_EPS):
    """|num| < eps; works element-wise on arrays (reference `utils.py:79-80`)."""
    return abs(num) < eps


def IsZeroVector(vec, eps=ZERO_EPS):
    """True when every component is below eps in magnitude (reference `utils.py:83-84`)."""
    return bool(np.all(np.abs(np.asarray(vec, dtype=float)) < eps))


def GetLength(vec):
    """Euclidean norm (reference `utils.py:87-88`)."""
    arr = np.asarray(vec, dtype=float)
    return math.sqrt(float(np.dot(arr, arr)))


def MinNorm(vec, minNorm=1.0):
    """Scale `vec` up so that its norm is at least `minNorm` (reference `utils.py:91-92`)."""
    arr = np.asarray(vec, dtype=float)
    return arr * max(1.0, minNorm / np.linalg.norm(arr))


def GetCenter(position0, position1):
    """Mid-point of a member (reference `utils.py:101-102`)."""
    return [(p + q) * 0.5 for p, q in zip(position0, position1)]


def GetAngles(position0, position1):
    """Four direction features of a member, lower end first (reference `utils.py:105-113`)."""
    lo, hi = (position0, position1) if position0[-1] < position1[-1] else (position1, position0)
    d = [h - l for l, h in zip(lo, hi)]
    full = math.sqrt(sum(c * c for c in d))
    plan = math.sqrt(sum(c * c for c in d[:2]))
    if IsZero(plan):
        return plan / full, d[2] / full, 0.0, 0.0
    return plan / full, d[2] / full, d[1] / plan, d[0] / plan


def GetPowerset(s):
    """Every subset of the sequence `s`, in binary counting order (subset i holds s[j] for the set bits j of i):
    the order in which the reference enumerates a cube's vertices (`utils.py:95-98`, `generate.py:168-174`)."""
    for i in range(1 << len(s)):
        yield [item for j, item in enumerate(s) if (i >> j) & 1]


def InfinteLoop():
    """0, 1, 2, ... without end (`utils.py:117-121`; the name is the reference's)."""
    i = 0
    while True:
        yield i
        i += 1

