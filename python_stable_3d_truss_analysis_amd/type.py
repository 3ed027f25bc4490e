"""Value types that feed the solver: `MemberType` and `SupportType`.

Behavioural mirror of the reference's `slientruss3d/type.py:5-88`.  The encoding the
HIP path consumes is `SupportType.ConstraintBits` (new): one bit per constrained
axis, which is what the reference's `GetResistanceMask` (`type.py:48-74`) expresses
as a boolean vector.
"""
import numpy as np

from .utils import CheckDim, IsZero, InvalidSupportTypeError


class MemberType:
    """(a, e, density): cross-section area, Young's modulus, density (`type.py:5-27`)."""

    __slots__ = ("a", "e", "density")

    def __init__(self, a=1.0, e=1.0, density=1.0):
        self.a = float(a)
        self.e = float(e)
        self.density = float(density)

    def __repr__(self):
        return f"MemberType(a={self.a}, e={self.e}, density={self.density})"

    def __eq__(self, other):
        # tolerance equality, as the reference (`type.py:14-15`)
        return (IsZero(self.a - other.a) and IsZero(self.e - other.e)
                and IsZero(self.density - other.density))

    def __hash__(self):
        return hash((self.a, self.e, self.density))

    def Set(self, other):
        """In-place overwrite; members sharing this instance all see it (`type.py:20-21`)."""
        self.a, self.e, self.density = other.a, other.e, other.density

    def Serialize(self):
        return [self.a, self.e, self.density]

    def Copy(self):
        return MemberType(self.a, self.e, self.density)


class SupportType:
    """Support enumeration 0..4 (`type.py:30-35`)."""
    NO = 0
    PIN = 1
    ROLLER_X = 2
    ROLLER_Y = 3
    ROLLER_Z = 4

    _NAMES = {0: "NO", 1: "PIN", 2: "ROLLER_X", 3: "ROLLER_Y", 4: "ROLLER_Z"}
    # axis bits: x=1, y=2, z=4.  A roller constrains exactly its own axis (`type.py:52-57`).
    _BITS3 = {0: 0, 1: 7, 2: 1, 3: 2, 4: 4}
    _BITS2 = {0: 0, 1: 3, 2: 1, 3: 2}

    @staticmethod
    def GetResistanceNumber(supportType, dim):
        """Number of reaction components of one joint (`type.py:37-46`)."""
        if supportType == SupportType.PIN:
            return dim
        if supportType in (SupportType.ROLLER_X, SupportType.ROLLER_Y, SupportType.ROLLER_Z):
            return 1
        if supportType == SupportType.NO:
            return 0
        raise InvalidSupportTypeError(f"[GetResistanceNumber] No such support type [{supportType}] !")

    @staticmethod
    def ConstraintBits(supportType, dim):
        """Bit mask of constrained axes of one joint (x=1, y=2, z=4): the HIP path's encoding.

        Unknown types raise for both dimensions (the reference's 3D branch builds the
        exception without raising it, `type.py:62`, and then fails later on `None`).
        """
        table = SupportType._BITS3 if CheckDim(dim) == 3 else SupportType._BITS2
        try:
            return table[supportType]
        except (KeyError, TypeError):
            raise InvalidSupportTypeError(
                f"[GetResistanceMask] No such {dim}D-support type [{supportType}] !") from None

    @staticmethod
    def GetResistanceMask(supportType, dim):
        """Boolean vector, True where the axis is constrained (`type.py:48-74`)."""
        bits = SupportType.ConstraintBits(supportType, dim)
        return np.array([bool(bits >> axis & 1) for axis in range(dim)])

    @staticmethod
    def GetFromString(string):
        """'PIN' -> SupportType.PIN, etc. (`type.py:76-81`; a table here instead of eval)."""
        for value, name in SupportType._NAMES.items():
            if name == string:
                return value
        raise InvalidSupportTypeError(f"[GetFromString] No such support type [{string}] !")

    @staticmethod
    def GetFromType(supportType):
        """SupportType.PIN -> 'PIN' (`type.py:83-88`)."""
        return SupportType._NAMES.get(supportType)


# Plain integer enumerations used by the callers of Solve() (`type.py:91-111`).
class MetapathType:
    USE_IMPLICIT = 0
    NO_IMPLICIT = 1


class TaskType:
    OPTIMIZATION = 0
    REGRESSION = 1


class LinkType:
    LeftBottom_RightTop = 0
    RightBottom_LeftTop = 1
    Cross = 2
    Random = 3


class GenerateMethod:
    DFS = 0
    BFS = 1
    Random = 2
