"""MI355X-native batched direct-stiffness solver behind the `Truss.Solve()` API of
slientruss3d (reference: leo27945875/Python_Stable_3D_Truss_Analysis).

`Truss`, `Member`, `MemberType`, `SupportType` mirror the reference's Python model;
`solve_batch` / `PackedBatch` are the batched entry points underneath `Truss.Solve()`.
The arithmetic runs in hand-written HIP kernels for gfx950 reached through a C ABI
(`include/trs_solver.h`); there is no CPU fallback.
"""
from .type import GenerateMethod, LinkType, MemberType, MetapathType, SupportType, TaskType
from .truss import Member, Truss
from .utils import HipExtensionError, TrussNotStableError

__all__ = ["Truss", "Member", "MemberType", "SupportType", "MetapathType", "TaskType",
           "LinkType", "GenerateMethod", "HipExtensionError", "TrussNotStableError",
           "solve_batch", "pack_trusses", "PackedBatch", "BatchResult", "RaggedSolver", "DeviceBatch",
           "ShardedSolver", "solve_batch_sharded", "solve_batch_distributed"]


def __getattr__(name):
    # torch-dependent names are resolved lazily so that the model imports without torch
    if name in ("solve_batch", "pack_trusses", "PackedBatch", "BatchResult", "RaggedSolver", "DeviceBatch"):
        from . import batch
        return getattr(batch, name)
    if name in ("ShardedSolver", "solve_batch_sharded", "solve_batch_distributed"):
        from . import shard
        return getattr(shard, name)
    raise AttributeError(name)
