"""MI355X-native batched direct-stiffness solver behind the `Truss.Solve()` API of
slientruss3d (reference: leo27945875/Python_Stable_3D_Truss_Analysis).

`Truss`, `Member`, `MemberType`, `SupportType` mirror the reference's Python model;
`solve_batch` / `PackedBatch` are the batched entry points underneath `Truss.Solve()`.
The arithmetic runs in hand-written HIP kernels for gfx950 reached through a C ABI
(`include/trs_solver.h`); there is no CPU fallback.
"""
import os as _os

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; streams beyond that SHARE
# a queue, and work on one then waits behind the other's.  This package uses four lanes for a ragged batch
# (`batch.RaggedSolver`) plus a DMA stream in `data.dataset_stream` and three streams in the host-fed pipeline: with four
# queues the dataset stream's device -> host copy lands on a lane's queue for some lane counts (measured: 351-471 K
# samples/s delivered instead of 512-521 K, EXPERIMENTS R5.5).  Eight queues unless the caller has chosen; read by the
# runtime when the process first touches the GPU, so this must come before that (importing torch does not).
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized():
        # too late for this process: the runtime read the variable when the GPU was first touched.  Say so - the
        # multi-stream paths then run on four hardware queues, 10-30 % slower ("Hardware queues")
        import warnings as _warnings
        _warnings.warn("python_stable_3d_truss_analysis_amd imported after the GPU was initialised: GPU_MAX_HW_QUEUES=8 "
                       "can no longer take effect; export it in the launcher (see INTEGRATION.md)", RuntimeWarning,
                       stacklevel=2)
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"

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
