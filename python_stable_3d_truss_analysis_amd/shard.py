"""Batch sharding over the GPUs of one node.

Every truss is a closed problem, so the batch splits with NO data-path collective: rank r of W
solves its own shard on its own GPU (one process per GPU); results stay on the rank or are
gathered on the host.  `torch.distributed` is used only by the callers for barriers/timing and by
`gather_results` (an object gather of small result arrays; gloo or nccl).
"""
import numpy as np

from .batch import BatchResult, PackedBatch


def shard_indices(costs, world_size):
    """Size-balanced partition: sort by cost (n_free^3) descending and deal round-robin.

    Returns a list of index arrays, one per rank; the union is a permutation of range(B)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    return [np.sort(order[r::world_size]) for r in range(world_size)]


def shard_batch(packed: PackedBatch, rank, world_size):
    """The shard of `packed` that `rank` solves, and its indices in the full batch."""
    idx = shard_indices(packed.n_free.astype(np.float64) ** 3, world_size)[rank]
    return packed.take(idx), idx


def gather_results(local: BatchResult, idx, total, group=None):
    """Reassemble the full-batch dense results on every rank (host-side gather of small arrays)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (idx, local.displace, local.external, local.internal, local.info),
                           group=group)
    nJ = max(p[1].shape[1] for p in parts)
    nM = max(p[3].shape[1] for p in parts)
    out = BatchResult(np.zeros([total, nJ, 3]), np.zeros([total, nJ, 3]), np.zeros([total, nM]),
                      np.zeros([total], dtype=np.int32))
    for pidx, u, f, n, info in parts:
        out.displace[pidx, :u.shape[1]] = u
        out.external[pidx, :f.shape[1]] = f
        out.internal[pidx, :n.shape[1]] = n
        out.info[pidx] = info
    return out
