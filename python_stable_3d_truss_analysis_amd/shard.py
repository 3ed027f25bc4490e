"""Batch sharding over the GPUs of one node.

Every truss is a closed problem, so the batch splits with NO data-path collective: each GPU gets a
size-balanced shard and runs the whole pipeline on it in its own process; results are gathered on the
host.  Two ways to drive it:

* SPMD (`torchrun` / `bench.py --gpus N`): every rank calls `solve_batch_distributed(packed)` - it
  solves `shard_batch(packed, rank, world)` on its own GPU and, if asked, reassembles the full result
  on every rank with `gather_results` (an object gather of small host arrays; gloo or nccl).
  `torch.distributed` carries nothing else.
* single controller (`ShardedSolver`, `solve_batch_sharded`): one Python process owns the batch and
  keeps one worker PROCESS per GPU alive; shards and results travel through shared host memory
  (no pickling of the arrays).  This is what the callers of the solve path use when more than one
  device is visible - the reference's loops `generate.py:342-374`, `data.py:107-114` and
  `ga.py:155-160` are the workloads it covers.

Workers are started with the `spawn` method before/independently of the parent's GPU state and never
exec another program.
"""
import os
import traceback

import numpy as np

from .batch import BatchResult, PackedBatch

_FIELDS = tuple(PackedBatch.__dataclass_fields__)


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


def solve_batch_distributed(packed: PackedBatch, device=None, gather=True, reorder=False, group=None,
                            solver=None):
    """SPMD entry point: call on every rank of an initialised `torch.distributed` group with the SAME
    `packed`.  Rank r solves its shard on `device` (default: `cuda:LOCAL_RANK % device_count`) with the
    HIP pipeline (`batch.solve_batch`; `solver` replaces it in CPU tests of the plumbing) and returns
    the full-batch result (gather=True, identical on every rank) or `(local_result, idx)`."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine, idx = shard_batch(packed, rank, world)
    if solver is None:
        import torch
        from .batch import solve_batch
        if device is None:
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}"
        solver = lambda shard: solve_batch(shard, device=device, reorder=reorder)
    local = solver(mine) if mine.B else BatchResult(
        np.zeros([0, packed.nJ_max, 3]), np.zeros([0, packed.nJ_max, 3]), np.zeros([0, packed.nM_max]),
        np.zeros([0], dtype=np.int32))
    if not gather:
        return local, idx
    return gather_results(local, idx, packed.B, group)


# ---- single-controller path: one persistent worker process per GPU ------------------------------------

def _shm_array(shape, dtype):
    """A numpy array in POSIX shared memory: (array, handle).  The handle must stay referenced."""
    from multiprocessing import shared_memory
    nbytes = max(1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
    shm = shared_memory.SharedMemory(create=True, size=nbytes)
    return np.ndarray(shape, dtype=dtype, buffer=shm.buf), shm


def _attach(name, shape, dtype):
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=name)
    return np.ndarray(shape, dtype=dtype, buffer=shm.buf), shm


def _import_callable(path):
    import importlib
    mod, name = path.split(":")
    return getattr(importlib.import_module(mod), name)


def _worker_main(device, conn, test_solver):
    """Worker loop: receives shard descriptors (shared-memory names), solves on `device`, writes the
    rows of its shard into the shared result arrays.  No fallback: a missing GPU/library raises in
    the worker and the error text travels back to the controller.  `test_solver` ("module:function",
    CPU tests of this plumbing only) replaces the HIP pipeline by a stand-in."""
    try:
        if test_solver is None:
            from .batch import DeviceBatch, solve_batch
            import torch
            torch.cuda.set_device(torch.device(device))
            # every solve of the shard's geometry in one call: one upload, one reordering
            solver = lambda shard, opts: solve_batch(
                shard, device=device, reorder=opts.get("reorder", False),
                max_slab_bytes=opts.get("max_slab_bytes", 64 << 30),
                sections=opts.get("sections") if opts.get("sections") is not None else [None])
        else:
            stand_in = _import_callable(test_solver)

            def solver(shard, opts):
                import dataclasses
                out = []
                for sec in (opts.get("sections") if opts.get("sections") is not None else [None]):
                    ones = np.ones_like(shard.A)
                    out.append(stand_in(shard if sec is None else dataclasses.replace(
                        shard, A=ones * sec[0], E=ones * sec[1], rho=ones * sec[2])))
                return out
        conn.send(("ready", None))
    except Exception:  # pragma: no cover - reported to the parent
        conn.send(("error", traceback.format_exc()))
        return
    resident = {}  # geometry key -> DeviceBatch kept between calls (GA: only the sections change)
    while True:
        msg = conn.recv()
        if msg[0] == "stop":
            return
        _, inputs, outputs, idx_desc, opts = msg
        handles = []
        try:
            arrays = {}
            for f, (name, shape, dtype) in inputs.items():
                arrays[f], h = _attach(name, shape, dtype)
                handles.append(h)
            idx, h = _attach(*idx_desc)
            handles.append(h)
            shard = PackedBatch(*(arrays[f] for f in _FIELDS))
            outs = {}
            for k, (name, shape, dtype) in outputs.items():
                outs[k], h = _attach(name, shape, dtype)
                handles.append(h)
            if opts.get("fitness") is not None:
                # GA generation (ga.py:139-160): the shard's geometry stays resident on this GPU
                key = opts.get("geometry_key")
                dev = resident.get(key)
                if dev is None or dev.B != shard.B:
                    resident.clear()
                    dev = resident[key] = DeviceBatch(shard, device)
                    dev.packed = None  # the host arrays are views of shared memory that goes away
                else:
                    dev.set_sections(shard.A, shard.E, shard.rho)
                w, sv, dv = dev.solve_fitness(*opts["fitness"])
                outs["fit"][idx, 0] = w.cpu().numpy()
                outs["fit"][idx, 1] = sv.cpu().numpy()
                outs["fit"][idx, 2] = dv.cpu().numpy()
                outs["info"][0, idx] = dev.info.cpu().numpy()
                del arrays, shard, outs, idx
                conn.send(("done", None))
                continue
            for slot, res in enumerate(solver(shard, opts)):
                nJ, nM = res.displace.shape[1], res.internal.shape[1]
                outs["u"][slot, idx, :nJ] = res.displace
                outs["f"][slot, idx, :nJ] = res.external
                outs["n"][slot, idx, :nM] = res.internal
                outs["info"][slot, idx] = res.info
            del arrays, shard, outs, idx
            conn.send(("done", None))
        except Exception:
            conn.send(("error", traceback.format_exc()))
        finally:
            for h in handles:
                try:
                    h.close()
                except BufferError:  # a view still alive: the mapping goes with the process
                    pass


class ShardedSolver:
    """One worker process per device, each running the HIP pipeline on its shard of every batch.

        with ShardedSolver(["cuda:0", "cuda:1"]) as pool:
            res = pool.solve(packed, reorder=True)

    `devices=None` takes every visible GPU.  The same device may be listed twice (two workers share
    it): that is how the multi-process path is tested on a 1-GPU box."""

    def __init__(self, devices=None, _test_solver=None):
        import multiprocessing as mp
        if devices is None:
            import torch
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        if not devices:
            from .utils import HipExtensionError
            raise HipExtensionError("no GPU visible: the truss solver has no CPU fallback")
        self.devices = list(devices)
        ctx = mp.get_context("spawn")
        self._workers = []
        for dev in self.devices:
            parent, child = ctx.Pipe()
            proc = ctx.Process(target=_worker_main, args=(dev, child, _test_solver), daemon=True)
            proc.start()
            child.close()
            self._workers.append((proc, parent))
        for proc, conn in self._workers:
            kind, text = conn.recv()
            if kind != "ready":
                self.close()
                from .utils import HipExtensionError
                raise HipExtensionError(f"shard worker failed to start:\n{text}")

    @property
    def world_size(self):
        return len(self.devices)

    def fitness(self, packed: PackedBatch, allow_stress, allow_displace, geometry_key=None):
        """GA population evaluation (`ga.py:139-160`) sharded over the workers: one batched solve plus
        the `trs_fitness` reductions per shard.  Returns (fit [B,3] = weight, stress violation,
        displacement violation; info [B]).  With the same `geometry_key` and batch size as the
        previous call only the member sections cross to the GPUs."""
        fit, info = self._run(packed, {"fitness": (float(allow_stress), float(allow_displace)),
                                       "geometry_key": geometry_key}, 1, want_fit=True)
        return fit, info[0]

    def solve(self, packed: PackedBatch, reorder=False, sections=None, max_slab_bytes=64 << 30):
        """Full-batch dense results, rows in the order of `packed`.

        `sections=None`: one solve with the batch's own member sections -> `BatchResult`.
        `sections=[None, (a, e, density), ...]`: several solves of the same geometry (None = own
        sections, a triple = every member set to it, reference `data.py:107-114`) -> list of
        `BatchResult`; the shard's inputs cross to the worker once."""
        nsolve = 1 if sections is None else len(sections)
        results = self._run(packed, {"reorder": reorder, "sections": sections,
                                     "max_slab_bytes": max_slab_bytes}, nsolve, want_fit=False)
        return results[0] if sections is None else results

    def _run(self, packed, opts, nsolve, want_fit):
        B, nJm, nMm = packed.B, packed.nJ_max, packed.nM_max
        keep = []
        out_desc, outs = {}, {}
        spec = [("info", [nsolve, B], np.int32)]
        if want_fit:
            spec.append(("fit", [B, 3], np.float64))
        else:
            spec += [("u", [nsolve, B, nJm, 3], np.float64), ("f", [nsolve, B, nJm, 3], np.float64),
                     ("n", [nsolve, B, nMm], np.float64)]
        for k, shape, dtype in spec:
            outs[k], h = _shm_array(shape, dtype)
            outs[k][...] = 0
            keep.append(h)
            out_desc[k] = (h.name, shape, np.dtype(dtype).str)
        parts = shard_indices(packed.n_free.astype(np.float64) ** 3, self.world_size)
        busy = []
        try:
            for (proc, conn), idx in zip(self._workers, parts):
                if len(idx) == 0:
                    continue
                in_desc = {}
                for f in _FIELDS:
                    src = getattr(packed, f)
                    dst, h = _shm_array((len(idx),) + src.shape[1:], src.dtype)
                    np.take(src, idx, axis=0, out=dst)
                    keep.append(h)
                    in_desc[f] = (h.name, dst.shape, src.dtype.str)
                ishm, h = _shm_array([len(idx)], np.int64)
                ishm[:] = idx
                keep.append(h)
                conn.send(("solve", in_desc, out_desc, (h.name, [len(idx)], "<i8"), opts))
                busy.append(conn)
            errors = []
            for conn in busy:
                try:
                    kind, text = conn.recv()
                except (EOFError, OSError):   # the worker process died (e.g. killed for memory)
                    kind, text = "error", "shard worker exited without an answer"
                if kind != "done":
                    errors.append(text)
            if errors:
                from .utils import HipExtensionError
                raise HipExtensionError("shard worker failed:\n" + "\n".join(errors))
            if want_fit:
                return outs["fit"].copy(), outs["info"].copy()
            return [BatchResult(outs["u"][s].copy(), outs["f"][s].copy(), outs["n"][s].copy(),
                                outs["info"][s].copy()) for s in range(nsolve)]
        finally:
            del outs
            for h in keep:
                try:
                    h.close()
                except BufferError:
                    pass
                try:
                    h.unlink()
                except FileNotFoundError:
                    pass

    def close(self):
        for proc, conn in self._workers:
            try:
                conn.send(("stop",))
            except (BrokenPipeError, OSError):
                pass
        for proc, conn in self._workers:
            proc.join(timeout=30)
            if proc.is_alive():  # pragma: no cover
                proc.kill()      # exactly the process this object started
            conn.close()
        self._workers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def solve_batch_sharded(packed: PackedBatch, devices=None, reorder=False, sections=None):
    """One-shot form of `ShardedSolver.solve` (starts and stops the workers: prefer the pool for a
    stream of batches)."""
    with ShardedSolver(devices) as pool:
        return pool.solve(packed, reorder=reorder, sections=sections)


def visible_devices():
    """Device names of every visible GPU without initialising the GPU runtime in this process."""
    try:
        import torch
        return [f"cuda:{i}" for i in range(torch.cuda.device_count())]
    except Exception:  # pragma: no cover
        return []
