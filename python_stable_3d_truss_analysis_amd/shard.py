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

#: the arrays a shard crosses to its worker with (the general member form; a batch in the table form is expanded first)
_FIELDS = tuple(f for f in PackedBatch.__dataclass_fields__ if f not in ("type_idx", "types"))


def shard_indices(costs, world_size):
    """Size-balanced partition: sort by cost (n_free^3) descending and deal round-robin.

    Returns a list of index arrays, one per rank; the union is a permutation of range(B)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    return [np.sort(order[r::world_size]) for r in range(world_size)]


def shard_batch(packed: PackedBatch, rank, world_size):
    """The shard of `packed` that `rank` solves, and its indices in the full batch."""
    idx = shard_indices(packed.n_free.astype(np.float64) ** 3, world_size)[rank]
    return packed.take(idx), idx


def _same_host(group):
    """True when every rank of the group runs on this host (then results travel through shared memory)."""
    import socket
    import torch.distributed as dist
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, (socket.gethostname(), os.stat("/dev/shm").st_dev if os.path.isdir("/dev/shm") else -1),
                           group=group)
    return all(n == names[0] and n[1] != -1 for n in names)


def _shm_files(shapes, headroom=64 << 20):
    """Rank 0 of `gather_results`: four files of the full-batch shapes in /dev/shm with their pages allocated, or None
    when /dev/shm has no room for them (a container's default tmpfs is 64 MB; 65 536 cube trusses are 2.2 GB) -
    whatever was created is removed again."""
    import tempfile
    sizes = [int(np.prod([max(1, d) for d in shape])) * np.dtype(dt).itemsize for shape, dt in shapes]
    paths = []
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize < sum(sizes) + headroom:
            return None
        for nbytes in sizes:
            fd, path = tempfile.mkstemp(prefix="trs_gather_", dir="/dev/shm")
            paths.append(path)
            try:
                # the pages are allocated (zeroed) HERE, in one call: ranks that fault fresh pages of one tmpfs
                # file in at the same time serialise on it (measured: 1.8 s instead of 0.3 s for 0.55 GB)
                os.posix_fallocate(fd, 0, nbytes)
            finally:
                os.close(fd)
        return paths
    except OSError:
        for path in paths:
            try:
                os.unlink(path)
            except OSError:
                pass
        return None


def _map_file(path, shape, dtype, private):
    """The file as an array: shared (this rank's writes reach the file) or private (copy-on-write: the file's pages
    are read as they are, a write touches a private copy of its page only)."""
    import mmap
    shape1 = tuple(max(1, d) for d in shape)
    fd = os.open(path, os.O_RDWR)
    try:
        mm = mmap.mmap(fd, int(np.prod(shape1)) * np.dtype(dtype).itemsize,
                       flags=mmap.MAP_PRIVATE if private else mmap.MAP_SHARED, prot=mmap.PROT_READ | mmap.PROT_WRITE)
    finally:
        os.close(fd)
    return np.frombuffer(mm, dtype=dtype).reshape(shape1)[tuple(slice(0, d) for d in shape)]


def gather_results(local: BatchResult, idx, total, group=None, widths=None):
    """Reassemble the full-batch dense results on every rank.  `widths` = (nJ_max, nM_max) of the full batch
    (every rank holds the same `packed`, so `solve_batch_distributed` passes them; else they are agreed on first).

    Ranks of ONE host (the case this package is built for: the GPUs of one node) with room in /dev/shm: rank 0
    creates four files of the full-batch shapes there (fresh pages are zero = the result padding), every rank maps
    them shared and writes ITS rows, and every rank returns arrays mapped PRIVATELY (copy-on-write) from the same
    files - no pickling, no copy of another rank's rows, an in-place edit on one rank stays on that rank, and nothing
    is left in /dev/shm (the names are unlinked once every rank has its mapping; the memory lives until the last
    array is dropped).  65 536 cube trusses (2.2 GB of results): one memcpy of each rank's share.
    The choice is made COLLECTIVELY: rank 0 checks the room and allocates, and broadcasts the file names or None -
    then every rank takes the path below.
    Across hosts, or without room in /dev/shm: one `all_gather` of a contiguous float64 buffer per rank (no Python
    objects)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if widths is None:
        w = [None] * world
        dist.all_gather_object(w, (local.displace.shape[1], local.internal.shape[1]), group=group)
        widths = (max(x[0] for x in w), max(x[1] for x in w))
    nJ, nM = int(widths[0]), int(widths[1])
    shapes = (([total, nJ, 3], np.float64), ([total, nJ, 3], np.float64), ([total, nM], np.float64), ([total], np.int32))
    parts = (local.displace, local.external, local.internal, local.info)
    if _same_host(group):
        box = [_shm_files(shapes) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        paths = box[0]
        if paths is not None:
            try:
                for path, (shape, dt), part in zip(paths, shapes, parts):
                    view = _map_file(path, shape, dt, private=False)
                    if part.ndim == 1:
                        view[idx] = part
                    else:
                        view[idx, :part.shape[1]] = part
                    del view
                dist.barrier(group)               # every rank's rows are in the files
                views = [_map_file(path, shape, dt, private=True) for path, (shape, dt) in zip(paths, shapes)]
                dist.barrier(group)               # every rank has its mapping:
            finally:
                if rank == 0:
                    for path in paths:            # the names can go (nothing stays in /dev/shm whatever happens next)
                        try:
                            os.unlink(path)
                        except OSError:
                            pass
            return BatchResult(views[0], views[1], views[2], views[3])
    # several hosts (or no room in /dev/shm): one contiguous buffer per rank (rows padded to the largest shard),
    # gathered as tensors
    counts = [None] * world
    dist.all_gather_object(counts, len(idx), group=group)
    cmax, width = max(counts + [1]), 6 * nJ + nM + 2
    buf = np.zeros([cmax, width])
    k = len(idx)
    wJ, wM = int(local.displace.shape[1]), int(local.internal.shape[1])
    buf[:k, 0] = idx
    buf[:k, 1] = local.info
    buf[:k, 2:2 + 3 * wJ] = local.displace.reshape(k, 3 * wJ)          # (k may be 0: an empty shard)
    buf[:k, 2 + 3 * nJ:2 + 3 * nJ + 3 * wJ] = local.external.reshape(k, 3 * wJ)
    buf[:k, 2 + 6 * nJ:2 + 6 * nJ + wM] = local.internal
    on_gpu = dist.get_backend(group) == "nccl"
    mine = torch.from_numpy(buf)
    if on_gpu:
        mine = mine.cuda()
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    out = BatchResult(np.zeros([total, nJ, 3]), np.zeros([total, nJ, 3]), np.zeros([total, nM]),
                      np.zeros([total], dtype=np.int32))
    for r, t in enumerate(got):
        a = t.cpu().numpy()[:counts[r]]
        rows = a[:, 0].astype(np.int64)
        out.info[rows] = a[:, 1].astype(np.int32)
        out.displace[rows] = a[:, 2:2 + 3 * nJ].reshape(-1, nJ, 3)
        out.external[rows] = a[:, 2 + 3 * nJ:2 + 6 * nJ].reshape(-1, nJ, 3)
        out.internal[rows] = a[:, 2 + 6 * nJ:2 + 6 * nJ + nM]
    return out


def _solve_shard(shard, device, reorder, max_slab_bytes=64 << 30, sections=None):
    """A shard through the HIP pipeline on `device` - the ONLY solver of this module (no CPU fallback: without
    a GPU or the library `batch.solve_batch` raises)."""
    from .batch import solve_batch
    return solve_batch(shard, device=device, reorder=reorder, max_slab_bytes=max_slab_bytes, sections=sections)


def solve_batch_distributed(packed: PackedBatch, device=None, gather=True, reorder=False, group=None):
    """SPMD entry point: call on every rank of an initialised `torch.distributed` group with the SAME
    `packed`.  Rank r solves its shard on `device` (default: `cuda:LOCAL_RANK % device_count`) with the
    HIP pipeline (`batch.solve_batch`) and returns the full-batch result (gather=True, identical on every
    rank) or `(local_result, idx)`.  The native host helpers of this rank (joint order, generator) size
    their thread teams by the rank's share of the host CPUs (`generate.set_host_thread_share`)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine, idx = shard_batch(packed, rank, world)
    if device is None:
        import torch
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}"
    from .generate import set_host_thread_share
    set_host_thread_share(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    local = _solve_shard(mine, device, reorder) if mine.B else BatchResult(
        np.zeros([0, packed.nJ_max, 3]), np.zeros([0, packed.nJ_max, 3]), np.zeros([0, packed.nM_max]),
        np.zeros([0], dtype=np.int32))
    if not gather:
        return local, idx
    return gather_results(local, idx, packed.B, group, widths=(packed.nJ_max, packed.nM_max))


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


def _worker_main(device, conn, n_workers=1):
    """Entry point of a worker process: the HIP pipeline on `device` (no fallback: a missing GPU or library
    raises here and the error text travels back to the controller), then the request loop."""
    try:
        import torch
        torch.cuda.set_device(torch.device(device))
        # every solve of the shard's geometry in one call: one upload, one reordering
        solver = lambda shard, opts: _solve_shard(
            shard, device, opts.get("reorder", False), opts.get("max_slab_bytes", 64 << 30),
            opts.get("sections") if opts.get("sections") is not None else [None])
    except Exception:  # pragma: no cover - reported to the parent
        conn.send(("error", traceback.format_exc()))
        return
    _worker_loop(device, conn, solver, n_workers)


def _worker_loop(device, conn, solver, n_workers=1):
    """Request loop of a worker: receives shard descriptors (shared-memory names), runs
    `solver(shard, opts) -> [BatchResult per section variant]` and writes the rows of its shard into the
    shared result arrays.  The worker's native host helpers (joint order, generator: OpenMP) get the
    worker's share of the host CPUs, not all of them (`generate.set_host_thread_share`)."""
    from .generate import host_threads, set_host_thread_share
    set_host_thread_share(n_workers)
    conn.send(("ready", None))
    resident = None  # (geometry key, shapes, DeviceBatch) kept between calls (GA: only the sections change)
    while True:
        msg = conn.recv()
        if msg[0] == "stop":
            return
        if msg[0] == "host_threads":
            conn.send(("done", host_threads()))
            continue
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
                # GA generation (ga.py:139-160): the shard's geometry stays resident on this GPU - only for a
                # caller who NAMES the geometry (geometry_key) and only while the padded shapes agree; without
                # a key every call uploads the shard it was given
                from .batch import DeviceBatch
                key = opts.get("geometry_key")
                shapes = (shard.B, shard.nJ_max, shard.nM_max)
                if key is not None and resident is not None and resident[:2] == (key, shapes):
                    dev = resident[2]
                    dev.set_sections(shard.A, shard.E, shard.rho)
                else:
                    resident = None
                    dev = DeviceBatch(shard, device)
                    dev.packed = None  # the host arrays are views of shared memory that goes away
                    if key is not None:
                        resident = (key, shapes, dev)
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

    #: entry point of the worker processes (module-level function, pickled by name for `spawn`)
    _worker_target = staticmethod(_worker_main)

    def __init__(self, devices=None):
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
            proc = ctx.Process(target=self._worker_target, args=(dev, child, len(self.devices)), daemon=True)
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

    def host_threads(self):
        """OpenMP team size of the native host helpers in every worker (together at most the CPUs the
        container has, or one each when there are fewer CPUs than workers)."""
        out = []
        for proc, conn in self._workers:
            conn.send(("host_threads",))
            out.append(int(conn.recv()[1]))
        return out

    def fitness(self, packed: PackedBatch, allow_stress, allow_displace, geometry_key=None):
        """GA population evaluation (`ga.py:139-160`) sharded over the workers: one batched solve plus
        the `trs_fitness` reductions per shard.  Returns (fit [B,3] = weight, stress violation,
        displacement violation; info [B]).  With a `geometry_key` (any hashable that names the geometry, e.g.
        `id(truss)`) equal to the previous call's, and the same padded shapes, the geometry stays resident on
        the workers' GPUs and only the member sections cross; without one every call uploads the batch."""
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
        packed = packed.general()
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
