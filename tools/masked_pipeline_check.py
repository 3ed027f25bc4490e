#!/usr/bin/env python3
"""Host-fed pipeline timeline with the copy kernels on their own CUs (TRS_PCIE_CUS), called from the default
stream and from a side stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch, generate as gen
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
packed = gen.generate_cube_batch(rng.integers(8, 191, size=B), gridRange=(6, 6, 6), seed=7)
pinned, pool = packed.pinned(), batch.ResultPool(tracked=True)
host_in = {f: torch.from_numpy(getattr(pinned, f)) for f in batch.RaggedSolver.GATHER}
host_out = batch.host_result_arrays(torch, pool, B, pinned.nJ_max, pinned.nM_max, torch.device("cuda:0"))
solver = batch.RaggedSolver(pinned, "cuda:0", reorder=True, max_slab_bytes=48 << 30, host_io=(host_in, host_out))
side = torch.cuda.Stream("cuda:0")
for name, ctx in (("default stream", torch.cuda.stream(torch.cuda.default_stream("cuda:0"))), ("side stream", torch.cuda.stream(side))):
    with ctx:
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            solver.step(); tq = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: step {dt * 1e3:.1f} ms = {B / dt / 1e6:.2f} M solves/s (the host had queued it after {tq * 1e3:.1f} ms)")
        rec = []
        solver.step(record=rec); torch.cuda.synchronize()
        if "-v" in sys.argv:
            for nm, e0, e1 in rec:
                bk = solver.buckets[int(nm.split()[1])]
                print(f"{e0.elapsed_time(e1):8.2f} ms  {nm} [{bk['count']} x {bk['dev'].rows}]")

# which neighbour slows the pull: the same pipeline without the device work (pull + push only)
for bk in solver.buckets:
    bk["dev"].solve = lambda: None
    bk["order_on_device"] = False
with torch.cuda.stream(side):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        solver.step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pull + push only (no order, no solve): step {dt * 1e3:.1f} ms")
    rec = []
    solver.step(record=rec); torch.cuda.synchronize()
    t = {nm: e0.elapsed_time(e1) for nm, e0, e1 in rec}
    k = len(solver.buckets) - 1
    print(f"  last bucket: pull {t[f'bucket {k} pull begins']:.1f} -> {t[f'bucket {k} pulled']:.1f} ms, push {t[f'bucket {k} push begins']:.1f} -> {t[f'bucket {k} pushed']:.1f} ms")

# the copy kernels by themselves: every bucket's pull (then push) back to back on one stream, then both at once
from python_stable_3d_truss_analysis_amd import _capi
lib = solver.lib
s_up, s_run, s_down = tuple(solver._streams)
def launch(which, blocks, stream):
    for bk, tabs in zip(solver.buckets, solver._tables):
        _capi.check(lib.trs_copy_rows(*(tabs[0] if which == 0 else tabs[1][0]), bk["count"], bk["rows"].data_ptr(), which, blocks, stream.cuda_stream), "copy")
nJ64, nM64 = packed.nJ.astype(np.int64), packed.nM.astype(np.int64)
pulled_bytes = int((nJ64 * 49 + nM64 * 24).sum())      # live bytes: xyz, loads, cbits per joint; conn, E, A per member
pushed_bytes = int((nJ64 * 48 + nM64 * 8).sum()) + 4 * B
print(f"pull {pulled_bytes / 1e9:.2f} GB, push {pushed_bytes / 1e9:.2f} GB per step")
for blocks in (16, 32, 64, 128, 256):
    out = []
    for mode in ("pull", "push", "both"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode in ("pull", "both"): launch(0, blocks, s_up)
        if mode in ("push", "both"): launch(1, blocks, s_down)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    print(f"{blocks:4d} work-groups: pulls alone {out[0]:.1f} ms ({pulled_bytes / out[0] / 1e6:.1f} GB/s), pushes alone {out[1]:.1f} ms "
          f"({pushed_bytes / out[1] / 1e6:.1f} GB/s), both {out[2]:.1f} ms")

# what the copy traffic costs the solver kernels: a RESIDENT solver of the same batch stepped on the run stream,
# alone and beside pulls / pushes that run on the copy streams all the while
res = batch.RaggedSolver(packed, "cuda:0", reorder=True)
blocks = int(os.environ.get("TRS_PCIE_BLOCKS", "64"))
def timed_step(stream, traffic):
    torch.cuda.synchronize()
    for which in traffic:
        for _ in range(2):
            launch(which, blocks, s_up if which == 0 else s_down)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(); res.step(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for label, stream, traffic in (("every CU, no traffic", side, ()), ("run-stream CUs, no traffic", s_run, ()),
                               ("run-stream CUs beside pulls", s_run, (0,)), ("run-stream CUs beside pushes", s_run, (1,)),
                               ("run-stream CUs beside both", s_run, (0, 1))):
    ts = [timed_step(stream, traffic) for _ in range(3)]
    print(f"resident step on {label}: {min(ts):.1f} ms")
# fewer work-groups for the pushes: stores to host memory are fire-and-forget, what a wave has in flight is bounded
# only by its own next wait - the fewer waves, the shorter the queue the solver's own stores stand in
def launch_push(nblocks):
    for _ in range(2):
        launch(1, nblocks, s_down)
for nblocks in (2, 4, 8, 16, 32, 64):
    torch.cuda.synchronize(); t0 = time.perf_counter(); launch(1, nblocks, s_down); torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        launch_push(nblocks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_run):
            e0.record(); res.step(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"pushes with {nblocks:3d} work-groups: alone {alone:.1f} ms ({pushed_bytes / alone / 1e6:.1f} GB/s); resident step beside them {min(ts):.1f} ms")
