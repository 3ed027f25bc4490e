#!/usr/bin/env python3
"""The ragged cube step with its buckets dealt round-robin onto TWO (or more) streams, each with a workspace of its own:
does the next bucket's work fill the tail of the current bucket's kernels?  (`STREAMS=2 python tools/two_stream_step.py`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

ns = int(os.environ.get("STREAMS", 2))
sizes, tensors = bench.cube_workload(int(os.environ.get("CUBES", 65536)), 0, device="cuda:0")
solver = batch.RaggedSolver(sizes, reorder=True, tensors=tensors, max_slab_bytes=int(os.environ.get("MAXSLAB_GB", 48)) << 30)
solver.step(); torch.cuda.synchronize()
solver.adopt_launch_hints()
dev = solver.device
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
# a workspace per extra stream; the buckets go to the streams by longest-processing-time-first on their slab sizes
spaces = [(solver._S, solver._uf, solver._work, solver._env)] + \
         [tuple(torch.empty_like(t) for t in (solver._S, solver._uf, solver._work, solver._env)) for _ in range(ns - 1)]
cost = [bk["count"] * bk["dev"].rows * bk["dev"].ld for bk in solver.buckets]
load, lane = [0] * ns, {}
for k in sorted(range(len(cost)), key=lambda k: -cost[k]):
    l = min(range(ns), key=lambda l: load[l])
    lane[k] = l
    load[l] += cost[k]
for k, bk in enumerate(solver.buckets):
    db, Bb = bk["dev"], bk["count"]
    if db.small:
        continue
    S, uf, work, env = spaces[lane[k]]
    wb, ei = solver.lib.trs_assemble_work_bytes(db.nJ_max, db.nM_max, db.n_max), solver.lib.trs_env_ints(db.n_max)
    db._slab = (S[:Bb * db.rows * db.ld].view(Bb, db.rows, db.ld), uf[:Bb * db.rows].view(Bb, db.rows),
                work[:Bb * wb].view(Bb, wb), env[:Bb * ei].view(Bb, ei))

def step_streams(order):
    nJ_full, nM_full = int(solver.u.shape[1]), int(solver.N.shape[1])
    inp = solver.inputs
    start = torch.cuda.Event(); start.record()
    for s in streams:
        s.wait_event(start)
    for pos, k in enumerate(order):
        bk = solver.buckets[k]
        s = streams[lane[k]]
        with torch.cuda.stream(s):
            if bk["fused_io"]:
                db, ordr = bk["dev"], bk["ordered"]
                _capi.check(solver.lib.trs_joint_order_rows(
                    bk["count"], db.nJ_max, db.nM_max, bk["rows"].data_ptr(), int(inp["xyz"].shape[1]),
                    int(inp["conn"].shape[1]), inp["xyz"].data_ptr(), inp["conn"].data_ptr(),
                    inp["cbits"].data_ptr(), inp["loads"].data_ptr(), inp["E"].data_ptr(), inp["A"].data_ptr(),
                    inp["nJ"].data_ptr(), inp["nM"].data_ptr(), ordr["perm"].data_ptr(), ordr["reach"].data_ptr(),
                    db.xyz.data_ptr(), db.conn.data_ptr(), db.cbits.data_ptr(), db.loads.data_ptr(),
                    db.E.data_ptr(), db.A.data_ptr(), db.nJ.data_ptr(), db.nM.data_ptr(),
                    int(solver.device_effort), s.cuda_stream), "trs_joint_order_rows")
                db.solve_rows(bk["rows"], solver.outs[0], nJ_full, nM_full)
            else:   # (the small-system bucket: gather, solve, scatter)
                gather, scatters = solver._tables[k]
                _capi.check(solver.lib.trs_copy_rows(*gather, bk["count"], bk["rows"].data_ptr(), 0, 0, s.cuda_stream), "gather")
                bk["dev"].solve()
                _capi.check(solver.lib.trs_copy_rows(*scatters[0], bk["count"], bk["rows"].data_ptr(), 1, 0, s.cuda_stream), "scatter")
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        e = torch.cuda.Event(); e.record(s); cur.wait_event(e)

def bench_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

ref = solver.u.clone()
nb = len(solver.buckets)
print(f"{len(solver.buckets)} buckets, slab cap {os.environ.get('MAXSLAB_GB', 48)} GB, workspace {sum(t.numel() * t.element_size() for t in spaces[0]) / 2**30:.1f} GB per stream")
print(f"one stream (RaggedSolver.step with workspaces re-pointed): {bench_ms(solver.step):.2f} ms")
orders = {"as sorted (largest first)": list(range(nb)),
          "interleaved large/small": [x for pair in zip(range(nb // 2), range(nb - 1, nb // 2 - 1, -1)) for x in pair] + ([nb // 2] if nb % 2 else [])}
for name, order in orders.items():
    order = list(dict.fromkeys(order))
    ms = bench_ms(lambda: step_streams(order))
    torch.cuda.synchronize()
    same = bool((solver.u == ref).all())
    print(f"{ns} streams, {name}: {ms:.2f} ms (results bitwise equal: {same})")
