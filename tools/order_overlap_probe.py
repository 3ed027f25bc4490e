#!/usr/bin/env python3
"""Can the joint order of the NEXT bucket hide behind the factorisation of the current one?  A full resident step
on one stream with, beside it on a second stream, the gathers + joint orders of a second solver's buckets (extra
work): if the pair takes about as long as the step alone, a two-stream pipeline would save the order's time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch, _capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sizes, tensors = bench.cube_workload(B, 0, device="cuda:0")
main = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
other = batch.RaggedSolver(sizes, reorder=True, tensors=tensors)
for s in (main, other):
    s.step()
torch.cuda.synchronize()
main.adopt_launch_hints()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def orders_only(solver, stream):
    with torch.cuda.stream(stream):
        for bk, (gather, scatter) in zip(solver.buckets, solver._tables):
            _capi.check(solver.lib.trs_copy_rows(*gather, bk["count"], bk["rows"].data_ptr(), 0, 0, stream.cuda_stream), "gather")
            if bk["order_on_device"]:
                batch.joint_order_device(torch, bk["raw"], effort=solver.plan[1], out=bk["ordered"])

def run(step, orders):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if orders:
        orders_only(other, sb)
    if step:
        with torch.cuda.stream(sa):
            main.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

for label, args in (("step alone", (True, False)), ("gathers + orders alone", (False, True)), ("both at once", (True, True))):
    ts = [run(*args) for _ in range(4)]
    print(f"{label}: {min(ts):.1f} ms")
# the same with the two streams on disjoint sets of CUs (orders on `n` CUs, the step on the rest)
for n in (32, 48, 64):
    ms = batch._MaskedStreams(torch, torch.device("cuda:0"), main.lib, n, 8)
    sb, sa, _ = tuple(ms)
    out = []
    for args in ((True, False), (False, True), (True, True)):
        out.append(min(run(*args) for _ in range(4)))
    print(f"orders on {n} CUs, step on {256 - n - 8}: step alone {out[0]:.1f} ms, orders alone {out[1]:.1f} ms, both {out[2]:.1f} ms")
    ms.close()
