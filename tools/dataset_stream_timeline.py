#!/usr/bin/env python3
"""Config 5 end to end (data.dataset_stream): samples/s delivered in host memory against the resident rate, by chunk
size and ring depth, with the wall time between consecutive chunk hand-overs.
    python tools/dataset_stream_timeline.py [--samples 131072]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from python_stable_3d_truss_analysis_amd import MemberType, TaskType
from python_stable_3d_truss_analysis_amd import data as gdata

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=131072)
args = ap.parse_args()
kw = dict(seed=11, numCubeRange=(8, 190), gridRange=(6, 6, 6), fixedMemberType=MemberType(1., 1e7, 0.1),
          taskType=TaskType.REGRESSION, device="cuda:0", forceScale=1e3, displaceScale=0.1, positionScale=100.)
for chunk, slots in ((32768, 2), (16384, 2)):
    for _ in gdata.dataset_stream(slots * chunk, chunk=chunk, slots=slots, **kw):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for first, meta, t in gdata.dataset_chunks(args.samples, chunk=chunk, **kw):
        pass
    torch.cuda.synchronize()
    resident = time.perf_counter() - t0
    t0 = time.perf_counter()
    stamps, nbytes, rec = [], 0, []
    for graphs in gdata.dataset_stream(args.samples, chunk=chunk, slots=slots, record=rec, **kw):
        stamps.append(time.perf_counter() - t0)
        nbytes += graphs.nbytes
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"chunk {chunk:6d} slots {slots}: resident {args.samples / resident / 1e3:7.1f} K/s, streamed {args.samples / dt / 1e3:7.1f} K/s "
          f"({nbytes / args.samples / 1e3:.1f} KB/sample, {nbytes / dt / 1e9:.1f} GB/s); hand-overs at "
          + " ".join(f"{s * 1e3:.0f}" for s in stamps) + " ms")
    base = rec[0][1]
    print("    " + "; ".join(f"{name} {base.elapsed_time(a):.0f}-{base.elapsed_time(b):.0f}" for name, a, b in rec[:8]))
    gdata.release_stream_buffers()
