#!/usr/bin/env python3
"""BASELINE config 5 as an SPMD job: every rank (one per GPU; `torchrun --nproc-per-node N tools/dataset_spmd.py`,
or plainly `python tools/dataset_spmd.py` for one GPU) generates its share of a dataset of random cube trusses,
solves each sample twice (real sections + the fixed-section prior) and forms the graph features, all on its
GPU (data.dataset_chunks).  No communication; every rank reports its own rate."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from python_stable_3d_truss_analysis_amd import MemberType, data
from python_stable_3d_truss_analysis_amd.type import TaskType

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=131072)
ap.add_argument("--chunk", type=int, default=16384)
ap.add_argument("--no-prefetch", action="store_true")
ap.add_argument("--order", default="profile", choices=("profile", "fast", "rcm"))
ap.add_argument("--generate", default="device", choices=("device", "host"), help="where the chunks are generated")
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()}"
t0 = time.perf_counter()
count, bad, bytes_down = 0, 0, 0
for first, packed, tensors in data.dataset_chunks(args.samples, rank, world, args.chunk, seed=1, device=device,
                                                  fixedMemberType=MemberType(1., 1e7, 0.1),
                                                  taskType=TaskType.REGRESSION, forceScale=1e3, displaceScale=0.1,
                                                  positionScale=100., reorder=args.order, prefetch=not args.no_prefetch,
                                                  generate=args.generate):
    host = {k: v.cpu() for k, v in tensors.items() if hasattr(v, "cpu")}   # what a data loader would store
    bad += int(host["info"].sum())
    bytes_down += sum(v.numel() * v.element_size() for v in host.values())
    count += packed.B
dt = time.perf_counter() - t0
print(json.dumps({"rank": rank, "world": world, "samples": count, "seconds": dt, "samples_per_s": count / dt,
                  "order": args.order, "prefetch": not args.no_prefetch, "info_nonzero": bad, "downloaded_MB": bytes_down / 1e6}))
