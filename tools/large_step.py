#!/usr/bin/env python3
"""Steps of the large-truss route (bench.py's `large_truss` leg) for profiling:
    rocprofv3 --kernel-trace --stats -- python3 tools/large_step.py [trusses]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from python_stable_3d_truss_analysis_amd import batch

ap = argparse.ArgumentParser()
ap.add_argument("trusses", type=int, nargs="?", default=512)
args = ap.parse_args()
args.large_trusses = args.trusses
out = bench.large_truss_leg(args, torch.device("cuda:0"), torch, batch)
print(json.dumps(out, indent=1))
