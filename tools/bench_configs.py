#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (informational; bench.py is the contract).

  config 3  GenerateRandomCubeTrusses-like mixed sizes (native generator), bucketed ragged batching
  config 4  one GA generation on bar-120: population 1024 in one batched solve + fitness reduction
  config 5  dataset sample = two solves (actual + fixed section) of cube trusses, graph tensors on host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from python_stable_3d_truss_analysis_amd import MemberType, Truss, batch, _capi
if os.environ.get("TRS_LIB_VARIANT"):  # A/B builds of tools/build_variants.sh
    _capi.LIB_PATH = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd", "variants",
                                  f"libtrs_{os.environ['TRS_LIB_VARIANT']}.so")
from python_stable_3d_truss_analysis_amd import generate as gen
from python_stable_3d_truss_analysis_amd.ga import GA


def config3(B, out, reorder, key, slab_gb=48, granularity=64):
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    packed = gen.generate_cube_batch(rng.integers(8, 191, size=B), gridRange=(6, 6, 6), seed=7)
    t_gen = time.perf_counter() - t0
    t_rcm = 0.0
    if reorder:
        t0 = time.perf_counter()
        packed = batch.permute_joints(packed, batch.joint_order(packed, reorder))
        t_rcm = time.perf_counter() - t0
    groups = batch.size_buckets(packed, int(slab_gb) << 30, granularity)
    subs = [packed.take(i).trimmed() for i in groups]
    torch.cuda.synchronize()
    total_gpu = 0.0
    bad = 0
    for sub in subs:
        dev = batch.DeviceBatch(sub)
        dev.solve(); torch.cuda.synchronize()            # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dev.solve(); e1.record(); torch.cuda.synchronize()
        total_gpu += e0.elapsed_time(e1) * 1e-3
        bad += int((dev.info != 0).sum().item())
        del dev
    out[key] = {"B": B, "buckets": len(groups), "n_free_mean": float(packed.n_free.mean()),
                      "n_free_max": int(packed.n_free.max()), "generate_s": t_gen, "reorder_s": t_rcm,
                      "solve_s_resident": total_gpu, "solves_per_s": B / total_gpu, "info_nonzero": bad}


def config4(out):
    import random
    with open(os.path.join(ROOT, "tests", "golden", "data", "bar-120_input_0.json")) as fh:
        truss = Truss(3).LoadFromJSON(data=json.load(fh))
    random.seed(0)
    types = [MemberType(i, random.uniform(1e7, 3e7), random.uniform(0.1, 1.0)) for i in range(1, 21)]
    ga = GA(truss, types, nIteration=3, nPop=1024, nElite=256)
    pop = ga.Initialize()
    ga.GetFitnessBatch(pop)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ga.GetFitnessBatch(pop)
    dt = (time.perf_counter() - t0) / 5
    out["config4"] = {"nPop": 1024, "generation_eval_s": dt, "fitness_evals_per_s": 1024 / dt}


def config5(B, out, order=True):
    """Dataset sample = generate -> two batched solves (actual + fixed section) -> graph tensors.
    Host form: results downloaded, features formed on the host (graphfeat.c).  Device form: the two solves
    share one upload / reordering, the features are formed on the GPU (graphfeat.hip) and only float32
    feature tensors come down."""
    from python_stable_3d_truss_analysis_amd import data as gdata
    from python_stable_3d_truss_analysis_amd.type import TaskType
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    packed = gen.generate_cube_batch(rng.integers(8, 191, size=B), gridRange=(6, 6, 6), seed=11)
    t1 = time.perf_counter()
    fixed = MemberType(1., 1e7, 0.1)
    gdata.solve_actual_and_prior(packed.take(np.arange(min(B, 256))), fixed, device="cuda:0", reorder=order)   # warm
    t2 = time.perf_counter()
    actual, prior = gdata.solve_actual_and_prior(packed, fixed, device="cuda:0", reorder=order)
    t3 = time.perf_counter()
    graphs = gdata.hetero_tensors_batch(packed, actual, prior, fixed.a, TaskType.REGRESSION)
    t4 = time.perf_counter()
    out["config5_host_features"] = {
        "B": B, "generate_s": t1 - t0, "two_solves_end_to_end_s": t3 - t2, "graphs_s": t4 - t3,
        "samples_per_s": B / ((t1 - t0) + (t3 - t2) + (t4 - t3)), "graphs": len(graphs),
        "info_nonzero": int((actual.info != 0).sum() + (prior.info != 0).sum())}
    del graphs, actual, prior
    gdata.feature_tensors_device(packed.take(np.arange(min(B, 256))), fixed, TaskType.REGRESSION, reorder=order)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    tensors = gdata.feature_tensors_device(packed, fixed, TaskType.REGRESSION, reorder=order)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    host = {k: v.cpu() for k, v in tensors.items() if hasattr(v, "cpu") and k != "conn"}
    t7 = time.perf_counter()
    graphs = gdata.graphs_from_tensors(packed, {**host, "conn": tensors["conn"].cpu()})
    t8 = time.perf_counter()
    out["config5_device_features"] = {
        "B": B, "generate_s": t1 - t0, "solves_and_features_on_device_s": t6 - t5, "download_float32_s": t7 - t6,
        "graph_objects_s": t8 - t7,
        "samples_per_s_tensors": B / ((t1 - t0) + (t6 - t5) + (t7 - t6)),
        "samples_per_s_with_graph_objects": B / ((t1 - t0) + (t6 - t5) + (t8 - t6)),
        "info_nonzero": int(host["info"].sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cubes", type=int, default=65536)
    ap.add_argument("--only", choices=("ga", "generator", "rcm", "profile", "dataset"), help="run one configuration only")
    ap.add_argument("--samples", type=int, default=16384, help="dataset samples of config 5")
    ap.add_argument("--slab-gb", type=int, default=48, help="stiffness-slab memory per launch pipeline")
    ap.add_argument("--order", default="profile", choices=("profile", "fast", "rcm"), help="joint order of the dataset path")
    ap.add_argument("--bucket", type=int, default=64, help="size-bucket granularity (multiple of 64)")
    args = ap.parse_args()
    out = {}
    if args.only in (None, "ga"):
        config4(out)
    if args.only in (None, "generator"):
        config3(args.cubes, out, False, "config3_generator_order", args.slab_gb, args.bucket)
    if args.only in (None, "dataset"):
        config5(args.samples, out, args.order)
    if args.only in (None, "rcm"):
        config3(args.cubes, out, "rcm", "config3_rcm_order", args.slab_gb, args.bucket)
    if args.only in (None, "profile"):
        config3(args.cubes, out, "profile", "config3_profile_order", args.slab_gb, args.bucket)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
