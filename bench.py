#!/usr/bin/env python3
"""Benchmark of the batched Truss.Solve() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): the 942-bar truss (tests/golden/data/bar-942_input_0.json,
n_free = 696) packed once and replicated to `--batch` (default 4096) INDEPENDENT problems per GPU,
inputs resident in HBM before the timed region.  One step = one pass of the whole pipeline over
the batch: dofmap -> assemble -> potrf -> potrs -> recover (five kernel launches through the C ABI).
Weak scaling: every rank solves its own batch; no collective on the data path.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (trs_potrf_kernel, FP64 MFMA bound): algorithmic FLOP / measured
                average launch duration (events recorded on the launch stream inside the timed steps)
  cpu_baseline  the numpy oracle (faithful restatement of the reference's per-truss path) timed on
                the host cores of this box, bounded sample, 1 thread
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# FP64 dense peaks of MI355X: 78.6 TFLOP/s matrix (= vector) per the public datasheet quoted in
# SURVEY.md section 8d; /opt/skills/guides/MI355X_MICROARCH.md lists no FP64 row.  HBM 8 TB/s (guide).
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
STAGES = ("dofmap", "assemble", "potrf", "potrs", "recover")


def load_case(name):
    with open(os.path.join(ROOT, "tests", "golden", "data", name + ".json")) as fh:
        return json.load(fh)


def algorithmic_counts(n, nJ, nM):
    """Per-truss algorithmic work (DESIGN.md section 'Kernels')."""
    npad = (n + 63) // 64 * 64
    inputs = 8 * nM + 16 * nM + 24 * nJ + nJ + 24 * nJ
    # upper part by 16-row tiles incl. diagonal tiles, + rhs column chunk (16 wide)
    upper = sum((npad + 16 - (c // 16) * 16) for c in range(npad)) * 8
    return {
        "potrf_flops": n ** 3 / 3.0 + n ** 2,           # factor + fused forward substitution
        "assemble_bytes": inputs + upper,               # K written once (upper part) + inputs
        "assemble_bytes_full_contract": inputs + 8 * n * n + 8 * n,  # SURVEY section 8d figure
        "potrs_bytes": 8 * n * (n + 1) / 2 + 16 * n,    # U read once + y in, u out
        "recover_bytes": 8 * nM + 16 * nM + 24 * nJ + 8 * n + 24 * nJ + 24 * nJ + 8 * nM,
    }


def cpu_baseline(data, seconds=15.0):
    """Time the numpy oracle (kind 'port') single-threaded on this box; bounded sample."""
    from oracle import truss_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    orc.solve(data)  # warm
    t0, runs = time.perf_counter(), 0
    while True:
        orc.solve(data)
        runs += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or runs >= 1000:
            break
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return {"value": runs / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{runs} sequential oracle.solve() calls on bar-942 ({dt:.1f} s), BLAS threads=1; "
                      f"host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="trusses per GPU")
    ap.add_argument("--case", default="bar-942_input_0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from oracle import truss_oracle as orc
    from python_stable_3d_truss_analysis_amd import batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    data = load_case(args.case)
    packed = batch.pack_json([data]).replicate(args.batch)
    dev = batch.DeviceBatch(packed, device)
    n, nJ, nM = int(packed.n_free[0]), int(packed.nJ[0]), int(packed.nM[0])

    def step(events=None):
        calls = (dev.dofmap, dev.assemble, dev.potrf, dev.potrs, dev.recover)
        if events is None:
            dev.solve()
            return
        for call, ev in zip(calls, events):
            ev[0].record()
            call()
            ev[1].record()

    for _ in range(args.warmup):
        step()
    # events are recorded on torch's current stream, which is the stream the C ABI launches on
    all_events = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                   for _ in STAGES] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(all_events[k])
    barrier()
    elapsed = time.perf_counter() - t0

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    stage_ms = {s: float(np.mean([all_events[k][i][0].elapsed_time(all_events[k][i][1])
                                  for k in range(args.steps)])) for i, s in enumerate(STAGES)}
    res = dev.result()
    ref = orc.solve(data) if rank == 0 else None

    if rank == 0:
        total_trusses = world * args.batch * args.steps
        counts = algorithmic_counts(n, nJ, nM)
        potrf_s = stage_ms["potrf"] * 1e-3
        achieved_tflops = counts["potrf_flops"] * args.batch / potrf_s / 1e12
        asm_gbs = counts["assemble_bytes"] * args.batch / (stage_ms["assemble"] * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "potrf_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        err_u = float(np.abs(res.displace[0, :nJ] - ref["u"]).max() / np.abs(ref["u"]).max())
        err_n = float(np.abs(res.internal[0, :nM] - ref["N"]).max() / np.abs(ref["N"]).max())
        line = {
            "metric": "truss solves/sec (batched Solve)",
            "value": total_trusses / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic: bundled bar-942 truss replicated as independent problems",
            "config": {"workload": f"{args.case} x {args.batch} independent copies per GPU "
                                   f"(nJ {nJ}, nM {nM}, n_free {n})",
                       "batch_per_gpu": args.batch, "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": {"kernel": "trs_potrf_kernel", "bound": "mfma", "achieved": achieved_tflops,
                         "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "flop_per_truss": counts["potrf_flops"], "avg_launch_ms": stage_ms["potrf"]},
            "stages_ms": stage_ms,
            "assemble_roofline": {"bound": "hbm", "achieved": asm_gbs, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": asm_gbs / PEAK_HBM_GBS,
                                  "bytes_per_truss": counts["assemble_bytes"],
                                  "note": "upper part by 16-row tiles + rhs column; the full symmetric "
                                          "figure of SURVEY 8d would be "
                                          f"{counts['assemble_bytes_full_contract']} B"},
            "max_rel_err_vs_oracle": {"u": err_u, "N": err_n, "info_nonzero": int((res.info != 0).sum())},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(data, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
