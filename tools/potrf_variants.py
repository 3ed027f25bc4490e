#!/usr/bin/env python3
"""A/B timing of solver-library variants (tools/build_variants.sh) in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Times each pipeline stage with events on the launch stream and
checks every variant's solution against the first one.

    python tools/potrf_variants.py [--batch 4096] [--rounds 5] tagA tagB ...   ('default' = the product lib)
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from python_stable_3d_truss_analysis_amd import _capi, batch


def load_variant(tag):
    path = _capi.LIB_PATH if tag == "default" else os.path.join(
        ROOT, "python_stable_3d_truss_analysis_amd", "variants", f"libtrs_{tag}.so")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _capi.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="+")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--case", default="bar-942_input_0")
    ap.add_argument("--dense", action="store_true", help="no envelope tile skipping")
    ap.add_argument("--fused", action="append", default=[],
                    help="tags (repeatable) that run with the option compact=1: compact entry lists + fused factorisation")
    ap.add_argument("--option", action="append", default=[],
                    help="tag:name=value (repeatable): DeviceBatch.options[name] = value while that tag runs "
                         "(batch.DEFAULT_OPTIONS otherwise); the options travel as per-call flags")
    args = ap.parse_args()
    with open(os.path.join(ROOT, "tests", "golden", "data", args.case + ".json")) as fh:
        data = json.load(fh)
    dev = batch.DeviceBatch(batch.pack_json([data]).replicate(args.batch), use_envelope=not args.dense)
    libs = {t: load_variant(t.split("@")[0]) for t in args.tags}   # 'tag@x' = a second instance of a build
    stages = ("dofmap", "assemble", "potrf", "potrs", "recover")
    times = {t: {s: [] for s in stages} for t in args.tags}
    ref_u = None
    narrow_known = dev.all_narrow
    options = {t: {"compact": True} for t in args.fused}
    for spec in args.option:
        tag, kv = spec.split(":")
        name, value = kv.split("=")
        options.setdefault(tag, {})[name] = bool(int(value.partition("/")[0]))
    base_options = dict(dev.options)
    for rnd in range(args.rounds + 1):
        for tag in args.tags:
            dev.lib = libs[tag]
            dev.all_narrow = narrow_known and not tag.endswith("nohint")   # 'tag@nohint': every kernel is launched
            dev.options = dict(base_options, **options.get(tag, {}))
            evs = []
            for s in stages:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); getattr(dev, s)(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if rnd == 0:   # warm-up round + correctness
                u = dev.u[0].cpu().numpy()
                info = int(dev.info.abs().sum().item())
                if ref_u is None:
                    ref_u = u
                err = float(np.abs(u - ref_u).max() / np.abs(ref_u).max())
                print(f"{tag}: info_nonzero={info} max_rel_diff_vs_first={err:.2e}")
                continue
            for s, (e0, e1) in zip(stages, evs):
                times[tag][s].append(e0.elapsed_time(e1))
    for tag in args.tags:
        if hasattr(libs[tag], "trs_debug_stamps"):   # diagnostic build: per-phase wave-cycle shares of potrf
            buf = (ctypes.c_ulonglong * 8)()
            libs[tag].trs_debug_stamps(buf, 1)
            tot = float(sum(buf)) or 1.0
            names = ("diag_gemm", "factor_rest", "item_gemm", "item_wait_f", "item_trsm_store", "end_barrier", "wait_d", "chol16")
            if not args.dense:  # the wave-per-matrix kernel's phases
                names = ("tile_loads", "block_update", "factorisation", "load_column+stores", "items", "fence", "-", "-")
            print(f"{tag} stamp shares: " + ", ".join(f"{n} {buf[i] / tot:.3f}" for i, n in enumerate(names)))
        med = {s: float(np.median(v)) for s, v in times[tag].items()}
        mn = {s: float(np.min(v)) for s, v in times[tag].items()}
        total = sum(med.values())
        print(f"{tag:>12}: potrf med {med['potrf']:.3f} min {mn['potrf']:.3f} ms | assemble {med['assemble']:.3f} "
              f"| potrs {med['potrs']:.3f} | recover {med['recover']:.3f} | dofmap {med['dofmap']:.4f} | total {total:.3f} ms "
              f"-> {args.batch / total * 1e3:.0f} solves/s")


if __name__ == "__main__":
    main()
