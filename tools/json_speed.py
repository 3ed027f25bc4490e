#!/usr/bin/env python3
"""Bulk JSON reader: 1e5 cube-7 documents (files) -> PackedBatch, native (csrc/jsonpack.c, OpenMP) against
the Python path (json.load + batch.pack_json) on a sample; run on the GPU box's host cores."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from python_stable_3d_truss_analysis_amd import batch
from python_stable_3d_truss_analysis_amd import generate as gen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = gen.generate_cube_batch([7] * N, gridRange=(5, 5, 5), seed=1)
tmp = tempfile.mkdtemp(prefix="trs_json_")
t0 = time.perf_counter()
paths = []
for b in range(N):
    path = os.path.join(tmp, f"cube-7_case_{b}.json")
    with open(path, "w") as fh:
        json.dump(gen.packed_to_json(p, b), fh)
    paths.append(path)
print(f"wrote {N} files in {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
got = batch.pack_json_files(paths)
t_native = time.perf_counter() - t0
sample = paths[: max(1000, N // 50)]
t0 = time.perf_counter()
want = batch.pack_json([json.load(open(q)) for q in sample])
t_py = (time.perf_counter() - t0) * N / len(sample)
for f in want.__dataclass_fields__:
    assert np.array_equal(getattr(want, f), getattr(got.take(np.arange(len(sample))).trimmed(), f)
                          if f not in ("nJ", "nM", "dim", "n_free") else getattr(got, f)[:len(sample)]), f
for f in p.__dataclass_fields__:
    assert np.array_equal(getattr(got, f), getattr(p, f)), f
print(json.dumps({"files": N, "native_s": t_native, "files_per_s": N / t_native,
                  "python_s_extrapolated": t_py, "speedup": t_py / t_native, "cpus": os.cpu_count()}))
for q in paths:
    os.remove(q)
os.rmdir(tmp)
