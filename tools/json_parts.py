import sys, os, json, time, ctypes, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from python_stable_3d_truss_analysis_amd import batch, generate as gen
from python_stable_3d_truss_analysis_amd.generate import _load
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = gen.generate_cube_batch([7] * 2000, gridRange=(5, 5, 5), seed=1)
texts = [json.dumps(gen.packed_to_json(p, b)) for b in range(2000)]
tmp = tempfile.mkdtemp(prefix="trs_json_")
paths = []
t0 = time.perf_counter()
for b in range(N):
    path = os.path.join(tmp, f"c{b}.json")
    with open(path, "w") as fh: fh.write(texts[b % 2000])
    paths.append(path)
print("wrote", N, "files", time.perf_counter() - t0, "mean bytes", np.mean([len(t) for t in texts]))
lib = _load()
for rep in range(2):
    t0 = time.perf_counter()
    raw = [os.fsencode(q) for q in paths]; arr = (ctypes.c_char_p * N)(*raw); bufs = (ctypes.c_void_p * N)(); lens = np.zeros([N], dtype=np.int64)
    t1 = time.perf_counter()
    lib.trs_json_read_files.restype = ctypes.c_int
    rc = lib.trs_json_read_files(ctypes.c_int(N), arr, bufs, lens.ctypes.data_as(ctypes.c_void_p))
    t2 = time.perf_counter()
    lib.trs_json_free_files(ctypes.c_int(N), bufs)
    t3 = time.perf_counter()
    got = batch.pack_json_files(paths)
    t4 = time.perf_counter()
    allt = [texts[b % 2000] for b in range(N)]
    t5 = time.perf_counter()
    got2 = batch.pack_json_texts(allt)
    t6 = time.perf_counter()
    print(f"python prep {t1-t0:.3f}  read_files {t2-t1:.3f}  free {t3-t2:.3f}  pack_json_files total {t4-t3:.3f}  pack_json_texts (in memory) {t6-t5:.3f}")
for q in paths: os.remove(q)
os.rmdir(tmp)
