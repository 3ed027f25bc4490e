#!/bin/bash
# rocprofv3 evidence of the large-truss route (tools/large_step.py = bench.py's `large_truss` leg):
#   tools/large_pmc.sh <tag> -> gpurun_out/<tag>/{large.json, kernel_stats.csv, pmc.txt}   (counters in passes of their own)
set -u
TAG=${1:-largepmc}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 tools/large_step.py > "$OUT/large.json" 2> "$OUT/large.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/large_step.py > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 tools/large_step.py > "$OUT/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/large_step.py > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/large_step.py > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/pmc.txt"
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
tot, calls = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for path in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "WRITE_SIZE":
            calls[name] += 1
print("# large-truss route (512 cube trusses, 8x8x8 grid, 300..400 cubes): per kernel, per LAUNCH (12 steps per run)")
for name, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not name.startswith("trs_") or not calls[name]:
        continue
    n = calls[name]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0) if gui else 0.0
    wait = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0
    rd, wr = 2 * c.get("FETCH_SIZE", 0.0) * 1024 / 1e9 / n, c.get("WRITE_SIZE", 0.0) * 1024 / 1e9 / n
    print(f"{name:42s} launches {n:3d} | mfma busy per SIMD {busy:5.3f} | wait/wave {wait:5.3f} | read {rd:7.2f} GB | written {wr:7.2f} GB | HBM {rd + wr:7.2f} GB per launch")
PY
rm -rf "$OUT/stats" "$OUT/sq" "$OUT/fetch" "$OUT/write"
