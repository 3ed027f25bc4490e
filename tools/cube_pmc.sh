#!/bin/bash
# PMC counters of the ragged cube-truss step (tools/cube_step.py), summed per kernel over the run:
# matrix-core busy fraction, wait share, HBM bytes.  Separate rocprofv3 passes, counters only.
#   tools/cube_pmc.sh <tag> -> gpurun_out/<tag>/cube_pmc.txt
set -u
TAG=${1:-cubepmc}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 tools/cube_step.py --steps 2 --lanes 1 > "$OUT/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/cube_step.py --steps 2 --lanes 1 > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/cube_step.py --steps 2 --lanes 1 > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/cube_pmc.txt"
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
for path in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
print("# ragged cube step (65 536 trusses, 3 passes of the step per run): per kernel, summed over the run")
for name, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not name.startswith("trs_"):
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0) if gui else 0.0
    wait = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0
    gb = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024 / 1e9
    print(f"{name:45s} gui_active/8 {gui / 8 / 1e6:9.2f} Mcyc | mfma busy per SIMD {busy:5.3f} | wait/wave {wait:5.3f} | HBM {gb:8.2f} GB")
PY
