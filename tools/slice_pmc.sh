#!/bin/bash
# HBM traffic of the sliced pipeline (tools/slice_pipeline.py) per variant: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes (never combined with other traces), summed over every kernel of the run.
#   tools/slice_pmc.sh <tag> -> gpurun_out/<tag>/slices.txt
set -u
TAG=${1:-slices}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
STEPS=5
for variant in "4096" "256" "256 --shared" "512 --shared" "1024 --shared"; do
  name=$(echo $variant | tr -d ' -')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${name}_$ctr" -- python3 tools/slice_pipeline.py --only $variant --steps $STEPS > "$OUT/${name}_$ctr.log" 2>&1
  done
  python3 - "$OUT" "$name" "$STEPS" "$variant" <<'PY' >> "$OUT/slices.txt"
import csv, glob, sys
out, name, steps, variant = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    for path in glob.glob(f"{out}/pmc_{name}_{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == ctr and row["Kernel_Name"].find("trs_") >= 0:
                s += float(row["Counter_Value"])
    tot[ctr] = s
# (steps + 1 warm-up + the unsliced reference solve) passes of the pipeline ran; FETCH_SIZE in KiB, doubled on gfx950
runs = steps + 2
gb = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9 / runs
print(f"slices {variant}: read {2 * tot['FETCH_SIZE'] * 1024 / 1e9 / runs:.2f} GB + written {tot['WRITE_SIZE'] * 1024 / 1e9 / runs:.2f} GB = {gb:.2f} GB per step (approx.: {runs} pipeline passes in the run)")
PY
done
cat "$OUT/slices.txt"
