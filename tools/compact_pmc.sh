#!/bin/bash
# HBM counters of the ragged cube step in slab form and in compact form (TRS_OPTIONS=compact=1), per kernel:
#   tools/compact_pmc.sh <tag> -> gpurun_out/<tag>/{slab,compact}.txt     (separate --pmc passes, counters only)
set -u
TAG=${1:-compactpmc}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for form in slab compact; do
  if [ $form = compact ]; then export TRS_OPTIONS="compact=1"; else unset TRS_OPTIONS; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d "$OUT/${form}_fetch" -- python3 tools/cube_step.py --steps 2 --lanes 1 > "$OUT/${form}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${form}_write" -- python3 tools/cube_step.py --steps 2 --lanes 1 > "$OUT/${form}_write.log" 2>&1
  python3 - "$OUT" $form <<'PY' | tee "$OUT/$form.txt"
import csv, glob, sys
from collections import defaultdict
out, form = sys.argv[1], sys.argv[2]
tot, calls = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for path in glob.glob(f"{out}/{form}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "WRITE_SIZE":
            calls[name] += 1
print(f"# ragged cube step, {form} form, 65 536 trusses, one lane; 6 steps per run (cube_step.py --steps 2): per kernel, per STEP")
steps = 6.0
grand = 0.0
for name, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not name.startswith("trs_"):
        continue
    rd, wr = 2 * c.get("FETCH_SIZE", 0.0) * 1024 / 1e9 / steps, c.get("WRITE_SIZE", 0.0) * 1024 / 1e9 / steps
    grand += rd + wr
    print(f"{name:45s} launches/step {calls[name] / steps:6.1f} | read {rd:8.2f} GB | written {wr:8.2f} GB | HBM {rd + wr:8.2f} GB")
print(f"{'all trs_ kernels':45s} {'':21s} HBM {grand:8.2f} GB per step")
PY
done
