#!/bin/bash
# Everything under profiles/ for one tag, in one gpurun call (run from the repo root ON THE GPU BOX):
#   tools/capture_profiles.sh r04a
# -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_summary.json      rocprofv3 kernel stats + PMC passes of the default bench.py step (bar-942 x 4096)
#    gpurun_out/<tag>_dense_kernel_stats.csv, <tag>_dense_pmc_summary.json   the same for the dense-mode step (bench.py --dense)
#    gpurun_out/<tag>_cube_kernel_stats.csv, <tag>_cube_pmc.txt     the same for the ragged cube step (tools/cube_step.py)
#    gpurun_out/<tag>_bench.json.log                                the full default bench line of the same sources
#    gpurun_out/<tag>_potrf_traffic.json                            the stamped traffic record bench.py quotes (-> profiles/potrf_traffic.json)
# Counters are collected in their own passes, never together with a trace (tools/profile_pmc.sh, tools/cube_pmc.sh).
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out
timeout 900 tools/profile_pmc.sh ${TAG}_pmc > $OUT/${TAG}_pmc_files.log 2>&1
python3 tools/summarize_pmc.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_summary.json > $OUT/${TAG}_sum.log 2>&1
find $OUT/${TAG}_pmc/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
python3 - "$OUT/${TAG}_pmc_summary.json" "$OUT/${TAG}_potrf_traffic.json" <<'PY' > $OUT/${TAG}_traffic.log 2>&1
import json, os, shutil, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import make_traffic_record as m
os.makedirs("profiles", exist_ok=True)
m.main(sys.argv[1], "trs_potrf_narrow_kernel<false, 2>", 4096, "profile")
shutil.copy("profiles/potrf_traffic.json", sys.argv[2])
PY
# the DENSE-mode launch of the same path (bench.py --dense: envelope off, every tile stored and factored) - the launch
# SURVEY 8d's n^3/3 figure describes, `roofline.frac_dense_8d` of the bench line
timeout 900 tools/profile_pmc.sh ${TAG}_dense_pmc --dense > $OUT/${TAG}_dense_pmc_files.log 2>&1
python3 tools/summarize_pmc.py $OUT/${TAG}_dense_pmc $OUT/${TAG}_dense_pmc_summary.json > $OUT/${TAG}_dense_sum.log 2>&1
find $OUT/${TAG}_dense_pmc/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_dense_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cube_stats -- python3 tools/cube_step.py --lanes 1 > $OUT/${TAG}_cube_stats.log 2>&1   # (ONE lane: per-kernel durations without the overlap of the default four)
find $OUT/${TAG}_cube_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_cube_kernel_stats.csv
timeout 600 tools/cube_pmc.sh ${TAG}_cubepmc > /dev/null 2>&1
cp $OUT/${TAG}_cubepmc/cube_pmc.txt $OUT/${TAG}_cube_pmc.txt
timeout 900 python3 bench.py > $OUT/${TAG}_bench.json.log 2> $OUT/${TAG}_bench.err
ls -la $OUT/${TAG}_*.csv $OUT/${TAG}_*.json $OUT/${TAG}_*.txt $OUT/${TAG}_*.log 2>/dev/null | awk '{print $5, $9}'
