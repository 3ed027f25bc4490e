#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats in one run, PMC counters in separate runs (never combined with other traces).
# Usage: tools/profile_pmc.sh <tag> [bench args...]      -> gpurun_out/<tag>/{stats,pmc_*}/...
#   Every pass profiles the SAME bench configuration: the user's bench args (e.g. --dense) go to the PMC passes
#   and to the kernel-trace pass alike; only the step counts differ (PMC: 2 steps; kernel trace: the default
#   number, so that the cold first launch does not weigh on the averages).  Env STATS_ARGS overrides the
#   kernel-trace pass's arguments altogether.
set -u
TAG=${1:-prof}; shift || true
QUIET="--no-cpu-baseline --cpu-pool-seconds 0 --no-dense-ref --no-pcie --cube-batch 0 --dataset-samples 0 --cube-total 0 --dataset-total 0"
USER_ARGS="$*"
ARGS="--steps 2 --warmup 1 $QUIET $USER_ARGS"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
STATS_ARGS=${STATS_ARGS:-$QUIET $USER_ARGS}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $STATS_ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_lds" -- python3 bench.py $ARGS > "$OUT/pmc_lds.log" 2>&1
find "$OUT" -name "*.csv" | head -30
