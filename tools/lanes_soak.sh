#!/bin/bash
# Soak of the multi-lane paths through PyTorch streams after the joint-order race fix (EXPERIMENTS R5.1): the
# configurations that stalled or corrupted results in round 4 (R4.9), each in processes of its own under a timeout.
#   tools/lanes_soak.sh [processes] [iterations]    -> gpurun_out/soak/
cd "$(dirname "$0")/.."
P=${1:-3}; N=${2:-240}
OUT=gpurun_out/soak; mkdir -p $OUT
run() {  # tag command...
  local tag=$1; shift
  for p in $(seq 1 $P); do
    timeout 400 "$@" > $OUT/${tag}_$p.log 2>&1
    echo "$tag process $p: exit $? | $(tail -1 $OUT/${tag}_$p.log | cut -c1-160)"
  done
}
run dataset_l3 python3 tools/lanes_stress.py dataset 3 $N
run dataset_l2 python3 tools/lanes_stress.py dataset 2 $N
run fresh_l2   python3 tools/lanes_stress.py fresh 2 $N
run kept_l2    python3 tools/lanes_stress.py kept 2 $N
run kept_l3    python3 tools/lanes_stress.py kept 3 $N
P=1 run diff_l3 python3 tools/lanes_diff.py 3 150
