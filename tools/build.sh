#!/bin/bash
# Build the in-tree libraries; fail loudly (a silent failed build once cost three GPU runs).
set -e
cd "$(dirname "$0")/.."
out=$(make -C python_stable_3d_truss_analysis_amd/csrc -j4 2>&1) || { echo "$out" | grep -E "error|Stop|Error" ; echo "BUILD FAILED"; exit 1; }
echo "$out" | grep -E "error:" && { echo "BUILD FAILED"; exit 1; }
ls -la --time-style=+%T python_stable_3d_truss_analysis_amd/*.so | awk '{print $6, $7}'
echo "BUILD OK"
