#!/bin/bash
# Build experiment variants of the solver library with different potrf tuning macros:
#   tools/build_variants.sh "tag1:-DTRS_POTRF_RS=2 -DTRS_POTRF_WAVES_PER_SIMD=4" "tag2:..."
#   ONLY="potrf" tools/build_variants.sh ...   recompiles just the named sources and links the product build's other objects
# -> python_stable_3d_truss_analysis_amd/variants/libtrs_<tag>.so   (git-ignored; travels with gpurun)
#   PATCH=1 tools/build_variants.sh ...         builds from a scratch copy of csrc/ with tools/patches/experiment_code_r1_r5.diff
#                                               applied (the TRS_EXP_* / knock-out switches of rounds 1-5; tools/patches/README.md)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/python_stable_3d_truss_analysis_amd/csrc"
mkdir -p ../variants
if [ -n "${PATCH:-}" ]; then
  SCRATCH=$(mktemp -d /tmp/trs_csrc_XXXX)
  mkdir -p "$SCRATCH/python_stable_3d_truss_analysis_amd" "$SCRATCH/include"
  cp -r "$ROOT/python_stable_3d_truss_analysis_amd/csrc" "$SCRATCH/python_stable_3d_truss_analysis_amd/"
  cp "$ROOT"/include/*.h "$SCRATCH/include/"
  (cd "$SCRATCH" && patch -p1 -s < "$ROOT/tools/patches/experiment_code_r1_r5.diff")
  VARDIR="$ROOT/python_stable_3d_truss_analysis_amd/variants"
  cd "$SCRATCH/python_stable_3d_truss_analysis_amd/csrc"
else
  VARDIR=../variants
fi
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  objs=""
  for f in dofmap assemble potrf potrs recover small graphfeat order rows cubegen capi; do
    if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $f "; then objs="$objs $f.o"; continue; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -c $f.hip -o /tmp/var_${tag}_$f.o &
    objs="$objs /tmp/var_${tag}_$f.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $VARDIR/libtrs_$tag.so
  echo "built variants/libtrs_$tag.so ($flags)"
done
