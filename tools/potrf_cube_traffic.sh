export TMPDIR=/tmp
# HBM bytes of the cube batch's factorisation per knock-out build:  tools/potrf_cube_traffic.sh [tags ...]
for t in ${@:-default nifb ndl nkl niu nis}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/trf_${t}_$c -- python3 tools/potrf_cube_traffic.py $t > gpurun_out/trf_${t}_$c.log 2>&1
    n=$(grep -o "measured dispatches: [0-9]*" gpurun_out/trf_${t}_$c.log | grep -o "[0-9]*$")
    echo "$t $c ($n dispatches): $(python3 tools/potrf_cube_traffic_sum.py gpurun_out/trf_${t}_$c ${n:-13})"
  done
done
