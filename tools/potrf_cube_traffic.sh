export TMPDIR=/tmp
for t in default nifb ndl nkl niu nis; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/trf_${t}_$c -- python3 tools/potrf_cube_traffic.py $t > gpurun_out/trf_${t}_$c.log 2>&1
    echo "$t $c: $(python3 tools/potrf_cube_traffic_sum.py gpurun_out/trf_${t}_$c 13)"
  done
done
