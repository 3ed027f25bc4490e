L=python_stable_3d_truss_analysis_amd/libtrs_hip.so
for cfg in "--lanes 4 --noise 1" "--lanes 5 --noise 0" "--lanes 3 --noise 2" "--lanes 4 --noise 0"; do
  timeout 280 tools/repro_streams $L --trusses 16384 --steps 400 --variants 2 $cfg 2>&1 | grep -E "^RESULT|STALL" | tail -1
done
for n in 0 1 2; do
  timeout 280 tools/repro_families $L --trusses 16384 --small 16384 --steps 400 --noise $n 2>&1 | grep -E "^RESULT|STALL" | tail -1
done
