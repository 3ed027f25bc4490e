mkdir -p gpurun_out/r6
for k in 1 2 3; do
  timeout 400 python bench.py --no-pcie --no-dense-ref --large-trusses 0 --dataset-total 0 > gpurun_out/r6/rep_$k.json 2> gpurun_out/r6/rep_$k.err
done
python - <<'PY'
import json
for k in (1, 2, 3):
    l = [x for x in open(f"gpurun_out/r6/rep_{k}.json") if x.startswith("{")]
    b = json.loads(l[0])
    r = b["repeats"]
    print(f"run {k}: value {b['value'] / 1e6:.3f} M (first region, {b['steps']} steps, {b['ms_per_step']:.4f} ms/step); repeats min/median/max "
          f"{r['value']['min'] / 1e6:.3f} / {r['value']['median'] / 1e6:.3f} / {r['value']['max'] / 1e6:.3f} M; potrf launch {b['roofline']['avg_launch_ms']:.4f} ms, "
          f"frac {b['roofline']['frac']:.3f}; cube {b['cube_batch']['ms_per_step']:.2f} ms = {b['cube_batch']['value'] / 1e6:.3f} M, hbm frac {b['cube_batch']['roofline']['hbm']['frac']:.3f}; "
          f"dataset {b['dataset']['value'] / 1e3:.0f} K samples/s")
PY
