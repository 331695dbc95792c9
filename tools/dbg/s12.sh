set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "layernorm or ln_fold" 2>&1 | tail -2
for e in 64 8 4 16 64 8; do
  OVHIP_ROWSTATS_WGS=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s12_bench_e$e.json 2>> gpurun_out/s12_bench.err
  echo "== bench ROWSTATS_WGS=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s12_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["breakdown"]["ln"])
PY
done
