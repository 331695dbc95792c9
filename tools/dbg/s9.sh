set -e
mkdir -p gpurun_out
for e in 0 1 0 1; do
  OVHIP_GEMM_GELU_LDS=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s9_bench_e$e.json 2>> gpurun_out/s9_bench.err
  echo "== bench GELU_LDS=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s9_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
OVHIP_GEMM_GELU_LDS=1 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm_ln_fold or gelu" 2>&1 | tail -2
