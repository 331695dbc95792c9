set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > gpurun_out/s19_ops.log 2>&1 || { tail -30 gpurun_out/s19_ops.log; exit 1; }
tail -2 gpurun_out/s19_ops.log
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "large14 or sharp or batch_invariance" > gpurun_out/s19_model.log 2>&1 || { tail -30 gpurun_out/s19_model.log; exit 1; }
tail -2 gpurun_out/s19_model.log
for e in 0 1 0 1; do
  OVHIP_GEMM_DEFER=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s19_bench_e$e.json 2>> gpurun_out/s19_bench.err
  echo "== bench DEFER=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s19_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["loss"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
