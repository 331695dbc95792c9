set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "rowparts or gemm_residual or variants" > gpurun_out/s17_ops.log 2>&1 || { tail -40 gpurun_out/s17_ops.log; exit 1; }
tail -2 gpurun_out/s17_ops.log
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "tiny or large14 or small8 or batch_invariance or sharp" > gpurun_out/s17_model.log 2>&1 || { tail -40 gpurun_out/s17_model.log; exit 1; }
tail -2 gpurun_out/s17_model.log
for e in 1 0 1 0; do
  OVHIP_ROWPARTS=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s17_bench_e$e.json 2>> gpurun_out/s17_bench.err
  echo "== bench ROWPARTS=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s17_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["loss"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
