set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > gpurun_out/s6_tests.log 2>&1 || { tail -30 gpurun_out/s6_tests.log; exit 1; }
tail -2 gpurun_out/s6_tests.log
for e in 1 0 1 0; do
  OVHIP_GEMM_EXP=$e python tools/gemm_stamps.py > gpurun_out/s6_stamps_e$e.log 2>&1
  echo "== EXP=$e (1 = no touch)"; grep -h "avg of\|K-tile us\|kernel " gpurun_out/s6_stamps_e$e.log | grep -A2 "^out\|^proj" | cut -c1-200
done
OVHIP_GEMM_EXP=0 python tools/gemm_wave_stamps.py out proj 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s6_wave.log; cat gpurun_out/s6_wave.log
for e in 1 0 1 0; do
  OVHIP_GEMM_EXP=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s6_bench_e$e.json 2>> gpurun_out/s6_bench.err
  echo "== bench EXP=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s6_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
