set -e
mkdir -p gpurun_out
for e in 0 1 0 1; do
  OVHIP_GEMM_EXP=$e python tools/gemm_stamps.py > gpurun_out/s4_stamps_e$e.log 2>&1
  echo "== EXP=$e"; grep -h "avg of\|K-tile us\|kernel " gpurun_out/s4_stamps_e$e.log | cut -c1-200
done
for e in 0 1 0 1; do
  OVHIP_GEMM_EXP=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/s4_bench_e$e.json 2>> gpurun_out/s4_bench.err
  echo "== bench EXP=$e"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/s4_bench_e$e.json
done
