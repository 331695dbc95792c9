set -e
mkdir -p gpurun_out
for e in 0 2 4 8 0 2 3; do
  OVHIP_GEMM_EXP=$e python tools/gemm_stamps.py > gpurun_out/s2_stamps_e$e.log 2>&1
  echo "== EXP=$e"; grep -h "avg of\|K-tile us\|kernel " gpurun_out/s2_stamps_e$e.log
done
for e in 0 2 0 2; do
  OVHIP_GEMM_EXP=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/s2_bench_e$e.json 2>> gpurun_out/s2_bench.err
  echo "== bench EXP=$e"; cut -c1-200 gpurun_out/s2_bench_e$e.json | grep -o '"ms_per_step": [0-9.]*'
done
