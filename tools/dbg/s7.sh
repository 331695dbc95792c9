set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "fp8" -s > gpurun_out/s7_fp8_tests.log 2>&1 || { tail -40 gpurun_out/s7_fp8_tests.log; exit 1; }
grep -h "fp8\|passed\|failed" gpurun_out/s7_fp8_tests.log | tail -20
for p in "bf16" "fp8-mixed" "fp8"; do
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --precision $p > gpurun_out/s7_bench_$p.json 2>> gpurun_out/s7_bench.err
  echo "== $p"; grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*\|"loss": [0-9.]*\|"frac": [0-9.]*' gpurun_out/s7_bench_$p.json | tr '\n' ' '; echo
done
