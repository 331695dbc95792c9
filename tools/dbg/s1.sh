set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > gpurun_out/s1_tests.log 2>&1 || { tail -30 gpurun_out/s1_tests.log; exit 1; }
tail -3 gpurun_out/s1_tests.log
python tools/gemm_stamps.py > gpurun_out/s1_stamps_base.log 2>&1
OVHIP_GEMM_EXP=1 python tools/gemm_stamps.py > gpurun_out/s1_stamps_skip0.log 2>&1
python tools/gemm_stamps.py > gpurun_out/s1_stamps_base2.log 2>&1
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
OVHIP_GEMM_EXP=1 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/s1_bench_skip0.json 2>> gpurun_out/s1_bench.err
grep -h "avg of\|K-tile us\|kernel " gpurun_out/s1_stamps_base.log gpurun_out/s1_stamps_skip0.log gpurun_out/s1_stamps_base2.log
cat gpurun_out/s1_bench.json gpurun_out/s1_bench_skip0.json | cut -c1-400
