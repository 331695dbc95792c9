set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > gpurun_out/s20_ops.log 2>&1 || { tail -30 gpurun_out/s20_ops.log; exit 1; }
tail -2 gpurun_out/s20_ops.log
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "large14 or sharp or batch_invariance or repeatable" > gpurun_out/s20_model.log 2>&1 || { tail -30 gpurun_out/s20_model.log; exit 1; }
tail -2 gpurun_out/s20_model.log
python tools/gemm_wave_stamps.py qkv fc 2>&1 | grep -v amdgpu | tail -20
for i in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>> gpurun_out/s20.err | grep -o '"ms_per_step": [0-9.]*'; done
