set -e
mkdir -p gpurun_out
python tools/gemm_wave_stamps.py out proj > gpurun_out/s5_wave_stamps.log 2>&1
cat gpurun_out/s5_wave_stamps.log | grep -v amdgpu.ids
