set -e
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>> gpurun_out/s13.err | grep -o '"ms_per_step": [0-9.]*'; }
run A=1
run OVHIP_GEMM_SKINNY_TILES=0
run OVHIP_NO_TAIL_SPLIT=1
run OVHIP_NO_TAIL_SPLIT=1 OVHIP_GEMM_SKINNY_TILES=0
run A=1
run OVHIP_GEMM_SKINNY_TILES=0
run OVHIP_NO_TAIL_SPLIT=1
