set -e
mkdir -p gpurun_out
for b in 256 255 256 255; do echo "== batch $b"; python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --batch $b 2>> gpurun_out/s14.err | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*' | tr '\n' ' '; echo; done
