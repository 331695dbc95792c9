set -e
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --model vit-small-patch8-384 --batch 128 --breakdown > gpurun_out/s8_small.json 2> gpurun_out/s8.err
python - <<PY
import json
d=json.loads(open("gpurun_out/s8_small.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["breakdown"].items(): print(k, v)
PY
python tools/attn_probe.py > gpurun_out/s8_attn.log 2>&1; tail -20 gpurun_out/s8_attn.log
