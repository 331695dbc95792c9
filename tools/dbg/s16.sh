set -e
mkdir -p gpurun_out
for v in 0 4; do
OVHIP_GEMM_VARIANT=$v python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --model vit-small-patch8-384 --batch 128 --breakdown > gpurun_out/s16_v$v.json 2>> gpurun_out/s16.err
python - <<PY
import json
d=json.loads(open("gpurun_out/s16_v$v.json").read().strip().split("\n")[-1])
print("variant $v", d["ms_per_step"], d["value"])
for k,v in d["breakdown"].items(): print("   ", k, v if not isinstance(v,dict) else {a:v[a] for a in ("ms","launches","achieved") if a in v})
PY
done
