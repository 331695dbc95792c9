set -e
mkdir -p gpurun_out
for p in fp8 fp8-mixed; do
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --precision $p --breakdown > gpurun_out/s15_$p.json 2>> gpurun_out/s15.err
python - <<PY
import json
d=json.loads(open("gpurun_out/s15_$p.json").read().strip().split("\n")[-1])
print("$p", d["ms_per_step"], d["value"], d["roofline"]["frac"])
for k,v in d["breakdown"].items(): print("   ", k, v if not isinstance(v,dict) else {a:v[a] for a in ("ms","launches","achieved") if a in v})
PY
done
