set -e
mkdir -p gpurun_out
for e in 0 64 0 64 128 192; do
  OVHIP_BIG_PAD=$e python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s11_bench_e$e.json 2>> gpurun_out/s11_bench.err
  echo "== bench BIG_PAD=$e"; python - <<PY
import json
d=json.loads(open("gpurun_out/s11_bench_e$e.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
