set -e
mkdir -p gpurun_out
for v in "" halfgelu "" halfgelu; do
  if [ -n "$v" ]; then export OVHIP_LIB=libovhip_$v.so; else unset OVHIP_LIB; fi
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s18_bench_${v:-base}.json 2>> gpurun_out/s18.err
  echo "== bench ${v:-base}"; python - <<PY
import json
d=json.loads(open("gpurun_out/s18_bench_${v:-base}.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
