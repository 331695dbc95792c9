set -e
mkdir -p gpurun_out
for v in "" l2 l2d2 l1 "" l2; do
  if [ -n "$v" ]; then export OVHIP_LIB=libovhip_$v.so; else unset OVHIP_LIB; fi
  STAMPS=1 python tools/gemm_stamps.py > gpurun_out/s3_stamps_${v:-base}.log 2>&1
  echo "== LIB=${v:-base}"; grep -h "avg of\|K-tile us\|kernel " gpurun_out/s3_stamps_${v:-base}.log | cut -c1-230
done
for v in "" l2 l2d2 "" l2 l2d2; do
  if [ -n "$v" ]; then export OVHIP_LIB=libovhip_$v.so; else unset OVHIP_LIB; fi
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown > gpurun_out/s3_bench_${v:-base}.json 2>> gpurun_out/s3_bench.err
  echo "== bench ${v:-base}"; python - <<PY
import json
d=json.loads(open("gpurun_out/s3_bench_${v:-base}.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], {k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.get("breakdown",{}).items()})
PY
done
