#!/bin/bash
cd /root/repo
for r in 1 0; do
  OVHIP_GEMM_ROTATE=$r python bench.py --model vit-small-patch8-384 --batch 128 --cpu-seconds 0 --breakdown --steps 10 --warmup 3 > gpurun_out/s8_rot$r.json 2>> gpurun_out/s8_ab.err
  python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/s8_rot$r.json').read().strip().splitlines()[-1])
print('rotate=$r', d['value'], d['ms_per_step'])
b=d.get('breakdown') or {}
print({k:(round(v['ms'],3) if isinstance(v,dict) and 'ms' in v else v) for k,v in b.items()} if isinstance(b,dict) else b)
PY
done
