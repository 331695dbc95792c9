#!/bin/bash
# reverse row walk of the residual-epilogue GEMMs (bit 3) / GELU GEMMs (bit 1): does the consumer find the producer's tail in the Infinity Cache?
cd /root/repo
for r in 0 8 0 8 2 10; do
  OVHIP_GEMM_REVERSE=$r python bench.py --cpu-seconds 0 --steps 20 --warmup 3 --breakdown > gpurun_out/rev_l14_$r.json 2>> gpurun_out/rev_ab.err
  OVHIP_GEMM_REVERSE=$r python bench.py --model vit-small-patch8-384 --batch 128 --cpu-seconds 0 --breakdown --steps 10 --warmup 3 > gpurun_out/rev_s8_$r.json 2>> gpurun_out/rev_ab.err
  python - <<PY
import json
for f in ('rev_l14_$r','rev_s8_$r'):
    d=json.loads(open('/root/repo/gpurun_out/'+f+'.json').read().strip().splitlines()[-1])
    b=d.get('breakdown') or {}
    print(f, d['value'], d['ms_per_step'], {k:round(v,3) for k,v in b.items() if isinstance(v,(int,float))})
PY
done
