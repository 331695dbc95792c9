#!/bin/bash
# usage: ab_env3.sh "ENV=val" ...  -- S/8, B/16, Ti/16 bench lines per environment, twice round-robin
cd /root/repo
run() { env $1 python bench.py --model $2 --batch $3 --steps 10 --warmup 3 --cpu-seconds 0 --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('[$1]', '$2'[:14], d['value'], d['ms_per_step'], {k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in b.items() if k in ('gemm_fc','gemm_proj','gemm_qkv')})"; }
for rep in 1 2; do
for envs in "$@"; do
  run "$envs" vit-small-patch8-384 128
  run "$envs" vit-base-patch16-224 256
  run "$envs" vit-tiny-patch16-160 1024
done
done
