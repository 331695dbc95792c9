#!/bin/bash
# usage: ab_env.sh "ENV=val ..." "ENV=val ..." ...   -- one L/14 bench line (with breakdown) per environment, twice round-robin
cd /root/repo
for rep in 1 2; do
for envs in "$@"; do
  env $envs python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('[$envs]', d['value'], d['ms_per_step'], {k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in b.items()})"
done
done
