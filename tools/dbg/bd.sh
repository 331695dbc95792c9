#!/bin/bash
# usage: bd.sh model:batch ...  -- bench line with the per-class breakdown per preset
cd /root/repo
for spec in "$@"; do
  model=${spec%%:*}; batch=${spec##*:}
  python bench.py --model $model --batch $batch --steps 8 --warmup 3 --cpu-seconds 0 --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('$model', d['value'], d['ms_per_step'], {k:((round(v['ms'],3), round(v['achieved'])) if isinstance(v,dict) else round(v,3)) for k,v in b.items()})"
done
