#!/bin/bash
# usage: ab_lib.sh "<bench args>" lib1.so lib2.so ...  -- one bench line per library variant (tools/build_variant.py), twice round-robin
cd /root/repo
args=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  OVHIP_LIB=$lib python bench.py $args --cpu-seconds 0 --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('[$lib] [$args]', d['value'], d['ms_per_step'], {k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in b.items()})"
done
done
