#!/bin/bash
# usage: tools/bench_env_ab.sh "VAR=a" "VAR=b" ...   (each arm = one env assignment string; runs bench.py --breakdown per arm, same box)
for arm in "$@"; do
  echo "ARM $arm"
  env $arm timeout -k 10 250 python bench.py --cpu-seconds 0 --breakdown 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['ms'] for k,v in d['breakdown'].items() if k!='step_ms'})"
done
