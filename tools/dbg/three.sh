#!/bin/bash
# the three configurations used for A/B work, one line each (with the per-class breakdown)
cd /root/repo
run() { python bench.py --model $1 --batch $2 --steps $3 --warmup 3 --cpu-seconds 0 --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('$1'[:14], d['value'], d['ms_per_step'], {k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in b.items()})"; }
run vit-large-patch14-224 256 20
run vit-small-patch8-384 128 10
run vit-tiny-patch16-160 1024 10
run vit-base-patch16-224 256 10
