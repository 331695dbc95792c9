#!/bin/bash
# Run ON the GPU box: one bench line per preset (forward step, batch 256 unless given), results in gpurun_out/presets.jsonl
OUT=$PWD/gpurun_out
mkdir -p $OUT
: > $OUT/presets.jsonl
for spec in "$@"; do
  model=${spec%%:*}; batch=${spec##*:}
  timeout -k 10 280 python bench.py --model $model --batch $batch --steps 8 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 >> $OUT/presets.jsonl || echo "{\"model\": \"$model\", \"failed\": true}" >> $OUT/presets.jsonl
done
echo done
