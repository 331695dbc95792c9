#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: kernel-trace stats of three training steps (tools/train_step_probe.py).
set -e
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/tools/train_step_probe.py
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o run -- python3 $P > $OUT/train_under_rocprof.txt 2> $OUT/prof_train.err
echo done
