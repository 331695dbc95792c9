#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: the evidence files of a round, into gpurun_out/ (copied into profiles/ by hand /
# tools/collect_profiles.py).  Usage: bash tools/round_evidence.sh rNN
# Before sending: python tools/build_variant.py stamps gemm.hip -DOVHIP_STAMPS=1   (the stamp tools load libovhip_stamps.so)
set -e
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --breakdown > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"; tail -c 600 $OUT/${TAG}_bench.json | head -c 300; echo
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --precision fp8 > $OUT/${TAG}_bench_fp8.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --precision fp8-mixed > $OUT/${TAG}_bench_fp8_mixed.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 5 --warmup 3 --cpu-seconds 0 --precision fp8 --micro-batches 16 > $OUT/${TAG}_bench_fp8_mb16.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 5 --warmup 3 --cpu-seconds 0 --precision fp8-mixed --micro-batches 16 > $OUT/${TAG}_bench_fp8_mixed_mb16.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --model vit-small-patch8-384 --batch 128 > $OUT/${TAG}_bench_small8_384_b128.json 2>> $OUT/${TAG}_bench.err
echo "variants done"
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
bash tools/profile_pmc.sh > $OUT/profile_pmc.log 2>&1
echo "profiles done"
python tools/gemm_stamps.py > $OUT/${TAG}_gemm_stamps.log 2>&1
python tools/gemm_wave_stamps.py qkv out fc proj > $OUT/${TAG}_gemm_wave_stamps.log 2>&1
python tools/dbg/skinny_cross.py > $OUT/${TAG}_skinny_cross.log 2>&1
python tools/train_step_probe.py > $OUT/${TAG}_train_step.log 2>&1
python tools/optim_probe.py > $OUT/${TAG}_optim.log 2>&1
python tools/graph_probe.py > $OUT/${TAG}_graph.log 2>&1
python tools/attn_probe.py > $OUT/${TAG}_attn.log 2>&1
echo "all done"
