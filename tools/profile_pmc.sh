#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: MFMA-busy / LDS-conflict counters of the default bench (own passes,
# kernel-trace only, as the pool requires).  The per-kernel sums are read from the counter_collection.csv files by hand
# into profiles/rNN_pmc.json (no post-processing script).
set -e
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/bench.py
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -o run -- python3 $B --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_lds -o run -- python3 $B --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
echo done
