#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: kernel-trace stats of the default bench, then the two PMC passes.
# rocprofv3 is given `python3 bench.py` directly (no env/bash hop: the profiler initialises the GPU before the program starts).
set -e
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/bench.py
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o run -- python3 $B --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $B --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $B --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo done
