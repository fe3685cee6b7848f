#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM traffic counters for the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_$C.json 2> $OUT/pmc_$C.err || echo "pmc $C failed" >> $OUT/errors.txt
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.md
cat $OUT/summary.md
