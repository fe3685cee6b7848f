#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats, HBM traffic counters and SQ/LDS counters for the bench command,
# each in its own rocprofv3 pass.   Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r02}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# every pass runs the SAME steps: at k >= 13 the histogram pass runs once per flush, so its per-launch bytes depend on the batches a flush holds
STEPS=${STEPS:-20}
COMMON="--no-cpu-baseline --no-extra-regions --no-configs --steps $STEPS --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $COMMON "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py $COMMON "$@" > $OUT/bench_$C.json 2> $OUT/pmc_$C.err || echo "pmc $C failed" >> $OUT/errors.txt
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq_a -- python3 bench.py $COMMON "$@" > $OUT/bench_sq_a.json 2> $OUT/pmc_sq_a.err || echo "pmc sq_a failed" >> $OUT/errors.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_b -- python3 bench.py $COMMON "$@" > $OUT/bench_sq_b.json 2> $OUT/pmc_sq_b.err || echo "pmc sq_b failed" >> $OUT/errors.txt
python3 tools/summarize_prof.py $OUT > $OUT/summary.md
# keep what gets committed small: the kernel-stats CSV, the summary and the two JSON files
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
head -60 $OUT/summary.md
