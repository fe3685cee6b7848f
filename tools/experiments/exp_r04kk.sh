#!/bin/bash
# rocprofv3 kernel trace of the ragged, N-bearing batches (10 M reads of 35..150 bp; drop and expansion mode at 0.5 %, 0.05 % and no N)
set -o pipefail
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/prof_r04_ragged; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/experiments/ragged_ab.py 12 > $O/ragged_ab.txt 2> $O/stats.err; echo "rc=$?"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cat $O/ragged_ab.txt | cut -c1-260
head -12 $O/kernel_stats.csv | cut -c1-200
