#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k17_bins or golden_k13 or deferred_histogram or (random_reads_vs_oracle and (13 or 17))" > $O/t_gg.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_gg.txt
for K in 13 17; do
AB_STEPS=64 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_gg_k$K.txt
done
