#!/bin/bash
# histogram adds with compile-time indices only (page_hist_kernel<true>, count_smallk_kernel): A/B against the previous build + the wrap tests
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k17_bins or k8_lds or random_reads_vs_oracle or deferred_histogram" > $O/t_m.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_m.txt
for K in 13 8 17; do
  AB_STEPS=64 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_m_k$K.txt
done
