#!/bin/bash
# lens_kernel: one offset load per record; count_smallk_kernel: wrap test as NOT-AND / min3
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "k8_lds or errors_raise or rejects_offsets or bad_layout or few_bases or 9001 or uniform_length or random_reads_vs_oracle" > $O/t_dd.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_dd.txt
for K in 8 12; do
  AB_STEPS=128 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_dd_k$K.txt
done
