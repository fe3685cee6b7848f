#!/bin/bash
# lens_kernel over 4096 workgroups instead of 1024
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
for K in 8 12; do
  AB_STEPS=128 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_ee_k$K.txt
done
