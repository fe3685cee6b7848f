#!/bin/bash
# 16-bit histogram pass: (a) pages of the next group requested before the adds of this one [slower, not kept]; (b) eight pages in flight per wave
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
for K in 13 17; do
  AB_STEPS=64 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_s2_k$K.txt
done
