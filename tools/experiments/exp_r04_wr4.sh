#!/bin/bash
set -o pipefail
O=gpurun_out/r04/wr3; mkdir -p $O; export PYTHONUNBUFFERED=1
for rep in 1 2; do
AB_LIBS="libkdbhip.so libkdbhip_NT.so libkdbhip_SC1.so" AB_STEPS=300 bash tools/ab_libs.sh --no-configs 2>&1 | tee -a $O/ab2_k12.txt
done
AB_LIBS="libkdbhip.so libkdbhip_NT.so libkdbhip_SC1.so" AB_STEPS=100 bash tools/ab_libs.sh --no-configs --k 13 2>&1 | tee $O/ab2_k13.txt
AB_LIBS="libkdbhip.so libkdbhip_NT.so" AB_STEPS=64 bash tools/ab_libs.sh --no-configs --k 17 2>&1 | tee $O/ab2_k17.txt
