#!/bin/bash
# round 4: residues and pages (read exactly once) loaded non-temporal, so that the count vector of k = 12 survives in the Infinity Cache
set -o pipefail
O=gpurun_out/r04; mkdir -p $O; export PYTHONUNBUFFERED=1
AB_LIBS="libkdbhip.so libkdbhip_ntl.so libkdbhip_ntls.so" AB_STEPS=300 bash tools/ab_libs.sh --no-configs 2>&1 | tee $O/ab_nt_k12.txt
AB_LIBS="libkdbhip.so libkdbhip_ntl.so libkdbhip_ntls.so" AB_STEPS=100 bash tools/ab_libs.sh --no-configs --k 13 2>&1 | tee $O/ab_nt_k13.txt
AB_LIBS="libkdbhip.so libkdbhip_ntl.so" AB_STEPS=64 bash tools/ab_libs.sh --no-configs --k 15 2>&1 | tee $O/ab_nt_k15.txt
