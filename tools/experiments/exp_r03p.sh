#!/bin/bash
# round 3: the tile image's index kept in a scalar register (libkdbhip_exp.so) against HEAD (libkdbhip_base.so)
set -e
OUT=gpurun_out/r03p
mkdir -p $OUT
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=96 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
