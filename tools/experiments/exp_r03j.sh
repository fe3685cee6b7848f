#!/bin/bash
# round 3: hoods of the next tile read under the flush (libkdbhip_exp.so) against HEAD (libkdbhip_base.so), interleaved; fuzz on the new build
set -e
OUT=gpurun_out/r03j
mkdir -p $OUT
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=40 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
AB_STEPS=20 tools/ab_libs.sh --k 17 2>&1 | tee $OUT/ab_k17.txt
AB_STEPS=100 tools/ab_libs.sh --k 12 --expand 2>&1 | tee $OUT/ab_k12_expand.txt
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
