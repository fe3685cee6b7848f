#!/bin/bash
# round 3: scatter_bases_kernel compiled for k = 12 (libkdbhip_exp.so) against HEAD
set -e
OUT=gpurun_out/r03s
mkdir -p $OUT
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=100 tools/ab_libs.sh --k 12 --expand 2>&1 | tee $OUT/ab_k12e.txt
AB_STEPS=100 tools/ab_libs.sh --k 12 --forward 2>&1 | tee $OUT/ab_k12f.txt
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
