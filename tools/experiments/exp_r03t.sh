#!/bin/bash
# round 3: the request's answer arrives in the register that held the word offset (no clearing of the sixteen answers) (libkdbhip_exp.so) against HEAD
set -e
OUT=gpurun_out/r03t
mkdir -p $OUT
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=96 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
AB_STEPS=64 tools/ab_libs.sh --k 17 2>&1 | tee $OUT/ab_k17.txt
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -x -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
