#!/bin/bash
# round 3: encode fast path for chunks of sixteen plain letters + the degeneracy test on the forward word (libkdbhip_exp.so) against HEAD
set -e
OUT=gpurun_out/r03q
mkdir -p $OUT
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -x -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=100 tools/ab_libs.sh --k 12 --expand 2>&1 | tee $OUT/ab_k12e.txt
AB_STEPS=96 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
