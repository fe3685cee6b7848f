#!/bin/bash
# round 3: k = 17 level 1 in two rounds per tile (libkdbhip_exp.so) against four (libkdbhip_base.so); what hipMalloc costs by size
set -e
OUT=gpurun_out/r03m
mkdir -p $OUT
python tools/malloc_time.py 1 8 32 64 128 > $OUT/malloc_time.txt 2>&1 || true
cat $OUT/malloc_time.txt
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=64 tools/ab_libs.sh --k 17 2>&1 | tee $OUT/ab_k17.txt
