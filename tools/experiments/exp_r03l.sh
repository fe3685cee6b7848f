#!/bin/bash
# round 3: level 1's half-lines of high bytes written two at a time as whole 64-byte lines (libkdbhip_exp.so) against HEAD
# (libkdbhip_base.so); what hipMalloc costs by size
set -e
OUT=gpurun_out/r03l
mkdir -p $OUT
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -x -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
PYTHONPATH=$PWD python tools/malloc_time.py 1 8 32 64 128 > $OUT/malloc_time.txt 2>&1 || true
cat $OUT/malloc_time.txt
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=96 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
AB_STEPS=96 tools/ab_libs.sh --k 13 2>&1 | tee $OUT/ab_k13.txt
AB_STEPS=96 tools/ab_libs.sh --k 16 2>&1 | tee $OUT/ab_k16.txt
for pause in 12 12; do
  for L in libkdbhip_base.so libkdbhip_exp.so; do
  sleep $pause
  KDB_LIB=$PWD/kmerdb_amd/$L python bench.py --k 15 --steps 96 --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/m.json 2> $OUT/m.err
  python -c "
import json; d=json.load(open('$OUT/m.json')); print('pause $pause $L', d['ms_per_step'], {k: round(v,4) for k, v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)"
  done
done 2>&1 | tee $OUT/modes_k15.txt
