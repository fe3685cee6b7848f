#!/bin/bash
# round 3: lines rotated within their page by the page number (libkdbhip_exp.so) against HEAD (libkdbhip_base.so);
# what hipMalloc costs by size; the speeds of level 1 at k = 15 over processes run back to back / with a pause
set -e
OUT=gpurun_out/r03k
mkdir -p $OUT
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_exp.so timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu > $OUT/fuzz.log 2>&1 || { tail -20 $OUT/fuzz.log; exit 1; }
tail -1 $OUT/fuzz.log
python tools/malloc_time.py 1 8 32 64 128 2>&1 | tee $OUT/malloc_time.txt
export AB_LIBS="libkdbhip_base.so libkdbhip_exp.so"
AB_STEPS=200 tools/ab_libs.sh --k 12 2>&1 | tee $OUT/ab_k12.txt
AB_STEPS=96 tools/ab_libs.sh --k 15 2>&1 | tee $OUT/ab_k15.txt
AB_STEPS=64 tools/ab_libs.sh --k 17 2>&1 | tee $OUT/ab_k17.txt
for pause in 0 0 0 10 10 10; do
  sleep $pause
  KDB_LIB=$PWD/kmerdb_amd/libkdbhip_base.so python bench.py --k 15 --steps 96 --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/m.json 2> $OUT/m.err
  python -c "
import json; d=json.load(open('$OUT/m.json')); print('pause $pause', d['ms_per_step'], {k: round(v,4) for k, v in d['roofline']['kernels_ms_per_step'].items()}, d['roofline'].get('arena'), flush=True)"
done 2>&1 | tee $OUT/modes_k15.txt
