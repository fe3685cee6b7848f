#!/bin/bash
set -e
OUT=gpurun_out/r03d
mkdir -p $OUT
python tests/fuzz_gpu.py 60 1991 8,9,12,12,13,15,17 > $OUT/fuzz.txt 2>&1; tail -1 $OUT/fuzz.txt
for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L python bench.py --k 15 --steps 100 --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/k15_$L.json 2> $OUT/k15_$L.err || { echo FAILED; tail -3 $OUT/k15_$L.err; }
  python -c "
import json; d=json.load(open('$OUT/k15_$L.json')); print('k15 $L', d['ms_per_step'], d['roofline'].get('arena'), {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)"
done
tools/ab_valu.sh
