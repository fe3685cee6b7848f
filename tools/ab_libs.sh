#!/bin/bash
# A/B of several builds on ONE device, interleaved:  AB_LIBS="libkdbhip_base.so libkdbhip.so" tools/ab_libs.sh [bench args...]
set -e
OUT=gpurun_out/ab_libs
mkdir -p $OUT
LIBS=${AB_LIBS:-"libkdbhip_base.so libkdbhip.so"}
COMMON="--no-cpu-baseline --no-extra-regions"
for r in 1 2 3; do
  for L in $LIBS; do
    KDB_LIB=$PWD/kmerdb_amd/$L python bench.py --steps ${AB_STEPS:-100} --warmup 3 $COMMON "$@" > $OUT/t.json 2> $OUT/t.err || { echo FAILED $L; tail -3 $OUT/t.err; exit 1; }
    python -c "
import json; d=json.load(open('$OUT/t.json')); print('$* $L', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)"
  done
done
