#!/bin/bash
# A/B of two builds on ONE device: interleaved timing rounds, then one SQ instruction-count pass each.
#   tools/ab_valu.sh [bench args...]      compares kmerdb_amd/libkdbhip_base.so with kmerdb_amd/libkdbhip.so
set -e
OUT=gpurun_out/ab_valu
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra-regions"
for r in 1 2 3; do
  for L in libkdbhip_base.so libkdbhip.so; do
    KDB_LIB=$PWD/kmerdb_amd/$L python bench.py --steps 100 --warmup 3 $COMMON "$@" > $OUT/t.json 2> $OUT/t.err || { echo FAILED $L; tail -3 $OUT/t.err; exit 1; }
    python -c "
import json; d=json.load(open('$OUT/t.json')); print('$L', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)"
  done
done
for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_$L -- python3 bench.py --steps 3 --warmup 1 $COMMON "$@" > $OUT/p.json 2> $OUT/p.err || { echo "pmc failed $L"; tail -3 $OUT/p.err; }
  python3 tools/pmc_table.py $OUT/pmc_$L | grep "kernel\|scatter\|page_hist"
  rm -rf $OUT/pmc_$L
done
