#!/bin/bash
# round 3, final build: every k of the LDS-histogram paths, 10 M x 150 bp per step
set -e
OUT=gpurun_out/r03u
mkdir -p $OUT
for k in 8 9 10 11 12 13 14 15 16 17; do
  steps=96; [ $k = 17 ] && steps=128
  python bench.py --k $k --steps $steps --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/bench_k$k.json 2> $OUT/bench_k$k.err || { echo FAILED k=$k; tail -5 $OUT/bench_k$k.err; continue; }
  python -c "
import json; d=json.load(open('$OUT/bench_k$k.json')); r=d['roofline']; print('k=$k', d['ms_per_step'], d['gbase_per_s'], {k: round(v,3) for k, v in r['kernels_ms_per_step'].items()}, {n: round(v['hbm_frac'],3) for n, v in r['per_kernel'].items() if 'hbm_frac' in v}, flush=True)"
done
for mode in --forward --expand; do
  python bench.py --k 12 --steps 200 --warmup 3 --no-cpu-baseline --no-extra-regions $mode > $OUT/bench_k12$mode.json 2> $OUT/t.err
  python -c "
import json; d=json.load(open('$OUT/bench_k12$mode.json')); r=d['roofline']; print('k=12 $mode', d['ms_per_step'], d['gbase_per_s'], {k: round(v,3) for k, v in r['kernels_ms_per_step'].items()}, flush=True)"
done
