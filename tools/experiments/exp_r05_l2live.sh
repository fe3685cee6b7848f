#!/bin/bash
# Round 5: level 2 without the wait for its successor's pages in the request phase (liveness fix) against the build before (libkdbhip_base.so), and the touch on top
OUT=gpurun_out/l2live_ab.txt
: > $OUT
run() {  # lib, extra opts, label
  KDB_LIB=$PWD/kmerdb_amd/$1 python bench.py --k ${K:-15} --steps 64 --warmup 3 --no-cpu-baseline --no-extra-regions --no-configs $2 > gpurun_out/t.json 2> gpurun_out/t.err || { echo FAILED $3 >> $OUT; tail -3 gpurun_out/t.err >> $OUT; return; }
  python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('k=${K:-15} $3', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)" >> $OUT
}
for r in 1 2 3; do
  run libkdbhip_base.so "--opt l2_touch=0" base
  run libkdbhip.so "--opt l2_touch=0" live
  run libkdbhip.so "--opt l2_touch=1" live+touch
done
cat $OUT
