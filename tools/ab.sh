#!/bin/bash
# A/B two builds in ONE run on ONE device (devices differ by several %): interleaved rounds
for r in 1 2 3; do
  for L in kmerdb_amd/libkdbhip_base.so kmerdb_amd/libkdbhip.so; do
    KDB_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$L'.split('/')[-1], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
  done
done
