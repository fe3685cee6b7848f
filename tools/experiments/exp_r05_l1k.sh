#!/bin/bash
# Round 5: level 1 / level 2 of k = 15 with their shifts and masks compiled in (engine option l1_compiled_k) against the generic kernels, interleaved
OUT=gpurun_out/l1k_ab.txt
: > $OUT
for r in 1 2 3; do
  for V in 0 1; do
    python bench.py --k 15 --steps 64 --warmup 3 --no-cpu-baseline --no-extra-regions --no-configs --opt l1_compiled_k=$V > gpurun_out/t.json 2> gpurun_out/t.err || { echo FAILED $V >> $OUT; tail -3 gpurun_out/t.err >> $OUT; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('l1_compiled_k=$V', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)" >> $OUT
  done
done
cat $OUT
