#!/bin/bash
# Round 5: level 2 (k <= 16) with the pages of the tile after next touched into the L2 (engine option l2_touch) against without, interleaved
# (the option was taken out again with the experiment -- commit d5ae6d6 has the code; profiles/r05/l2touch_ab.txt what it measured)
OUT=gpurun_out/l2touch_ab.txt
: > $OUT
for r in 1 2 3; do
  for V in 0 1; do
    python bench.py --k ${K:-15} --steps 64 --warmup 3 --no-cpu-baseline --no-extra-regions --no-configs --opt l2_touch=$V > gpurun_out/t.json 2> gpurun_out/t.err || { echo FAILED $V >> $OUT; tail -3 gpurun_out/t.err >> $OUT; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('k=${K:-15} l2_touch=$V', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)" >> $OUT
  done
done
cat $OUT
