#!/bin/bash
# Round 5: an engine option off / on, interleaved, three pairs:  OPT=hist_pipe K=12 STEPS=200 tools/experiments/exp_r05_opt_ab.sh
OPT=${OPT:-hist_pipe}
OUT=gpurun_out/${OPT}_ab.txt
for r in 1 2 3; do
  for V in 0 1; do
    python bench.py --k ${K:-12} --steps ${STEPS:-100} --warmup 3 --no-cpu-baseline --no-extra-regions --no-configs --opt $OPT=$V > gpurun_out/t.json 2> gpurun_out/t.err || { echo FAILED $V >> $OUT; tail -3 gpurun_out/t.err >> $OUT; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('k=${K:-12} $OPT=$V', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)" >> $OUT
  done
done
