#!/bin/bash
# Round 5: scatter of batch i + 1 beside the histogram pass of batch i (engine options overlap / overlap_hist_cus / overlap_mask_mode):
# parity on a small input, then ms per step of the k = 12 headline for a sweep of CU partitions.   -> gpurun_out/overlap_sweep.txt
set -e
export KDB_ALLOW_CU_MASKS=1      # (CU-masked streams are a diagnostic: include/kdbhip.h)
OUT=gpurun_out/overlap_sweep.txt
mkdir -p gpurun_out
: > $OUT
python - >> $OUT 2>&1 <<'PY'
import numpy as np, kmerdb_amd
from kmerdb_amd import synth
from oracle import kmer_oracle
kmer_oracle.build()
for k in (9, 12, 13):
    for opts in ((("overlap", 1),), (("overlap", 1), ("overlap_hist_cus", 64)), (("overlap", 1), ("overlap_hist_cus", 64), ("overlap_mask_mode", 1))):
        want = None
        with kmerdb_amd.Engine(k, canonicalize=True) as eng:
            for n, v in opts:
                eng.set_option(n, v)
            tot = 0
            for i in range(5):
                b, o = synth.reads(30000 + 1000 * i, 150 if i % 2 == 0 else 101, seed=100 + i)
                if i == 3:
                    b[:20000] = ord("A")              # degenerate ids: the side list
                eng.submit(b, o)
                w, t = kmer_oracle.c_count(b, o, k, True, kmer_oracle.N_DROP)
                want = w if want is None else want + w
                tot += t
            got, total, _ = eng.finish()
            print("parity k=%d %s:" % (k, opts), "OK" if total == tot and np.array_equal(got, want) else "MISMATCH", flush=True)
PY
run() { python bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline --no-extra-regions --no-configs "$@" > gpurun_out/t.json 2> gpurun_out/t.err || { echo "FAILED $*" >> $OUT; tail -3 gpurun_out/t.err >> $OUT; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('$*', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, flush=True)" >> $OUT; }
for r in 1 2; do
  run --k 12
  run --k 12 --opt overlap=1
  for H in 32 48 64 80 96; do
    for M in 0 1 2; do
      run --k 12 --opt overlap=1 --opt overlap_hist_cus=$H --opt overlap_mask_mode=$M
    done
  done
done
run --k 13
run --k 13 --opt overlap=1
run --k 13 --opt overlap=1 --opt overlap_hist_cus=64 --opt overlap_mask_mode=1
run --k 13 --opt overlap=1 --opt overlap_hist_cus=96 --opt overlap_mask_mode=1
cat $OUT
