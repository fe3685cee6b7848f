"""Diagnostic (GPU box): short ragged reads, small k, N-expansion mode against the oracle; prints the first differing bins.
Usage: KDB_LIB=... python tools/experiments/repro_smallk.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import kmerdb_amd
from oracle import kmer_oracle as oracle
LET = np.frombuffer(b"ACGTN", dtype=np.uint8)
nbad = 0
for k in (2, 1, 3, 4, 5, 8):
    for seed in range(12):
        rng = np.random.Generator(np.random.PCG64(1000 * k + seed))
        nreads = 5000
        lens = rng.integers(k, k + 5, size=nreads)
        p_n = 0.0005 if seed % 2 == 0 else 0.005
        total = int(lens.sum())
        bases = LET[rng.choice(5, size=total, p=[(1 - p_n) / 4] * 4 + [p_n])].copy()
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        for canon in (True, False):
            for old in (0, 1):
                with kmerdb_amd.Engine(k, canonicalize=canon, n_mode=1, algo=2) as eng:
                    if old:
                        eng.set_option("smallk_old", 1)
                    eng.submit(bases, offsets)
                    got, tot, uniq = eng.finish()
                want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_EXPAND)
                if not (tot == want_total and np.array_equal(got, want)):
                    nbad += 1
                    d = np.flatnonzero(got != want)
                    print("MISMATCH k=%d seed=%d canon=%d old=%d total %d vs %d; %d bins differ; first: %s" % (
                        k, seed, canon, old, tot, want_total, d.size, [(int(i), int(got[i]), int(want[i])) for i in d[:6]]), flush=True)
                    # where are the N's?
                    npos = np.flatnonzero(bases == ord("N"))
                    print("   N positions: %d; first %s; nbytes %d" % (npos.size, npos[:10].tolist(), total), flush=True)
print("done, mismatches:", nbad)
