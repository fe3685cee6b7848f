#!/usr/bin/env python3
"""Worst-case inputs for the LDS-histogram path: every lane hits the same bin / bucket. Prints ms per pass."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmerdb_amd
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--k", type=int, default=12); ap.add_argument("--reads", type=int, default=2_000_000)
args = ap.parse_args()
n, L, k = args.reads, 150, args.k
out = {}
def run(name, bases):
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    d_b = torch.from_numpy(bases).cuda(); d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    for algo in (1, 2):
        with kmerdb_amd.Engine(k, algo=algo) as eng:
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), n); eng.sync()
            t = time.perf_counter()
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), n); eng.sync()
            dt = time.perf_counter() - t
            _, total, uniq = eng.finish(copy=False)
            assert total == 2 * n * (L - k + 1)
            out[f"{name}_algo{algo}_ms"] = round(dt * 1e3, 2)
            out[f"{name}_unique"] = uniq
run("polyA", np.full(n * L, ord("A"), dtype=np.uint8))
run("AC_repeat", np.tile(np.frombuffer(b"AC", dtype=np.uint8), n * L // 2))
rng = np.random.Generator(np.random.PCG64(1))
unit = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5000)]
run("5kb_repeat", np.tile(unit, n * L // 5000 + 1)[: n * L].copy())
run("random", np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n * L)])
run("at_rich_70", np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n * L, p=[0.35, 0.15, 0.15, 0.35])])
# reads drawn from a real genome (E. coli K-12, a data file of the reference's tests), both strands
import gzip
g = b"".join(l.strip() for l in gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/ref_data/Ecoli_K12MG1655.fasta.gz")) if not l.startswith(b">"))
ga = np.frombuffer(g, dtype=np.uint8)
starts = rng.integers(0, ga.size - L, n)
run("ecoli_reads", ga[(starts[:, None] + np.arange(L)[None, :])].reshape(-1).copy())
print(json.dumps(out))
