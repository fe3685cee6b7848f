#!/usr/bin/env python3
"""Golden vectors for k = 13..17 (the reference-generated fixtures stopped at k = 12): what the REFERENCE's own
kmer.shred (loaded unmodified from /root/reference, as in make_golden.py) emits for seeded records with N's, in both N modes and
both strand modes -- as sparse count vectors (parse.py:117-137: counts[id] += 1 for every id that is not None), since a dense
4^17 vector is 128 GiB -- and parse.parsefile's metadata and dense-vector sha256 at k = 13 (512 MiB: the largest k the reference's
own accumulation loop finishes in minutes here).

    python tests/golden/make_golden_largek.py        (build container only: needs /root/reference)

Output: largek.json.gz, inputs/largek_n.fq (data only).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, sha256_u64  # noqa: E402


def records():
    rng = np.random.Generator(np.random.PCG64(1317))
    L = np.array(list("ACGT"))

    def rand_seq(n, p_n=0.0):
        s = L[rng.integers(0, 4, size=n)]
        if p_n > 0:
            s = np.where(rng.random(n) < p_n, "N", s)
        return "".join(s)

    recs = [rand_seq(int(rng.integers(17, 220)), 0.0 if i % 2 else 0.01) for i in range(60)]
    recs += ["A" * 50, "ACGT" * 12, "ACGTTGCA" * 3 + "N" + "TGCATGCA" * 3, "ACGTTGCA" * 3 + "NN" + "TGCATGCA" * 3,
             "ACGTTGCAAC" * 2 + "NACGTGN" + "TGCATGCATG" * 2, "T" * 17, "G" * 16 + "N", "N" + "C" * 16,
             "ACGTACGTACGTACGTAN" + "ACGTACGTACGTACGTA"]
    return recs


def main():
    ref = load_reference()
    kmer, parse = ref["kmer"], ref["parse"]
    from Bio.SeqRecord import SeqRecord
    from Bio.Seq import Seq
    recs = records()
    out = {"records": recs, "cases": []}
    for k in (13, 14, 15, 16, 17):
        for rwn in (True, False):
            for canon in (True, False):
                counts = {}
                total = 0
                for s in recs:
                    if len(s) < k:
                        continue
                    ids, _, _ = kmer.shred(SeqRecord(Seq(s), id="s"), k, replace_with_none=rwn, canonicalize=canon)
                    for i in ids:                      # parse.py:133-136
                        if i is not None:
                            counts[int(i)] = counts.get(int(i), 0) + 1
                            total += 1
                keys = sorted(counts)
                out["cases"].append({"k": k, "replace_with_none": rwn, "canonicalize": canon, "total_kmers": total,
                                     "ids": keys, "counts": [counts[i] for i in keys]})
                print("k=%d rwn=%d canon=%d: %d k-mers, %d unique" % (k, rwn, canon, total, len(keys)), flush=True)
    # parse.parsefile itself at k = 13 on a committed input (dense 4^13 vector: sha256 + sparse form)
    pf = []
    with open(os.path.join(HERE, "inputs", "largek_n.fq"), "w") as f:          # the records above as a FASTQ file (committed input)
        for i, s in enumerate(recs):
            f.write("@lk%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        # (ragged_n.fq holds an all-N record: 4^13 fills in expansion mode -- drop mode only; contigs.fa a 12-base record: the reference raises)
        for rel, rwn, canon in (("inputs/ragged_n.fq", True, True), ("inputs/largek_n.fq", False, True), ("inputs/largek_n.fq", False, False),
                                ("inputs/reads150.fq.gz", True, True)):
            counts, meta, nullomers = parse.parsefile(rel, 13, replace_with_none=rwn, canonicalize=canon)
            nz = np.flatnonzero(counts)
            pf.append({"file": rel, "k": 13, "replace_with_none": rwn, "canonicalize": canon,
                       "metadata": {kk: (int(v) if isinstance(v, (int, np.integer)) else v) for kk, v in meta.items()},
                       "sha256_u64le": sha256_u64(counts), "sum": int(counts.sum()),
                       "ids": [int(i) for i in nz], "counts": [int(c) for c in counts[nz]],
                       "nullomer_array_len": int(len(nullomers))})
            print("parsefile %s rwn=%d canon=%d: total=%d unique=%d" % (rel, rwn, canon, meta["total_kmers"], meta["unique_kmers"]), flush=True)
    finally:
        os.chdir(cwd)
    out["parsefile_k13"] = pf
    import gzip
    with gzip.GzipFile(os.path.join(HERE, "largek.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())


if __name__ == "__main__":
    main()
