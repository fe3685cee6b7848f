#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's
own hot-path code (kmerdb/kmer.py, kmerdb/parse.py, kmerdb/util.py,
kmerdb/config.py from /root/reference, unmodified, loaded by path).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Biopython is not installed here (ordinary ModuleNotFoundError, no network), so
tests/golden/bio_standin/ provides the few Bio names those files import; the
only arithmetic it contributes is Seq.reverse_complement().  The reference's
package __init__ (CLI, BGZF, jsonschema) is NOT executed: the four modules are
loaded individually under a bare ``kmerdb`` namespace package.

Outputs (committed; data only):
    inputs/*.fa *.fq *.fq.gz          small seeded inputs written by this script
    ref_data/*                        data files the reference's tests hold (copied verbatim)
    kmer_to_id.json                   kmer.kmer_to_id known answers
    shred.json                        kmer.shred known answers
    parsefile.json                    parse.parsefile metadata + sha256 + small full vectors
    vectors.npz                       full count vectors for the larger cases
    errors.json                       exception classes the reference raises on bad input
"""
import gzip
import hashlib
import importlib.util
import json
import os
import shutil
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
REFPKG = os.path.join(REF, "kmerdb")


def load_reference():
    sys.path.insert(0, os.path.join(HERE, "bio_standin"))
    pkg = types.ModuleType("kmerdb")
    pkg.__path__ = [REFPKG]            # namespace only: kmerdb/__init__.py is not executed
    sys.modules["kmerdb"] = pkg
    mods = {}
    for name in ("config", "util", "kmer", "parse"):
        spec = importlib.util.spec_from_file_location(f"kmerdb.{name}", os.path.join(REFPKG, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"kmerdb.{name}"] = m
        setattr(pkg, name, m)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def sha256_u64(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def write_inputs():
    d = os.path.join(HERE, "inputs")
    os.makedirs(d, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(20240612))
    L = np.array(list("ACGT"))

    def rand_seq(n, p_n=0.0):
        s = L[rng.integers(0, 4, size=n)]
        if p_n > 0:
            s = np.where(rng.random(n) < p_n, "N", s)
        return "".join(s)

    files = {}
    # the survey's tiny FASTQ: EXPAND -> 32, DROP -> 12 at k=5 forward
    recs = ["ACGTACGTACGT", "ACGTNACGTTTT", "ACGTA"]
    files["tiny.fq"] = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(recs))
    # 200 synthetic 150-bp reads, no N (BASELINE workload shape)
    recs = [rand_seq(150) for _ in range(200)]
    files["reads150.fq"] = "".join(f"@r{i} extra words\n{s}\n+\n{'I' * 150}\n" for i, s in enumerate(recs))
    # ragged reads with N's (some windows with 2-3 N), lengths 12..300
    recs = []
    for i in range(120):
        n = int(rng.integers(12, 300))
        recs.append(rand_seq(n, p_n=0.02 if i % 3 else 0.0))
    recs.append("N" * 14)
    recs.append("ACGTACGTACGTNN")
    recs.append("NNACGTACGTACGT")
    recs.append("A" * 40)
    recs.append("ACACACACACACACACACAC")
    files["ragged_n.fq"] = "".join(f"@q{i}\n{s}\n+\n{'#' * len(s)}\n" for i, s in enumerate(recs))
    # multi-record, multi-line FASTA with a long record (tiling across blocks)
    fa = []
    for i, n in enumerate([5000, 61, 12, 23456]):
        s = rand_seq(n, p_n=0.001 if i == 3 else 0.0)
        fa.append(f">contig{i} some description\n" + "\n".join(s[j:j + 70] for j in range(0, n, 70)) + "\n")
    files["contigs.fa"] = "".join(fa)
    for name, text in files.items():
        with open(os.path.join(d, name), "w") as f:
            f.write(text)
    # gzip twin of one of them (content sniff path, util.py:80-88)
    with gzip.GzipFile(os.path.join(d, "reads150.fq.gz"), "wb", mtime=0) as f:
        f.write(files["reads150.fq"].encode())
    # bad inputs
    bad = {
        "short_read.fq": "@a\nACGTACGTACGT\n+\nIIIIIIIIIIII\n@b\nACG\n+\nIII\n",
        "lowercase.fa": ">x\nACGTacgtACGTACGT\n",
        "iupac_r.fa": ">x\nACGTACGTRACGTACGT\n",
        "empty.fa": "",
    }
    for name, text in bad.items():
        with open(os.path.join(d, name), "w") as f:
            f.write(text)
    return d


def main():
    t0 = time.time()
    ref = load_reference()
    kmer, parse = ref["kmer"], ref["parse"]
    from Bio.SeqRecord import SeqRecord
    from Bio.Seq import Seq
    ind = write_inputs()

    # data files the reference's own tests hold (fixtures = data)
    rd = os.path.join(HERE, "ref_data")
    os.makedirs(rd, exist_ok=True)
    for f in ("Cacetobutylicum_ATCC824.fasta.gz", "test_Cac_ATCC824.8.kdb", "sample.fa"):
        shutil.copyfile(os.path.join(REF, "test", "data", f), os.path.join(rd, f))

    rng = np.random.Generator(np.random.PCG64(7))

    # ---- kmer_to_id -------------------------------------------------------
    dinucs = [a + b for a in "ACGT" for b in "ACGT"]
    k2i = {"dinuc_canonical": [kmer.kmer_to_id(s) for s in dinucs],
           "dinuc_forward": [kmer.kmer_to_id(s, canonicalize=False) for s in dinucs],
           "n_is_none": kmer.kmer_to_id("ATCNATC") is None,
           "random": []}
    for k in list(range(1, 18)) + [20, 25, 31]:
        for _ in range(6):
            s = "".join(np.array(list("ACGT"))[rng.integers(0, 4, size=k)])
            k2i["random"].append([s, kmer.kmer_to_id(s, canonicalize=True), kmer.kmer_to_id(s, canonicalize=False)])
    json.dump(k2i, open(os.path.join(HERE, "kmer_to_id.json"), "w"), indent=0)

    # ---- shred ------------------------------------------------------------
    cases = []
    seqs = ["ACGTAC", "ACNTACG", "ACGTTGCAAC", "NNNN", "ANNA", "ACGTNNACGT", "AAAAAAAAAA", "ACGTACGTACGTNACGTNNAC",
            "TTTTTTTT", "GATTACAGATTACA", "NACGT", "ACGTN"]
    for s in seqs:
        for k in (1, 2, 3, 4, 5):
            if len(s) < k:
                continue
            for rwn in (True, False):
                for canon in (True, False):
                    ids, sids, pos = kmer.shred(SeqRecord(Seq(s), id="s"), k, replace_with_none=rwn, canonicalize=canon)
                    cases.append({"seq": s, "k": k, "replace_with_none": rwn, "canonicalize": canon,
                                  "ids": [int(x) for x in ids], "pos": [int(p) for p in pos]})
    json.dump(cases, open(os.path.join(HERE, "shred.json"), "w"))

    # ---- parsefile --------------------------------------------------------
    pf = []
    vectors = {}

    def run(relpath, k, rwn, canon, keep_vector):
        path = os.path.join(HERE, relpath)
        cwd = os.getcwd()
        os.chdir(HERE)                     # so that metadata["filename"] is the relative path
        try:
            counts, meta, nullomers = parse.parsefile(relpath, k, replace_with_none=rwn, canonicalize=canon)
        finally:
            os.chdir(cwd)
        assert counts.dtype == np.uint64 and counts.shape == (4 ** k,)
        key = f"{os.path.basename(relpath)}|k{k}|rwn{int(rwn)}|canon{int(canon)}"
        entry = {"file": relpath, "k": k, "replace_with_none": rwn, "canonicalize": canon,
                 "metadata": {kk: (int(v) if isinstance(v, (int, np.integer)) else v) for kk, v in meta.items()},
                 "sha256_u64le": sha256_u64(counts), "sum": int(counts.sum()),
                 "nullomer_array_dtype": str(nullomers.dtype), "nullomer_array_len": int(len(nullomers)),
                 "nullomer_array_sha256": sha256_u64(nullomers), "key": key}
        if keep_vector:
            vectors[key] = counts
        pf.append(entry)
        print(f"  {key}: total={meta['total_kmers']} unique={meta['unique_kmers']}  ({time.time() - t0:.0f}s)", flush=True)

    for f in ("inputs/tiny.fq",):
        for k in (1, 3, 5):
            for rwn in (True, False):
                for canon in (True, False):
                    run(f, k, rwn, canon, True)
    for f in ("inputs/reads150.fq", "inputs/reads150.fq.gz"):
        for k, rwn, canon in ((8, True, True), (8, True, False), (11, False, True), (12, True, True), (12, True, False)):
            run(f, k, rwn, canon, k <= 11)
    for k in (4, 6, 8):
        for rwn in (True, False):
            for canon in (True, False):
                run("inputs/ragged_n.fq", k, rwn, canon, True)
    for k, rwn, canon in ((9, False, True), (9, True, False), (12, False, True)):
        run("inputs/contigs.fa", k, rwn, canon, k <= 9)
    for k in (8, 12):
        for canon in (True, False):
            run("ref_data/sample.fa", k, False, canon, k == 8)
    if "--skip-genome" not in sys.argv:
        run("ref_data/Cacetobutylicum_ATCC824.fasta.gz", 8, False, True, True)
        run("ref_data/Cacetobutylicum_ATCC824.fasta.gz", 8, False, False, True)
    json.dump(pf, open(os.path.join(HERE, "parsefile.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **vectors)

    # ---- errors -----------------------------------------------------------
    errs = []
    for f, k in (("inputs/short_read.fq", 8), ("inputs/lowercase.fa", 4), ("inputs/iupac_r.fa", 4),
                 ("inputs/empty.fa", 4), ("inputs/does_not_exist.fa", 4)):
        try:
            parse.parsefile(os.path.join(HERE, f), k)
            errs.append({"file": f, "k": k, "raises": None})
        except BaseException as e:  # noqa: BLE001 - we record whatever class the reference raises
            errs.append({"file": f, "k": k, "raises": type(e).__name__})
    for args, label in (((None, 4), "filepath None"), ((os.path.join(HERE, "inputs/tiny.fq"), "4"), "k str"),
                        ((os.path.join(HERE, "inputs/tiny.fq"), 4, 1), "replace_with_none int")):
        try:
            parse.parsefile(*args)
            errs.append({"call": label, "raises": None})
        except BaseException as e:  # noqa: BLE001
            errs.append({"call": label, "raises": type(e).__name__})
    json.dump(errs, open(os.path.join(HERE, "errors.json"), "w"), indent=1)
    print(f"done in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
