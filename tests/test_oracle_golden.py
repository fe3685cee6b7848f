"""CPU: pin the oracle (oracle/kmer_oracle.{c,py}) against
  (a) the reference's own fixture pair  Cacetobutylicum_ATCC824.fasta.gz -> test_Cac_ATCC824.8.kdb
      (k=8, forward strand, every one of the 65,536 bins),
  (b) the known answers of the reference's test/test_kmer.py,
  (c) vectors produced by the reference's own kmer.py / parse.py (tests/golden/make_golden.py).
"""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def _load_records(oracle, path):
    recs = [s for _, s in oracle.read_records(path)]
    return oracle.pack_records(recs)


def read_kdb_counts(path):
    """Decompress a .kdb (concatenated gzip members), skip the YAML header, return (header_text, counts)."""
    with gzip.open(path, "rt") as f:
        text = f.read()
    header, body = text.split("\n" + "=" * 24 + "\n", 1)
    rows = np.array([line.split("\t") for line in body.strip().split("\n")])
    ids = rows[:, 1].astype(np.uint64)
    counts = np.zeros(len(ids), dtype=np.uint64)
    counts[ids.astype(np.int64)] = rows[:, 2].astype(np.uint64)
    return header, counts


# ---- (a) the reference's own fixture ------------------------------------------------------------

def test_c_oracle_reproduces_reference_kdb_fixture(oracle, golden_dir):
    header, expected = read_kdb_counts(os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb"))
    assert expected.size == 65536 and int(expected.sum()) == 4132866
    bases, offsets = _load_records(oracle, os.path.join(golden_dir, "ref_data", "Cacetobutylicum_ATCC824.fasta.gz"))
    counts, total = oracle.c_count(bases, offsets, 8, canonicalize=False, n_mode=oracle.N_EXPAND)
    assert total == 4132866
    assert np.array_equal(counts, expected)
    assert int(np.count_nonzero(counts)) == 64103          # header: unique_kmers
    assert "total_reads: 2" in header


# ---- (b) reference test/test_kmer.py ---------------------------------------------------------------

def test_kmer_to_id_known_answers(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "kmer_to_id.json")))
    dinucs = [a + b for a in "ACGT" for b in "ACGT"]
    # test_kmer.py:12-36 asserts 0..15, which is the forward encoding
    assert g["dinuc_forward"] == list(range(16))
    assert [oracle.py_kmer_to_id(s, canonicalize=False) for s in dinucs] == list(range(16))
    assert [oracle.py_kmer_to_id(s) for s in dinucs] == g["dinuc_canonical"]
    assert g["dinuc_canonical"] == [0, 1, 2, 3, 4, 5, 6, 2, 8, 9, 5, 1, 12, 8, 4, 0]
    assert oracle.py_kmer_to_id("ATCNATC") is None and g["n_is_none"] is True       # test_kmer.py:38-42
    for bad in (None, 1, 1.0, [1], {"hello": "world"}):                              # test_kmer.py:44-57
        with pytest.raises(TypeError):
            oracle.py_kmer_to_id(bad)
    import ctypes
    for s, canon, fwd in g["random"]:
        assert oracle.py_kmer_to_id(s, canonicalize=True) == canon
        assert oracle.py_kmer_to_id(s, canonicalize=False) == fwd
        arr = np.frombuffer(s.encode(), dtype=np.uint8).copy()
        for flag, want in ((1, canon), (0, fwd)):
            out = ctypes.c_uint64(0)
            rc = oracle.lib().kdbo_kmer_to_id(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(s), flag,
                                              ctypes.byref(out))
            assert rc == 0 and out.value == want
        assert oracle.py_id_to_kmer(fwd, len(s)) == s


# ---- (c) vectors generated from the reference's code ------------------------------------------------

def test_shred_matches_reference(oracle, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "shred.json")))
    assert len(cases) > 150
    for c in cases:
        ids, pos = oracle.py_shred(c["seq"], c["k"], replace_with_none=c["replace_with_none"],
                                   canonicalize=c["canonicalize"])
        # the reference enumerates the fills of an N window in set() order (kmer.py:559:
        # list(standard_lettersNA)), so within one position the ids are a multiset
        assert pos == c["pos"], c
        assert sorted(zip(pos, ids)) == sorted(zip(c["pos"], c["ids"])), c
        cids, cpos = oracle.c_shred(c["seq"], c["k"], canonicalize=c["canonicalize"],
                                    n_mode=oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND)
        assert sorted(zip(cpos.tolist(), cids.tolist())) == sorted(zip(c["pos"], c["ids"])), c


def _parsefile_cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "parsefile.json")))


def test_c_oracle_matches_reference_parsefile(oracle, golden_dir):
    vecs = np.load(os.path.join(golden_dir, "vectors.npz"))
    cases = _parsefile_cases(golden_dir)
    assert len(cases) >= 40
    seen_vec = 0
    for c in cases:
        bases, offsets = _load_records(oracle, os.path.join(golden_dir, c["file"]))
        counts, total = oracle.c_count(bases, offsets, c["k"], canonicalize=c["canonicalize"],
                                       n_mode=oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND)
        m = c["metadata"]
        assert total == m["total_kmers"] == c["sum"], c["key"]
        assert int(np.count_nonzero(counts)) == m["unique_kmers"], c["key"]
        assert _sha(counts) == c["sha256_u64le"], c["key"]
        assert len(offsets) - 1 == m["total_reads"]
        lens = np.diff(offsets.astype(np.int64))
        assert (int(lens.min()), int(lens.max()), int(lens.mean())) == (
            m["min_read_length"], m["max_read_length"], m["avg_read_length"])
        if c["key"] in vecs.files:
            assert np.array_equal(counts, vecs[c["key"]]), c["key"]
            seen_vec += 1
    assert seen_vec >= 30


def test_py_oracle_matches_reference_parsefile_small(oracle, golden_dir):
    vecs = np.load(os.path.join(golden_dir, "vectors.npz"))
    for c in _parsefile_cases(golden_dir):
        if not c["file"].endswith(("tiny.fq", "ragged_n.fq")) or c["k"] > 6:
            continue
        recs = [s for _, s in oracle.read_records(os.path.join(golden_dir, c["file"]))]
        counts, total = oracle.py_count(recs, c["k"], replace_with_none=c["replace_with_none"],
                                        canonicalize=c["canonicalize"])
        assert total == c["metadata"]["total_kmers"]
        assert np.array_equal(counts, vecs[c["key"]]), c["key"]


def test_survey_known_answers(golden_dir):
    """Hashes recorded in SURVEY.md 8(c) from the survey session agree with what make_golden.py produced."""
    by_key = {c["key"]: c for c in _parsefile_cases(golden_dir)}
    assert by_key["sample.fa|k8|rwn0|canon1"]["sha256_u64le"] == "71ba27c06c2ce31145b66f3ba5b11755f7197e8b217f8aedf56cd201e8abd491"
    assert by_key["sample.fa|k8|rwn0|canon0"]["sha256_u64le"] == "1ef946e95fe474bdf492ec757f1f1cbe6e0e5f873f08e0d92f15490e15376b58"
    assert by_key["sample.fa|k12|rwn0|canon1"]["sha256_u64le"] == "94a407025156675925920852fe6f18e00dc5a639e20c4f78d64d03e242b1ee67"
    assert by_key["sample.fa|k12|rwn0|canon0"]["sha256_u64le"] == "3e760475f54a4f30e98fe7bc6a1cee856b2c7ecdd25611823a2b3e1bf2ffcffe"
    assert by_key["Cacetobutylicum_ATCC824.fasta.gz|k8|rwn0|canon1"]["sha256_u64le"] == "83452914a5623d0e1a850a5f3334e9096e243754894f7ee0fe3814a1cfa8b050"
    assert by_key["tiny.fq|k5|rwn0|canon0"]["sum"] == 32 and by_key["tiny.fq|k5|rwn1|canon0"]["sum"] == 12


def test_oracle_errors(oracle, golden_dir):
    errs = json.load(open(os.path.join(golden_dir, "errors.json")))
    raised = {e.get("file"): e["raises"] for e in errs if "file" in e}
    # the reference raises (never skips) on all of these
    assert raised["inputs/short_read.fq"] and raised["inputs/lowercase.fa"] and raised["inputs/iupac_r.fa"]
    for f, k, status in (("inputs/short_read.fq", 8, oracle.SHORT_READ), ("inputs/lowercase.fa", 4, oracle.BAD_RESIDUE),
                         ("inputs/iupac_r.fa", 4, oracle.BAD_RESIDUE)):
        bases, offsets = _load_records(oracle, os.path.join(golden_dir, f))
        with pytest.raises(oracle.OracleError) as ei:
            oracle.c_count(bases, offsets, k)
        assert ei.value.status == status


def test_oracle_mt_equals_scalar(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    recs = ["".join(np.array(list("ACGTN"))[rng.choice(5, size=int(rng.integers(9, 200)), p=[.24, .24, .24, .24, .04])])
            for _ in range(300)]
    bases, offsets = oracle.pack_records(recs)
    for canon in (True, False):
        for mode in (oracle.N_DROP, oracle.N_EXPAND):
            a, ta = oracle.c_count(bases, offsets, 7, canon, mode)
            b, tb = oracle.c_count(bases, offsets, 7, canon, mode, nthreads=4)
            assert ta == tb and np.array_equal(a, b)


# the one shape of tests/golden/iupac_next_to_n.json the restatement (and the engine) does not reproduce: with
# replace_with_none=False the reference's _substitute_na_doublets gets a window through only if every code in it occurs at
# least twice (kmer.py:612 replaces "N" where it means the code) -- "ANRRNA" at k = 4, 5 returns counts there; here it raises
IUPAC_EXPAND_EXCEPTIONS = {("ANRRNA", 4), ("ANRRNA", 5)}


def test_iupac_codes_next_to_n_follow_the_reference(oracle, golden_dir):
    """kmer_to_id returns None for a window that holds an N before it meets another IUPAC code (kmer.py:287-289): a record whose
    codes are all shielded by N's is accepted with replace_with_none=True (its N-windows dropped) and refused otherwise.  The
    vectors were produced by the reference's own kmer.shred / parse.parsefile (tests/golden/make_golden_iupac.py)."""
    g = json.load(open(os.path.join(golden_dir, "iupac_next_to_n.json")))
    n_ret = n_raise = 0
    for c in g["shred"]:
        if not c["replace_with_none"] and (c["seq"], c["k"]) in IUPAC_EXPAND_EXCEPTIONS:
            assert c["raises"] is None                    # (the reference returns; the documented divergence)
            continue
        mode = oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND
        for impl in ("py", "c"):
            try:
                if impl == "py":
                    ids, pos = oracle.py_shred(c["seq"], c["k"], replace_with_none=c["replace_with_none"], canonicalize=c["canonicalize"])
                else:
                    ids, pos = oracle.c_shred(c["seq"], c["k"], c["canonicalize"], mode)
                got = ([int(x) for x in ids], [int(x) for x in pos])
            except ValueError:
                got = None
            if c["raises"]:
                assert got is None, (impl, c["seq"], c["k"], c["replace_with_none"])
                n_raise += 1
            else:
                assert got == (c["ids"], c["pos"]), (impl, c["seq"], c["k"], c["replace_with_none"])
                n_ret += 1
    assert n_ret >= 150 and n_raise >= 150
    for c in g["parsefile"]:
        recs = [s for _, s in oracle.read_records(os.path.join(golden_dir, c["file"]))]
        bases, offsets = oracle.pack_records(recs)
        mode = oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND
        if c["raises"]:
            with pytest.raises(ValueError):
                oracle.c_count(bases, offsets, c["k"], c["canonicalize"], mode)
        else:
            want, total = oracle.c_count(bases, offsets, c["k"], c["canonicalize"], mode)
            assert [int(x) for x in want] == c["counts"] and total == c["metadata"]["total_kmers"]


# ---- (d) k = 13..17: the reference's kmer.shred / parse.parsefile (tests/golden/make_golden_largek.py) ----------------

def _largek(golden_dir):
    with gzip.open(os.path.join(golden_dir, "largek.json.gz"), "rt") as f:
        return json.load(f)


def test_oracle_equals_the_reference_at_k13_to_17(oracle, golden_dir):
    """Sparse count vectors of the reference's own shred at k = 13..17, both N modes, both strand modes (the dense-vector
    fixtures stop at k = 12: a 4^17 vector is 128 GiB)."""
    g = _largek(golden_dir)
    recs = g["records"]
    assert len(g["cases"]) == 5 * 4
    for c in g["cases"]:
        k, canon = c["k"], c["canonicalize"]
        mode = oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND
        ids = np.concatenate([oracle.c_shred(r, k, canon, mode)[0] for r in recs if len(r) >= k])
        uniq, cnt = np.unique(ids, return_counts=True)
        assert ids.size == c["total_kmers"], (k, canon, mode)
        assert [int(x) for x in uniq] == c["ids"] and [int(x) for x in cnt] == c["counts"], (k, canon, mode)


def test_oracle_equals_the_reference_parsefile_at_k13(oracle, golden_dir):
    g = _largek(golden_dir)
    for c in g["parsefile_k13"]:
        bases, offsets = _load_records(oracle, os.path.join(golden_dir, c["file"]))
        mode = oracle.N_DROP if c["replace_with_none"] else oracle.N_EXPAND
        counts, total = oracle.c_count(bases, offsets, 13, canonicalize=c["canonicalize"], n_mode=mode)
        assert total == c["metadata"]["total_kmers"] == c["sum"]
        nz = np.flatnonzero(counts)
        assert [int(i) for i in nz] == c["ids"] and [int(x) for x in counts[nz]] == c["counts"], c["file"]
        assert _sha(counts) == c["sha256_u64le"], c["file"]
        assert int(nz.size) == c["metadata"]["unique_kmers"]


# ---- (e) canonical mode without the stand-in ------------------------------------------------------------------------

def test_canonical_vector_is_the_fold_of_the_references_forward_fixture(oracle, golden_dir):
    """Canonical counts are pinned through vectors the reference's code produced with OUR stand-in for Bio.Seq.reverse_complement
    (tests/golden/bio_standin).  Independent of it: kmer.py:307-315 counts a window under min(id, id of its reverse complement), so the
    canonical vector of a genome is its forward vector folded along i <-> rc(i) -- and the forward vector here is the reference's OWN fixture
    (test_Cac_ATCC824.8.kdb, made by the reference with the real Biopython).  The fold uses nothing but the 2-bit map of kmer.py:44-49
    (A0 C1 G2 T3: the complement of code c is 3 - c).  The genome holds no N, so no window is dropped or expanded."""
    k = 8
    _, forward = read_kdb_counts(os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb"))
    ids = np.arange(4 ** k, dtype=np.uint64)
    rc = np.zeros_like(ids)
    for j in range(k):                                   # base j (from the left) of the k-mer becomes base k - 1 - j, complemented
        code = (ids >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)
        rc |= (np.uint64(3) - code) << np.uint64(2 * j)
    canon_id = np.minimum(ids, rc)
    want = np.zeros(4 ** k, dtype=np.uint64)
    np.add.at(want, canon_id.astype(np.int64), forward)
    bases, offsets = _load_records(oracle, os.path.join(golden_dir, "ref_data", "Cacetobutylicum_ATCC824.fasta.gz"))
    assert not np.any(bases == ord("N"))
    got, total = oracle.c_count(bases, offsets, k, canonicalize=True, n_mode=oracle.N_EXPAND)
    assert total == int(forward.sum()) and np.array_equal(got, want)
    # ... and the vector the reference's code produced through the stand-in is that fold too
    vecs = np.load(os.path.join(golden_dir, "vectors.npz"))
    key = "Cacetobutylicum_ATCC824.fasta.gz|k8|rwn0|canon1"
    assert key in vecs.files and np.array_equal(vecs[key], want)
