"""GPU (-m gpu): the HIP engine, called through the C ABI, against the oracle and the golden vectors.
Integer work: every comparison is bit-exact (np.array_equal on uint64)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALGOS = (1, 2)          # 1 = direct global atomics, 2 = LDS-histogram paths (k <= 7 in LDS, else paged scatter in one or two levels)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def _count(Engine, bases, offsets, k, canon, n_mode, algo, **opts):
    with Engine(k, canonicalize=canon, n_mode=n_mode, algo=algo) as eng:
        for name, v in opts.items():
            eng.set_option(name, v)
        eng.submit(bases, offsets)
        return eng.finish()


def test_golden_parsefile_all_cases(gpu_engine_cls, golden_dir):
    """kmerdb_amd.parse.parsefile == the reference's parse.parsefile on every golden case
    (includes the reference's own Cac genome fixture, k=8 forward)."""
    from kmerdb_amd import parse
    vecs = np.load(os.path.join(golden_dir, "vectors.npz"))
    cases = json.load(open(os.path.join(golden_dir, "parsefile.json")))
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        for c in cases:
            counts, meta, nullomers = parse.parsefile(c["file"], c["k"], replace_with_none=c["replace_with_none"],
                                                      canonicalize=c["canonicalize"])
            assert counts.dtype == np.uint64 and counts.shape == (4 ** c["k"],)
            assert meta == c["metadata"], c["key"]
            assert _sha(counts) == c["sha256_u64le"], c["key"]
            assert nullomers.dtype == np.uint64 and len(nullomers) == c["nullomer_array_len"]
            assert _sha(nullomers) == c["nullomer_array_sha256"], c["key"]
            if c["key"] in vecs.files:
                assert np.array_equal(counts, vecs[c["key"]]), c["key"]
    finally:
        os.chdir(cwd)


def test_golden_k13_to_17_equal_the_reference(gpu_engine_cls, golden_dir):
    """k = 13..17 against vectors the reference's own kmer.shred / parse.parsefile produced (tests/golden/make_golden_largek.py):
    sparse count vectors in both N modes and both strand modes through the engine (one- and two-level scatter paths, 34-bit ids),
    and parse.parsefile at k = 13 (metadata, dense-vector sha256)."""
    import gzip
    from kmerdb_amd import parse
    from oracle import kmer_oracle
    with gzip.open(os.path.join(golden_dir, "largek.json.gz"), "rt") as f:
        g = json.load(f)
    recs = g["records"]
    for c in g["cases"]:
        k = c["k"]
        if k >= 16 and c["replace_with_none"] == c["canonicalize"]:
            continue                      # (a 4^16 / 4^17 vector takes seconds to allocate: two of the four mode pairs there)
        bases, offsets = kmer_oracle.pack_records([r for r in recs if len(r) >= k])      # (packing only: no oracle arithmetic)
        uniq = np.array(c["ids"], dtype=np.uint64)
        with gpu_engine_cls(k, canonicalize=c["canonicalize"], n_mode=0 if c["replace_with_none"] else 1) as eng:
            eng.submit(bases, offsets)
            _, total, unique = eng.finish(copy=False)
            got = _sparse_got(eng, uniq)
        assert total == c["total_kmers"] and unique == uniq.size, (k, c["replace_with_none"], c["canonicalize"])
        assert np.array_equal(got, np.array(c["counts"], dtype=np.uint64)), (k, c["replace_with_none"], c["canonicalize"])
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        for c in g["parsefile_k13"]:
            counts, meta, nullomers = parse.parsefile(c["file"], 13, replace_with_none=c["replace_with_none"], canonicalize=c["canonicalize"])
            assert meta == c["metadata"], c["file"]
            assert _sha(counts) == c["sha256_u64le"], c["file"]
            assert len(nullomers) == c["nullomer_array_len"]
    finally:
        os.chdir(cwd)


def test_reference_kdb_fixture(gpu_engine_cls, golden_dir):
    """The reference's own known answer: Cac genome -> test_Cac_ATCC824.8.kdb (k=8, forward)."""
    from kmerdb_amd import parse
    from tests.test_oracle_golden import read_kdb_counts
    _, expected = read_kdb_counts(os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb"))
    counts, meta, _ = parse.parsefile(os.path.join(golden_dir, "ref_data", "Cacetobutylicum_ATCC824.fasta.gz"), 8,
                                      replace_with_none=False, canonicalize=False)
    assert np.array_equal(counts, expected)
    assert meta["total_kmers"] == 4132866 and meta["unique_kmers"] == 64103 and meta["nullomers"] == 1433
    assert meta["md5"] == "0a0f73e1c8b8285703e29279bafaabef"
    assert meta["sha256"] == "f9081291b62ff3387f1ca6ee2484669c849ed1840fdf2dd9dc3a0c93e9e87951"


def _sparse_expect(oracle, recs, k, canon, omode):
    """(unique ids, their counts, total) of the ids the oracle's shred emits for `recs` (N expansion included)."""
    ids = np.concatenate([oracle.c_shred(r, k, canon, omode)[0] for r in recs])
    uniq, cnt = np.unique(ids, return_counts=True)
    return uniq, cnt.astype(np.uint64), ids.size


def _sparse_got(eng, uniq):
    import torch
    t = eng.table_tensor()
    return t[torch.as_tensor(uniq.astype(np.int64), device=t.device)].cpu().numpy().astype(np.uint64)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("k", [1, 2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17])
def test_random_reads_vs_oracle(gpu_engine_cls, oracle, k, algo):
    """Seeded ragged reads with N's, both strand modes, both N modes (N expansion is the reference CLI's default:
    kmerdb/__init__.py:1889, kmer.py:541-565), vs the C oracle.  k >= 14: sparse compare, with the deferred histogram
    pass both on and off; the EXPAND records hold isolated N's (expanded in place) and one window with three N's
    (queued for expand_worklist_kernel), so both routes run through the two-level scatter kernels."""
    rng = np.random.Generator(np.random.PCG64(1000 + k))
    letters = np.array(list("ACGTN"))
    recs = []
    nrec = 400 if k <= 13 else 250
    for i in range(nrec):
        n = int(rng.integers(k, 400))
        p_n = 0.0 if i % 2 else (0.01 if k <= 13 else 0.002)
        recs.append("".join(letters[rng.choice(5, size=n, p=[(1 - p_n) / 4] * 4 + [p_n])]))
    recs += ["A" * (k + 30), "ACGT" * 20, "N" * min(k, 6 if k <= 13 else 3) + "ACGT" * 8, "T" * k,
             "ACGTTGCA" * 4 + "NNN" + "TGCATGCA" * 4]
    recs = [r for r in recs if len(r) >= k]
    bases, offsets = oracle.pack_records(recs)
    for canon in (True, False):
        for omode, gmode in ((oracle.N_DROP, 0), (oracle.N_EXPAND, 1)):
            if k == 17 and (canon, gmode) in ((False, 0),) + (((True, 1),) if algo == 1 else ()):
                continue       # (k = 17: allocating and clearing a 128 GiB vector takes 4 s per engine -- four engines on the scatter paths, two on direct atomics; the fuzz cases and config 4's test cover the rest)
            if k >= 14:
                # a 4^15 / 4^16 uint64 host vector is 8 / 32 GiB: compare through the sparse ids instead
                uniq, cnt, n_ids = _sparse_expect(oracle, recs, k, canon, omode)
                for defer in ((1, 0) if algo == 2 and (k < 17 or (canon and gmode == 0)) else (1,)):       # (k = 17: the undeferred pass once)
                    with gpu_engine_cls(k, canonicalize=canon, n_mode=gmode, algo=algo) as eng:
                        eng.set_option("defer_flush", defer)
                        eng.submit(bases, offsets)
                        _, total, unique = eng.finish(copy=False)
                        got = _sparse_got(eng, uniq)
                    assert total == n_ids and unique == uniq.size, (k, canon, omode, algo, defer)
                    assert np.array_equal(got, cnt), (k, canon, omode, algo, defer)
                continue
            want, want_total = oracle.c_count(bases, offsets, k, canon, omode)
            got, total, unique = _count(gpu_engine_cls, bases, offsets, k, canon, gmode, algo)
            assert total == want_total
            assert unique == int(np.count_nonzero(want))
            assert np.array_equal(got, want), (k, canon, omode, algo)


@pytest.mark.parametrize("algo", ALGOS)
def test_tile_and_buffer_boundaries(gpu_engine_cls, oracle, algo):
    """Records that straddle 16 KiB tiles, staging buffers (k-1 overlap tiling) and odd buffer ends."""
    rng = np.random.Generator(np.random.PCG64(77))
    L = np.array(list("ACGT"))
    k = 11
    lens = [16384 - 5, 11, 16384 * 2 + 3, 12, 40000, 16384, 11, 1, ]
    recs = ["".join(L[rng.integers(0, 4, size=n)]) for n in lens if n >= k]
    bases, offsets = oracle.pack_records(recs)
    want, want_total = oracle.c_count(bases, offsets, k, True, oracle.N_DROP)
    got, total, _ = _count(gpu_engine_cls, bases, offsets, k, True, 0, algo)
    assert total == want_total and np.array_equal(got, want)
    # tiny staging buffers force long records to be tiled across kdb_submit's double buffers
    got, total, _ = _count(gpu_engine_cls, bases, offsets, k, True, 0, algo, stage_bytes=4096, stage_reads=3)
    assert total == want_total and np.array_equal(got, want)


@pytest.mark.parametrize("algo", ALGOS)
def test_accumulates_over_submits_and_reset(gpu_engine_cls, oracle, algo):
    from kmerdb_amd import synth
    k = 9
    b1, o1 = synth.reads(3000, 150, seed=1)
    b2, o2 = synth.reads(2000, 100, seed=2)
    w1, _ = oracle.c_count(b1, o1, k, True, 0)
    w2, _ = oracle.c_count(b2, o2, k, True, 0)
    with gpu_engine_cls(k, algo=algo) as eng:
        eng.submit(b1, o1)
        eng.submit(b2, o2)
        got, total, _ = eng.finish()
        assert np.array_equal(got, w1 + w2) and total == int((w1 + w2).sum())
        eng.reset()
        eng.submit(b2, o2)
        got, _, _ = eng.finish()
        assert np.array_equal(got, w2)


@pytest.mark.parametrize("algo", ALGOS)
def test_adversarial_single_bin_and_low_complexity(gpu_engine_cls, oracle, algo):
    """All atomics of all XCDs on one bin (poly-A), and period-2 repeats: no increment may be lost."""
    k = 12
    recs = ["A" * 150] * 20000 + ["AC" * 75] * 5000 + ["T" * 150] * 1000
    bases, offsets = oracle.pack_records(recs)
    want, want_total = oracle.c_count(bases, offsets, k, True, 0)
    got, total, unique = _count(gpu_engine_cls, bases, offsets, k, True, 0, algo)
    assert total == want_total == 26000 * 139
    assert got[0] == 21000 * 139 and np.array_equal(got, want) and unique == int(np.count_nonzero(want))


def test_errors_raise_not_skip(gpu_engine_cls, golden_dir):
    from kmerdb_amd import parse
    for f, k in (("inputs/short_read.fq", 8), ("inputs/lowercase.fa", 4), ("inputs/iupac_r.fa", 4), ("inputs/empty.fa", 4)):
        with pytest.raises(ValueError):
            parse.parsefile(os.path.join(golden_dir, f), k)
    with pytest.raises(ValueError):
        gpu_engine_cls(18)
    with pytest.raises(ValueError):
        gpu_engine_cls(0)


def test_shred_matches_reference(gpu_engine_cls, golden_dir):
    from kmerdb_amd import kmer
    cases = json.load(open(os.path.join(golden_dir, "shred.json")))
    for c in cases:
        ids, sids, pos = kmer.shred(c["seq"], c["k"], replace_with_none=c["replace_with_none"], canonicalize=c["canonicalize"])
        assert pos == c["pos"], c
        assert sorted(zip(pos, ids)) == sorted(zip(c["pos"], c["ids"])), c
        assert sids == ["Untitled_sequence"] * len(ids)
    with pytest.raises(ValueError):
        kmer.shred("ACG", 5)


def test_device_resident_submit_and_table_tensor(gpu_engine_cls, oracle):
    import torch
    from kmerdb_amd import synth
    k = 12
    bases, offsets = synth.reads(20000, 150, seed=3)
    want, want_total = oracle.c_count(bases, offsets, k, True, 0)
    d_b = torch.from_numpy(bases).cuda()
    d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    table = torch.zeros(4 ** k, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for algo in ALGOS:
        with gpu_engine_cls(k, table_ptr=table.data_ptr(), algo=algo) as eng:
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)   # marks are idempotent
            _, total, _ = eng.finish(copy=False)
            assert total == 2 * want_total
            assert np.array_equal(table.cpu().numpy().view(np.uint64), 2 * want)
            assert np.array_equal(eng.table_tensor().cpu().numpy().view(np.uint64), 2 * want)
    assert np.array_equal(d_b.cpu().numpy(), bases)              # the buffer is what it was (marks come off after every batch)


def test_full_size_properties_k12(gpu_engine_cls, oracle):
    """BASELINE config 2 shape at a size the oracle cannot finish quickly: size-independent properties.
    Sum(counts) == n_reads*(151-k); forward and canonical vectors are related by folding; a
    1/16 prefix equals the oracle exactly."""
    from kmerdb_amd import synth
    k, n = 12, 1_000_000
    bases, offsets = synth.reads(n, 150, seed=synth.SEED0 + 2)
    res = {}
    for canon in (True, False):
        for algo in ALGOS:
            got, total, unique = _count(gpu_engine_cls, bases, offsets, k, canon, 0, algo)
            assert total == n * (151 - k) == int(got.sum())
            res[(canon, algo)] = got
        assert np.array_equal(res[(canon, 1)], res[(canon, 2)])
    # canonical = forward folded onto min(id, rc(id))
    ids = np.arange(4 ** k, dtype=np.uint64)
    rc = np.zeros_like(ids)
    x = ids.copy()
    for _ in range(k):
        rc = (rc << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    folded = np.zeros(4 ** k, dtype=np.uint64)
    np.add.at(folded, np.minimum(ids, rc).astype(np.int64), res[(False, 1)])
    assert np.array_equal(folded, res[(True, 1)])
    m = n // 16
    want, _ = oracle.c_count(bases[:m * 150], offsets[:m + 1], k, True, 0, nthreads=8)
    got, _, _ = _count(gpu_engine_cls, bases[:m * 150], offsets[:m + 1], k, True, 0, 2)
    assert np.array_equal(got, want)


def test_pinned_submit_and_threaded_staging(gpu_engine_cls, oracle):
    import kmerdb_amd
    from kmerdb_amd import synth
    k = 10
    bases, offsets = synth.reads(300000, 101, seed=9)          # 30 MB: several staging buffers at 8 MiB
    want, want_total = oracle.c_count(bases, offsets, k, True, 0, nthreads=8)
    pb = kmerdb_amd.pinned_empty(bases.size)
    pb[:] = bases
    with gpu_engine_cls(k) as eng:
        eng.set_option("stage_bytes", 8 << 20)
        eng.submit_pinned(pb, offsets)
        got, total, _ = eng.finish()
        assert total == want_total and np.array_equal(got, want)
        eng.reset()
        eng.set_option("copy_threads", 3)
        eng.submit(bases, offsets)
        got, total, _ = eng.finish()
        assert total == want_total and np.array_equal(got, want)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("L,k", [(5, 3), (16, 11), (17, 12), (33, 9), (150, 12), (4097, 8)])
def test_uniform_length_batches_use_computed_record_starts(gpu_engine_cls, oracle, L, k, algo):
    """All records the same length: no start marks are written, boundaries are computed (incl. L < 16, L > tile stride)."""
    rng = np.random.Generator(np.random.PCG64(L * 100 + k))
    n = 70000 // L + 3
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, size=n * L, p=[.2475, .2475, .2475, .2475, .01])]
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    for canon in (True, False):
        for omode, gmode in ((oracle.N_DROP, 0), (oracle.N_EXPAND, 1)):
            want, want_total = oracle.c_count(bases, offsets, k, canon, omode)
            got, total, _ = _count(gpu_engine_cls, bases, offsets, k, canon, gmode, algo)
            assert total == want_total and np.array_equal(got, want), (L, k, canon, omode)


def test_device_submit_rejects_offsets_that_do_not_tile_the_buffer(gpu_engine_cls):
    import torch
    from kmerdb_amd import synth
    bases, offsets = synth.reads(100, 50, seed=4)
    d_b = torch.from_numpy(bases).cuda()
    wild = offsets.copy()
    wild[40] = np.uint64(1 << 40)                   # an interior offset far outside the buffer (a start mark there would fault)
    swapped = offsets.copy()
    swapped[[10, 11]] = swapped[[11, 10]]           # not monotone
    for bad in (offsets + np.uint64(1), offsets[:-1], wild, swapped):
        d_o = torch.from_numpy(bad.view(np.int64).copy()).cuda()
        with gpu_engine_cls(8) as eng:
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(bad) - 1)
            with pytest.raises(ValueError):
                eng.sync()


@pytest.mark.parametrize("k", [13, 14, 15, 16])          # (k = 17 -- a 128 GiB vector per engine -- runs this shape in the fuzz cases and in config 4's test)
def test_two_level_on_skewed_and_tiled_input(gpu_engine_cls, oracle, k):
    """Two-level path: one dominant L1 bucket (poly-A), records straddling tiles and halves, N expansion at every k."""
    rng = np.random.Generator(np.random.PCG64(k))
    L = np.array(list("ACGT"))
    recs = ["A" * 300] * 400 + ["".join(L[rng.integers(0, 4, size=n)]) for n in (8190, 8195, 16390, 40000, 33, k, k + 1)]
    recs += ["ACGT" * 50 + "N" + "ACGT" * 10, "AC" * 4000, "GATTACA" * 5 + "NNN" + "TGCA" * 9]
    bases, offsets = oracle.pack_records(recs)
    for omode, gmode in ((oracle.N_DROP, 0), (oracle.N_EXPAND, 1)):
        uniq, cnt, n_ids = _sparse_expect(oracle, recs, k, True, omode)
        with gpu_engine_cls(k, canonicalize=True, n_mode=gmode, algo=2) as eng:
            eng.submit(bases, offsets)
            _, total, unique = eng.finish(copy=False)
            got = _sparse_got(eng, uniq)
        assert total == n_ids and unique == uniq.size
        assert np.array_equal(got, cnt)


def test_sub_batching_beyond_2gi_positions(gpu_engine_cls):
    """A single device-resident batch of 2.2 Gi residues is cut into sub-batches of 2^31 positions inside the
    LDS-histogram paths (element indices are 32-bit): single-level, multi-pass and two-level must agree with the
    direct path bin for bin."""
    import torch
    n, L = 11_000_000, 210                     # 2.31e9 residues > 2^31
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
    d_b = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for s in range(0, n * L, step):
        e = min(n * L, s + step)
        d_b[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device="cuda", dtype=torch.uint8).long()]
    d_o = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
    torch.cuda.synchronize()
    for k in (9, 13, 14):
        ref = None
        for algo in (1, 2):
            with gpu_engine_cls(k, algo=algo) as eng:
                eng.submit_device(d_b.data_ptr(), n * L, d_o.data_ptr(), n)
                _, total, _ = eng.finish(copy=False)
                assert total == n * (L - k + 1)
                t = eng.table_tensor().clone()
            if ref is None:
                ref = t
            else:
                assert torch.equal(ref, t), k
        del ref, t


def test_real_genome_k12_vs_oracle(gpu_engine_cls, oracle, golden_dir):
    """E. coli K-12 (a data file of the reference's test suite; 4.6 Mbp, one record): real, skewed k-mer spectrum."""
    from kmerdb_amd import parse
    path = os.path.join(golden_dir, "ref_data", "Ecoli_K12MG1655.fasta.gz")
    recs = [s for _, s in oracle.read_records(path)]
    bases, offsets = oracle.pack_records(recs)
    for canon in (True, False):
        want, want_total = oracle.c_count(bases, offsets, 12, canon, oracle.N_EXPAND)
        got, meta, _ = parse.parsefile(path, 12, replace_with_none=False, canonicalize=canon)
        assert meta["total_kmers"] == want_total and np.array_equal(got, want)
        assert meta["unique_kmers"] == int(np.count_nonzero(want))


@pytest.mark.parametrize("k,accum", [(15, 3 << 20), (9, 1 << 20), (15, -1)])
def test_accumulated_submits(gpu_engine_cls, oracle, k, accum):
    """Staged chunks are appended on the device and counted as few large batches (the default for k >= 15, where
    every batch pays one sweep of the 4^k vector); many small submits, several flushes, pinned and pageable sources."""
    import kmerdb_amd
    from kmerdb_amd import synth
    parts = [synth.reads(n, L, seed=100 + i, p_n=0.002) for i, (n, L) in enumerate([(9000, 150), (500, 4000), (20000, 31), (7000, 150), (1, 600000)])]
    want_ids = []
    with gpu_engine_cls(k) as eng:
        eng.set_option("accum_bytes", accum)
        eng.set_option("stage_bytes", 1 << 20)
        for i, (b, o) in enumerate(parts):
            if i % 2:
                pb = kmerdb_amd.pinned_empty(b.size)
                pb[:] = b
                eng.submit_pinned(pb, o)
            else:
                eng.submit(b, o)
            oo = o.astype(np.int64)
            want_ids.append(np.concatenate([oracle.c_shred(bytes(b[oo[r]:oo[r + 1]]), k, True, oracle.N_DROP)[0] for r in range(len(oo) - 1)]))
        _, total, unique = eng.finish(copy=False)
        want = np.concatenate(want_ids)
        uniq, cnt = np.unique(want, return_counts=True)
        import torch
        t = eng.table_tensor()
        got = t[torch.as_tensor(uniq.astype(np.int64), device=t.device)].cpu().numpy()
    assert total == want.size and unique == uniq.size
    assert np.array_equal(got.astype(np.uint64), cnt.astype(np.uint64))


def test_all_n_reads_expand_quickly_and_exactly(gpu_engine_cls):
    """EXPAND mode (the reference CLI's default) on all-N reads: 4^12 fills per window.  They are queued and spread
    over whole workgroups (seconds from a single lane otherwise); the result is known in closed form."""
    import time
    k, nreads, L = 12, 2, 150
    bases = np.full(nreads * L, ord("N"), dtype=np.uint8)
    offsets = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L)
    nwin = nreads * (L - k + 1)
    ids = np.arange(4 ** k, dtype=np.uint64)
    rc = np.zeros_like(ids)
    x = ids.copy()
    for _ in range(k):
        rc = (rc << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    for canon in (False, True):
        for algo in ALGOS:
            t0 = time.perf_counter()
            got, total, unique = _count(gpu_engine_cls, bases, offsets, k, canon, 1, algo)
            dt = time.perf_counter() - t0
            assert total == nwin * 4 ** k
            if canon:
                want = np.zeros(4 ** k, dtype=np.uint64)
                np.add.at(want, np.minimum(ids, rc).astype(np.int64), np.uint64(nwin))
            else:
                want = np.full(4 ** k, nwin, dtype=np.uint64)
            assert np.array_equal(got, want)
            assert dt < 20, dt


def _table_checksum(t):
    """position-weighted sum of the vector (int64 wrap-around), 2^28 bins at a time"""
    import torch
    acc = 0
    step = 1 << 28
    for s0 in range(0, t.numel(), step):
        c = t[s0:s0 + step]
        w = (torch.arange(s0, s0 + c.numel(), device=t.device, dtype=torch.int64) % 1000003) + 1
        acc = (acc + int((c * w).sum().item())) & ((1 << 63) - 1)
    return acc


@pytest.mark.gpu
@pytest.mark.parametrize("k", [13, 15, 16])            # (k = 17: test_k17_bins_counted_more_than_65535_times_in_one_flush and config 4's tests run deferred passes over its 128 GiB vector)
def test_deferred_histogram_pass_over_many_batches(gpu_engine_cls, oracle, k):
    """k >= 14: batches are partitioned as they come and added to the vector together (at sync, or after 16 batches).
    The result must not depend on how many batches were pending, on reset() dropping them, or on the option."""
    from kmerdb_amd import synth
    import torch
    parts = [synth.reads(400 + 37 * i, 150, seed=100 + i) for i in range(35)]       # more than the arena's first size (8 batches) several times over
    ids = np.concatenate([np.concatenate([oracle.c_shred(bytes(b[int(o[r]):int(o[r + 1])]).decode(), k, True, oracle.N_DROP)[0]
                                          for r in range(0, len(o) - 1, 7)]) for b, o in parts[:3]])
    want_total = sum((len(o) - 1) * (151 - k) for _, o in parts)
    tables = []
    for defer in (1, 0, 2, 3):
        with gpu_engine_cls(k, algo=2) as eng:
            if k == 13:
                eng.set_option("one_level_max_k", 12)       # (k = 13 takes one scatter level by default: here the two-level path, whole vector against the oracle)
            eng.set_option("defer_flush", 1 if defer else 0)
            if defer == 2:
                eng.set_option("pending_budget", 1)         # every batch exceeds the budget: flushed at once, buffers reused from the pool
            if defer == 3:
                eng.set_option("arena_grow", 2)             # the arena doubles whenever it has filled up (default: only once that pays)
            eng.set_option("accum_bytes", 0)                # one device batch per submit (small submits are merged otherwise)
            eng.submit(*parts[0])
            eng.reset()                                     # pending batch dropped with the vector
            for n, (b, o) in enumerate(parts):
                eng.submit(b, o)
                if defer == 1 and n == 2:
                    assert eng.get_option("pending_batches") == 3
                if defer == 2:
                    assert eng.get_option("pending_batches") == 0
            if defer in (1, 3):      # the arena was flushed when it was full (it holds 8 batches' worst cases; arena_grow = 2: then twice that, ...)
                assert eng.get_option("pending_batches") < 35
                # (how many of these small batches an arena holds depends on the level-2 grid: a partial page per ring and workgroup)
                assert eng.get_option("arena_reallocs") == 1 if defer == 1 else eng.get_option("arena_reallocs") >= 2
            _, total, unique = eng.finish(copy=False)
            assert eng.get_option("pending_batches") == 0
            assert total == want_total
            t = eng.table_tensor()
            assert int(t.sum().item()) == want_total
            # sampled reads of the first three batches: every one of their ids is present at least as often as sampled
            uniq, cnt = np.unique(ids, return_counts=True)
            got = t[torch.as_tensor(uniq.astype(np.int64), device=t.device)].cpu().numpy().astype(np.uint64)
            assert np.all(got >= cnt.astype(np.uint64))
            tables.append(t.clone() if k < 16 else _table_checksum(t))     # (two more 128 GiB vectors do not fit at k = 17)
    for other in tables[1:]:
        assert torch.equal(tables[0], other) if k < 16 else tables[0] == other
    # and against the oracle on the whole input for one k (8 GiB vectors are compared on the device above)
    if k == 13:
        bases = np.concatenate([b for b, _ in parts])
        offs = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o.astype(np.int64)) for _, o in parts]))]).astype(np.uint64)
        want, _ = oracle.c_count(bases, offs, k, True, oracle.N_DROP)
        assert np.array_equal(tables[0].cpu().numpy().view(np.uint64), want)


@pytest.mark.gpu
def test_arena_is_charged_what_level_2_planned_not_the_worst_case(gpu_engine_cls, oracle):
    """k = 16, batches pending in the page arena: the device hands every batch the pages behind the previous one's last
    (l2_plan_kernel's cursor); the host's bound follows it through asynchronous read-backs and only assumes the worst case for
    batches it has not heard of yet.  cursor <= bound <= worst cases added up; once the read-backs have landed the bound IS the
    cursor; an arena of seven worst cases then takes an eighth batch without a forced flush; counts equal the oracle's."""
    from kmerdb_amd import synth
    k = 16
    parts = [synth.reads(30000, 150, seed=300 + i) for i in range(9)]
    with gpu_engine_cls(k, algo=2) as eng:
        eng.set_option("accum_bytes", 0)
        eng.submit(*parts[0])
        worst1 = eng.get_option("arena_worst_case")
        cur1 = eng.get_option("arena_cursor")               # (synchronises: the read-back of batch 0 has landed)
        assert 0 < cur1 < worst1 and eng.get_option("arena_used_bound") == cur1
        for b, o in parts[1:4]:
            eng.submit(b, o)
        assert eng.get_option("arena_used_bound") <= eng.get_option("arena_worst_case") == 4 * worst1
        cur4 = eng.get_option("arena_cursor")
        assert cur1 < cur4 == eng.get_option("arena_used_bound") < 4 * worst1
        assert eng.get_option("pending_batches") == 4 and eng.get_option("hist_flushes") == 0
    assert 7 * (cur4 / 4) * 1.02 + worst1 <= 7 * worst1, "geometry of the test: the eighth batch has to fit by the cursor's account"
    with gpu_engine_cls(k, algo=2) as eng:
        eng.set_option("accum_bytes", 0)
        eng.set_option("arena_grow", 0)
        eng.set_option("arena_batches", 7)
        for n, (b, o) in enumerate(parts[:8]):
            eng.submit(b, o)
            eng.get_option("arena_cursor")                   # (a host that is not ahead of the device: every read-back has landed)
            if n == 6:                                       # seven batches: by their worst cases the arena would be full now, and flushed
                assert eng.get_option("pending_batches") == 7 and eng.get_option("hist_flushes") == 0
        assert eng.get_option("arena_pages") == 7 * worst1
        # the eighth went in as well (it is flushed with the others at once if a ninth could not follow)
        assert eng.get_option("pending_batches") + eng.get_option("flushed_batches") == 8 and eng.get_option("hist_flushes") <= 1
        assert eng.get_option("pending_batches") == 8 or eng.get_option("flushed_batches") == 8
        eng.submit(*parts[8])
        _, total, unique = eng.finish(copy=False)
        recs = [bytes(b[int(o[r]):int(o[r + 1])]).decode() for b, o in parts for r in range(0, len(o) - 1, 97)]
        uniq, cnt, _ = _sparse_expect(oracle, recs, k, True, oracle.N_DROP)
        got = _sparse_got(eng, uniq)
        assert total == sum((len(o) - 1) * (151 - k) for _, o in parts) and np.all(got >= cnt)


# ---------------------------------------------------------------------------------------------------------------
# round 2: error stickiness, 8-bit bytes, read-only device input, on-device samplesheet sum, table-less shred
# ---------------------------------------------------------------------------------------------------------------
def test_bad_layout_is_not_erased_by_a_later_good_batch(gpu_engine_cls):
    """A device batch whose offsets do not tile the buffer, followed by a good batch BEFORE the sync, must still raise."""
    import torch
    from kmerdb_amd import synth
    bases, offsets = synth.reads(100, 50, seed=4)
    d_b = torch.from_numpy(bases).cuda()
    good = torch.from_numpy(offsets.view(np.int64).copy()).cuda()
    bad = torch.from_numpy((offsets + np.uint64(1)).view(np.int64).copy()).cuda()
    with gpu_engine_cls(8) as eng:
        eng.submit_device(d_b.data_ptr(), bases.size, bad.data_ptr(), len(offsets) - 1)
        eng.submit_device(d_b.data_ptr(), bases.size, good.data_ptr(), len(offsets) - 1)
        with pytest.raises(ValueError):
            eng.sync()
        eng.reset()                                        # a reset clears it
        eng.submit_device(d_b.data_ptr(), bases.size, good.data_ptr(), len(offsets) - 1)
        eng.sync()


@pytest.mark.parametrize("uniform", [True, False])
def test_bytes_with_bit_7_set_raise_on_host_fed_paths(gpu_engine_cls, golden_dir, uniform):
    """0xC1 is not 'A': the reference raises on any byte outside its alphabet (kmer.py:170); bit 7 is the engine's own
    record mark, so host-fed input is checked for it before the marks are placed (uniform and ragged batches)."""
    from kmerdb_amd import kmer, synth
    bases, offsets = synth.reads(300, 60, seed=8)
    if not uniform:
        offsets = np.concatenate([offsets[:-2], offsets[-1:]])      # last record twice as long
    for pos in (0, 61, bases.size - 1):
        b = bases.copy()
        b[pos] |= 0x80
        for k, algo in ((9, 2), (5, 2), (9, 1)):
            with gpu_engine_cls(k, algo=algo) as eng:
                eng.submit(b, offsets)
                with pytest.raises(ValueError):
                    eng.finish()
    import kmerdb_amd.engine as E
    with pytest.raises(ValueError):
        E.ids_engine(3, True, 0).shred(b"ACGT\xc1CGTA")          # kdb_shred checks too
    with pytest.raises(ValueError):
        kmer.shred("ACGT\u00c1CGTA", 3)


@pytest.mark.parametrize("k,algo", [(12, 2), (12, 1), (5, 2), (14, 2), (17, 2)])
def test_device_buffer_rebatched_with_other_ragged_offsets(gpu_engine_cls, oracle, k, algo):
    """kdb_submit_device marks record starts in the caller's buffer (bit 7 of a record's first byte) when the batch is
    ragged.  The marks come off after every batch, so the same device buffer can be cut into records differently in the
    next submit -- windows never span records of the CURRENT offsets (parse.py:128-131) and none is lost to a stale mark."""
    import torch
    rng = np.random.Generator(np.random.PCG64(k * 31 + algo))
    n = 3000
    lens_a = rng.integers(k, k + 200, size=n)
    total = int(lens_a.sum())
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=total)].copy()
    off_a = np.concatenate([[0], np.cumsum(lens_a)]).astype(np.uint64)
    # other cuts of the same bytes: different ragged records; then one record; then uniform records
    cuts = np.unique(np.concatenate([[0, total], rng.integers(0, total, size=n // 2)]))
    cuts = cuts[np.concatenate([[True], np.diff(cuts) >= k])]
    if total - cuts[-1] < k:
        cuts = cuts[:-1]
    off_b = np.concatenate([cuts[cuts < total], [total]]).astype(np.uint64)
    L = 50
    nu = total // L
    off_u = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(L))
    d_b = torch.from_numpy(bases).cuda()
    dev = {name: torch.from_numpy(o.view(np.int64)).cuda() for name, o in (("a", off_a), ("b", off_b), ("u", off_u))}
    torch.cuda.synchronize()

    def want_of(off, nbytes):
        if k <= 13:
            return oracle.c_count(bases[:nbytes], off, k, True, 0)
        ids = np.concatenate([oracle.c_shred(bytes(bases[int(off[r]):int(off[r + 1])]), k, True, 0)[0] for r in range(len(off) - 1)])
        return np.unique(ids, return_counts=True), ids.size

    with gpu_engine_cls(k, algo=algo) as eng:
        for name, off, nbytes in (("a", off_a, total), ("b", off_b, total), ("u", off_u, nu * L), ("a", off_a, total)):
            eng.reset()
            eng.submit_device(d_b.data_ptr(), nbytes, dev[name].data_ptr(), len(off) - 1)
            want, want_total = want_of(off, nbytes)
            if k <= 13:
                got, tot, _ = eng.finish()
                assert tot == want_total and np.array_equal(got, want), name
            else:
                _, tot, uniq = eng.finish(copy=False)
                (u, c) = want
                t = eng.table_tensor()
                g = t[torch.as_tensor(u.astype(np.int64), device=t.device)].cpu().numpy().astype(np.uint64)
                assert tot == want_total and uniq == u.size and np.array_equal(g, c.astype(np.uint64)), name
            assert np.array_equal(d_b.cpu().numpy(), bases), name                    # nothing of the engine's is left in the buffer
        # two cuts of the buffer in ONE job, without a sync between them
        eng.reset()
        eng.submit_device(d_b.data_ptr(), total, dev["a"].data_ptr(), len(off_a) - 1)
        eng.submit_device(d_b.data_ptr(), total, dev["b"].data_ptr(), len(off_b) - 1)
        _, tot, _ = eng.finish(copy=False)
        assert tot == want_of(off_a, total)[1] + want_of(off_b, total)[1]


@pytest.mark.parametrize("uniform", [True, False])
def test_device_buffers_with_bit_7_set_raise(gpu_engine_cls, uniform):
    """Device-resident input gets no checking pass of its own, and still nothing is silent: a byte with bit 7 set is not a
    residue (kmer.py:170 raises).  Uniform batches: the counting kernels' front end reports it; ragged batches: it is a
    record-start mark too many (or, on a record's first byte, reported by the marking kernel)."""
    import torch
    from kmerdb_amd import synth
    bases, offsets = synth.reads(400, 70, seed=21)
    if not uniform:
        offsets = np.concatenate([offsets[:-2], offsets[-1:]])
    d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    for pos in (0, 70, 71, bases.size - 1):
        b = bases.copy()
        b[pos] |= 0x80
        for k, algo in ((12, 2), (6, 2), (12, 1), (15, 2)):
            d_b = torch.from_numpy(b).cuda()               # (a fresh copy: taking the marks off a ragged batch also clears a record's first byte)
            torch.cuda.synchronize()
            with gpu_engine_cls(k, algo=algo) as eng:
                eng.submit_device(d_b.data_ptr(), b.size, d_o.data_ptr(), len(offsets) - 1)
                with pytest.raises(ValueError):
                    eng.sync()
                eng.reset()                                # the error is sticky until the reset; a clean buffer counts again
                d_c = torch.from_numpy(bases).cuda()
                eng.submit_device(d_c.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
                eng.sync()
                if uniform:
                    eng.submit_device_const(d_b.data_ptr(), b.size, d_o.data_ptr(), len(offsets) - 1)
                    with pytest.raises(ValueError):
                        eng.sync()


def test_const_device_submit_never_writes_the_buffer(gpu_engine_cls, oracle):
    import torch
    from kmerdb_amd import synth
    k = 11
    bases, offsets = synth.reads(5000, 101, seed=12)
    want, want_total = oracle.c_count(bases, offsets, k, True, 0)
    d_b = torch.from_numpy(bases).cuda()
    d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    with gpu_engine_cls(k) as e1, gpu_engine_cls(k, canonicalize=False) as e2:        # two engines share one read-only buffer
        e1.submit_device_const(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
        e2.submit_device_const(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
        got, total, _ = e1.finish()
        assert total == want_total and np.array_equal(got, want)
        got2, _, _ = e2.finish()
        assert np.array_equal(got2, oracle.c_count(bases, offsets, k, False, 0)[0])
    assert np.array_equal(d_b.cpu().numpy(), bases)                   # not a single bit was written
    # ragged records: the LDS-histogram paths read the record starts from the offsets and never write the residues, so a read-only
    # buffer is fine; the direct-atomics kernel marks record starts in the buffer and has to refuse it
    ragged = np.concatenate([offsets[:-2], offsets[-1:]])
    d_r = torch.from_numpy(ragged.view(np.int64).copy()).cuda()
    with gpu_engine_cls(k) as e:
        e.submit_device_const(d_b.data_ptr(), bases.size, d_r.data_ptr(), len(ragged) - 1)
        got, total, _ = e.finish()
        want_r, want_r_total = oracle.c_count(bases, ragged, k, True, 0)
        assert total == want_r_total and np.array_equal(got, want_r)
    with gpu_engine_cls(k, algo=1) as e:
        e.submit_device_const(d_b.data_ptr(), bases.size, d_r.data_ptr(), len(ragged) - 1)
        with pytest.raises(ValueError):
            e.sync()
    assert np.array_equal(d_b.cpu().numpy(), bases)


def test_samplesheet_vectors_are_summed_on_the_device(gpu_engine_cls, oracle, golden_dir, tmp_path):
    """profile() over a 4-file samplesheet: per-file metadata equal to parsefile's, summed vector equal to the sum of
    the oracle's vectors, and exactly ONE device-to-host copy of the 4^k vector (kmerdb/__init__.py:1888-1903)."""
    import kmerdb_amd
    from kmerdb_amd import parse, profile
    files = [os.path.join(golden_dir, f) for f in ("inputs/reads150.fq", "inputs/ragged_n.fq", "ref_data/sample.fa", "inputs/reads150.fq.gz")]
    sheet = str(tmp_path / "sheet.txt")
    open(sheet, "w").write("\n".join(files) + "\n")
    for k, no_amb, dnc in ((9, False, False), (14, True, True)):      # (ragged_n.fq's shortest record has 14 residues)
        d2h = []
        orig_close = kmerdb_amd.Engine.close

        def spy(self):
            if getattr(self, "_h", None) is not None and self._h and self.nbins:
                d2h.append(self.get_option("d2h_bytes"))
            orig_close(self)
        kmerdb_amd.Engine.close = spy
        try:
            counts, md, _ = profile.profile([sheet], k, str(tmp_path / "o"), no_ambiguous=no_amb, do_not_canonicalize=dnc, write=False)
        finally:
            kmerdb_amd.Engine.close = orig_close
        assert sum(d2h) == max(d2h) == 8 * 4 ** k, d2h                 # one copy of one vector for four files (counted by up to four engines at once)
        want = None
        for f, fm in zip(files, md["files"]):
            recs = [s for _, s in oracle.read_records(f)]
            b, o = oracle.pack_records(recs)
            omode = oracle.N_DROP if no_amb else oracle.N_EXPAND
            if k <= 13:
                w, wt = oracle.c_count(b, o, k, not dnc, omode)
                want = w if want is None else want + w
                assert fm["total_kmers"] == wt and fm["unique_kmers"] == int(np.count_nonzero(w))
            else:
                uniq, cnt, n_ids = _sparse_expect(oracle, recs, k, not dnc, omode)
                assert fm["total_kmers"] == n_ids and fm["unique_kmers"] == uniq.size
                want = (uniq, cnt) if want is None else (np.concatenate([want[0], uniq]), np.concatenate([want[1], cnt]))
            assert fm["nullomers"] == 4 ** k - fm["unique_kmers"] and fm["total_reads"] == len(recs)
        if k <= 13:
            assert np.array_equal(counts, want)
        else:
            u, inv = np.unique(want[0], return_inverse=True)
            c = np.zeros(u.size, dtype=np.uint64)
            np.add.at(c, inv, want[1])
            assert np.array_equal(counts[u.astype(np.int64)], c) and int(counts.sum()) == int(c.sum())
        assert md["total_kmers"] == int(counts.sum()) and md["unique_kmers"] == int(np.count_nonzero(counts))


def test_shred_allocates_no_count_vector(gpu_engine_cls, oracle):
    """kmer.shred at k = 15 used to build an Engine (8 GiB hipMalloc + memset) per call."""
    import torch
    from kmerdb_amd import kmer
    import kmerdb_amd.engine as E
    seq = "ACGTTGCAGGCTTAACGATCGATCGGCTA" * 3
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in (15, 17):
        for canon in (True, False):
            ids, _, pos = kmer.shred(seq, k, replace_with_none=True, canonicalize=canon)
            want_ids, want_pos = oracle.c_shred(seq, k, canon, oracle.N_DROP)
            assert ids == want_ids.tolist() and pos == want_pos.tolist()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (1 << 30), (free0, free1)
    e = E.ids_engine(15, True, 0)
    with pytest.raises(Exception):
        e.finish()


def test_full_rings_refuse_and_retry(gpu_engine_cls, oracle):
    """Paged scatter: a ring that is full refuses the element and the round is repeated after the flush.  Buckets taken
    from the LEADING id bits (option sc_lo_bits=15) are badly uneven for canonical ids, and a low-entropy alphabet makes
    them worse: many refusals per round, same counts.  Every other place of the bucket field in the id gives the same vector."""
    rng = np.random.Generator(np.random.PCG64(3))
    n, L, k = 40000, 150, 12
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n * L, p=[0.7, 0.1, 0.1, 0.1])]
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    want, want_total = oracle.c_count(bases, offsets, k, True, oracle.N_DROP, nthreads=8)
    for lo in (15, 1, 6, 9, 12, 14):
        with gpu_engine_cls(k, algo=2) as eng:
            eng.set_option("sc_lo_bits", lo)
            assert eng.get_option("sc_lo_bits") == lo
            eng.submit(bases, offsets)
            got, total, _ = eng.finish()
        assert total == want_total and np.array_equal(got, want), lo
    # few workgroups, many tiles each: page sequences wrap through many pages per ring
    with gpu_engine_cls(k, algo=2) as eng:
        eng.set_option("sc_grid", 3)
        eng.submit(bases, offsets)
        got, total, _ = eng.finish()
        assert eng.get_option("arena_grow") == 1                  # the default: the page arena of k >= 13 grows when that pays
        for bad in (("arena_grow", 3), ("sc_lo_bits", 16), ("no_such_option", 1)):
            with pytest.raises(ValueError):
                eng.set_option(*bad)
    assert total == want_total and np.array_equal(got, want)


def test_fasta_is_streamed_in_blocks_with_overlapping_pieces(gpu_engine_cls, oracle, golden_dir, monkeypatch):
    """parsefile streams FASTA: with blocks far smaller than the records, every record goes through the engine in
    pieces that overlap by k - 1 residues (kdb_submit_ex, KDB_SUBMIT_CONTINUES).  Same vector, same read statistics."""
    from kmerdb_amd import parse, reader
    for fname, k in (("ref_data/sample.fa", 11), ("inputs/contigs.fa", 9), ("ref_data/Ecoli_K12MG1655.fasta.gz", 13)):
        path = os.path.join(golden_dir, fname)
        recs = [s for _, s in oracle.read_records(path)]
        bases, offsets = oracle.pack_records(recs)
        want, want_total = oracle.c_count(bases, offsets, k, True, oracle.N_EXPAND, nthreads=8)
        for block in ((5000, 70000) if "Ecoli" not in fname else (300000,)):
            monkeypatch.setattr(reader, "BLOCK_BYTES", block)
            got, meta, _ = parse.parsefile(path, k, replace_with_none=False, canonicalize=True)
            assert meta["total_kmers"] == want_total and np.array_equal(got, want), (fname, block)
            lens = [len(r) for r in recs]
            assert (meta["total_reads"], meta["min_read_length"], meta["max_read_length"], meta["avg_read_length"]) == \
                (len(lens), min(lens), max(lens), int(sum(lens) / len(lens)))


def test_bgzf_input_is_inflated_block_parallel(gpu_engine_cls, oracle, golden_dir, tmp_path):
    """A bgzip-style FASTQ: the reader takes the native block-parallel inflate; counts equal the oracle's on the same records."""
    from kmerdb_amd import fileutil, parse, reader
    src = os.path.join(golden_dir, "inputs", "reads150.fq")
    data = open(src, "rb").read() * 40
    p = str(tmp_path / "reads.fq.gz")
    with open(p, "wb") as f:
        for i in range(0, len(data), 65280):
            f.write(fileutil._bgzf_member(data[i:i + 65280]))
        f.write(fileutil._bgzf_member(b""))
    assert reader.is_bgzf(p) and isinstance(reader._open(p), reader._BgzfFile)
    got, meta, _ = parse.parsefile(p, 10)
    # the checker is the oracle on the records of the plain file (40 copies of them), not the HIP path on another file format
    recs = [s for _, s in oracle.read_records(src)]
    bases, offsets = oracle.pack_records(recs)
    want, want_total = oracle.c_count(bases, offsets, 10, True, oracle.N_DROP)
    assert np.array_equal(got, want * np.uint64(40)) and meta["total_kmers"] == 40 * want_total and meta["total_reads"] == 40 * len(recs)


@pytest.mark.parametrize("k,n_eng,root", [(2, 3, 1), (7, 2, 0), (12, 2, 0), (12, 5, 3), (15, 2, 1), (15, 3, 0)])
def test_kdb_reduce_sums_engines_of_one_process(gpu_engine_cls, oracle, k, n_eng, root):
    """kdb_reduce (one process driving several engines; here all on device 0): records dealt out between
    the engines, the vectors summed slice by slice into the root's == the oracle's vector of all the records, and the
    root's finish() reports the whole job (Sum == emitted by all).  k = 15: the deferred histogram pass is flushed first."""
    from kmerdb_amd.engine import reduce_engines
    rng = np.random.Generator(np.random.PCG64(4242 + 31 * k + n_eng))
    letters = np.array(list("ACGTN"))
    recs = ["".join(letters[rng.choice(5, size=int(rng.integers(k, 300)), p=[0.2495] * 4 + [0.002])]) for _ in range(600)]
    recs += ["A" * 200, "ACGT" * 40]
    for canon, omode, gmode in ((True, oracle.N_DROP, 0), (False, oracle.N_EXPAND, 1)):
        engines = [gpu_engine_cls(k, canonicalize=canon, n_mode=gmode) for _ in range(n_eng)]
        try:
            for j, e in enumerate(engines):
                mine = recs[j::n_eng]
                half = len(mine) // 2
                for part in (mine[:half], mine[half:]):
                    b, o = oracle.pack_records(part)
                    e.submit(b, o)
            reduce_engines(engines, root=root)
            if k <= 13:
                b, o = oracle.pack_records(recs)
                want, want_total = oracle.c_count(b, o, k, canon, omode)
                got, total, unique = engines[root].finish()
                assert total == want_total and unique == int(np.count_nonzero(want))
                assert np.array_equal(got, want), (k, n_eng, root, canon)
            else:
                uniq, cnt, n_ids = _sparse_expect(oracle, recs, k, canon, omode)
                _, total, unique = engines[root].finish(copy=False)
                assert total == n_ids and unique == uniq.size
                assert np.array_equal(_sparse_got(engines[root], uniq), cnt), (k, n_eng, root, canon)
            # the others no longer hold a vector of their own: their consistency check says so until they are reset
            other = engines[(root + 1) % n_eng]
            if n_eng > 1 and k >= 7:
                with pytest.raises(Exception):
                    other.finish(copy=False)
            other.reset()
            b, o = oracle.pack_records(recs[:5])
            other.submit(b, o)
            assert other.finish(copy=False)[1] == sum(len(oracle.c_shred(r, k, canon, omode)[0]) for r in recs[:5])
        finally:
            for e in engines:
                e.close()


def test_kdb_reduce_rejects_bad_arguments_and_surfaces_shard_errors(gpu_engine_cls, oracle):
    from kmerdb_amd.engine import reduce_engines
    a, b, c = gpu_engine_cls(9), gpu_engine_cls(9), gpu_engine_cls(10)
    try:
        with pytest.raises(ValueError):
            reduce_engines([a], root=0)
        with pytest.raises(ValueError):
            reduce_engines([a, a], root=0)
        with pytest.raises(ValueError):
            reduce_engines([a, c], root=0)
        with pytest.raises(ValueError):
            reduce_engines([a, b], root=2)
        ba, oa = oracle.pack_records(["ACGTACGTACGTAC"])
        a.submit(ba, oa)
        bb, ob = oracle.pack_records(["ACGTACGTACGTAC", "ACGTRCGTACGTAC"])       # R: not ACGTN (kmer.py:309 raises)
        b.submit(bb, ob)
        with pytest.raises(ValueError):
            reduce_engines([a, b], root=0)
        assert a.finish(copy=False)[1] == 6                # nothing was reduced into the root
    finally:
        for e in (a, b, c):
            e.close()


@pytest.mark.parametrize("name,k,no_amb", [("inputs/reads150.fq", 12, True), ("inputs/ragged_n.fq", 9, False), ("ref_data/sample.fa", 14, True),
                                           ("inputs/reads150.fq.gz", 8, True)])
def test_parsefile_devices_equals_parsefile(gpu_engine_cls, golden_dir, name, k, no_amb):
    """parse.parsefile_devices (one process, an engine + reader thread per device, kdb_reduce) on devices [0, 0, 0] with
    small blocks == parse.parsefile on one engine: vector, metadata, nullomers."""
    from kmerdb_amd import parse
    path = os.path.join(golden_dir, name)
    c1, m1, n1 = parse.parsefile(path, k, replace_with_none=no_amb)
    c3, m3, n3 = parse.parsefile_devices(path, k, [0, 0, 0], replace_with_none=no_amb, block_bytes=1 << 14)
    assert m1 == m3
    assert np.array_equal(c1, c3) and np.array_equal(n1, n3)


def test_k17_bins_counted_more_than_65535_times_in_one_flush(gpu_engine_cls, oracle):
    """k = 17: the histogram pass keeps two 16-bit counters per LDS word (bins v and v | 0x8000).  Two 17-mers that share
    a word (they differ in the leading bit of the first base: A.. / G..) each occur once in > 131072 reads at scattered
    offsets (never 16 lanes of a wave with one id: they travel through the rings, not the hot-id table), so both halves wrap
    twice, the low half's carries land in a counting high half, and the vector must still equal the oracle's."""
    k = 17
    rng = np.random.Generator(np.random.PCG64(1717))
    tail = "CGTTGCATCAGGTCAT"                                   # 16 residues
    kmers = ("A" + tail, "G" + tail)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_each = 140000
    recs = []
    for which in (0, 1):
        flank = letters[rng.integers(0, 4, size=(n_each, 100))]
        offs = rng.integers(0, 100 - k, size=n_each)
        km = np.frombuffer(kmers[which].encode(), dtype=np.uint8)
        for i in range(n_each):
            row = flank[i]
            row[offs[i]:offs[i] + k] = km
        recs += [bytes(r).decode() for r in flank]
    order = rng.permutation(len(recs))
    recs = [recs[i] for i in order]
    bases, offsets = oracle.pack_records(recs)
    for canon in (False, True):
        uniq, cnt, n_ids = _sparse_expect(oracle, recs, k, canon, oracle.N_DROP)        # every id of every read, from the oracle
        key = np.array([oracle.c_shred(km, k, canon, oracle.N_DROP)[0][0] for km in kmers], dtype=np.uint64)
        assert all(int(cnt[np.searchsorted(uniq, kk)]) >= n_each for kk in key)
        for defer in (1, 0):
            with gpu_engine_cls(k, canonicalize=canon, n_mode=0, algo=2) as eng:
                eng.set_option("defer_flush", defer)
                half = len(offsets) // 2
                eng.submit(bases[:int(offsets[half])], offsets[:half + 1])
                eng.submit(bases[int(offsets[half]):], offsets[half:] - offsets[half])
                _, total, unique = eng.finish(copy=False)
                got = _sparse_got(eng, uniq)
            assert total == n_ids == len(recs) * (100 - k + 1) and unique == uniq.size
            assert np.array_equal(got, cnt), (canon, defer)


def test_k8_lds_histogram_halves_wrap_exactly(gpu_engine_cls, oracle):
    """k = 8 lives in one CU's LDS as two 16-bit counters per word (bins v and v | 0x8000: count_smallk_kernel).  Two 8-mers that
    share a word (AAAAAAAA = id 0, GAAAAAAA = id 32768) are diluted 1 : 7 in random 8-mers -- fewer than 16 lanes of a wave
    hold one of them, so they take the plain one-atomic path -- and four workgroups see > 3 x 65536 of each: both halves
    wrap several times, low carries land in a counting high half, and the vector must equal the oracle's.  Then the same two
    8-mers undiluted (every lane of a wave holds the same id: the add-once-for-the-wave path, counts of 64 crossing 0xFFFF), a
    ragged variant, and a list of notes that overflows (one workgroup, > 1024 wraps)."""
    k = 8
    rng = np.random.Generator(np.random.PCG64(808))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    hot = np.frombuffer(b"AAAAAAAAGAAAAAAA", dtype=np.uint8).reshape(2, 8)

    def reads(n, dilution, L=8):
        rows = letters[rng.integers(0, 4, size=(n, L))]
        pick = rng.integers(0, 2 * dilution, size=n)
        for w in (0, 1):
            rows[pick == w, :8] = hot[w]
        return rows

    cases = []
    rows = reads(12_000_000, 8)
    cases.append(("diluted", rows.reshape(-1).copy(), np.arange(rows.shape[0] + 1, dtype=np.uint64) * np.uint64(8), 2))
    rows = reads(1_200_000, 1)
    cases.append(("undiluted", rows.reshape(-1).copy(), np.arange(rows.shape[0] + 1, dtype=np.uint64) * np.uint64(8), 4))
    rows = reads(1_000_000, 4, L=11)
    lens = rng.integers(8, 12, size=rows.shape[0])
    keep = (np.arange(11)[None, :] < lens[:, None])
    cases.append(("ragged", rows[keep].copy(), np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), 2))
    # one workgroup, 72 M windows over the four rotations of (ACGT)n: > 1024 wrap notes, the list overflows into direct adds
    rows = np.tile(np.frombuffer(b"ACGT" * 38, dtype=np.uint8)[:150], (505_000, 1))
    cases.append(("notes-overflow", rows.reshape(-1).copy(), np.arange(rows.shape[0] + 1, dtype=np.uint64) * np.uint64(150), 1))
    for name, bases, offsets, grid in cases:
        for canon in (False, True):
            want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_DROP, nthreads=8)
            if name == "diluted" and not canon:
                assert want[0] > 5 * 65536 * grid and want[32768] > 5 * 65536 * grid
            with gpu_engine_cls(k, canonicalize=canon) as eng:
                eng.set_option("sc_grid", grid)
                eng.submit(bases, offsets)
                got, total, unique = eng.finish()
            assert total == want_total and unique == int(np.count_nonzero(want)), (name, canon)
            assert np.array_equal(got, want), (name, canon)


@pytest.mark.parametrize("k", [8, 12, 13, 15])
def test_n_dense_reads_expand_through_rings_and_beyond_their_allowance(gpu_engine_cls, oracle, k):
    """N-expansion mode (the reference CLI's default, kmer.py:545-565): the 4 or 16 fills of a window with one or two N's travel
    through the rings (k <= 8: the LDS histogram) like every other id.  A workgroup's page sequence has room for half as many
    fills as it has window positions; reads with an N every few bases make 10-16 fills per position, so most of them take the
    direct path to the vector -- every count must still equal the oracle's.  Then a sparse-N batch (all fills through the rings)."""
    rng = np.random.Generator(np.random.PCG64(1300 + k))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    for name, nreads, step in (("dense", 3000, max(k // 2, 4)), ("sparse", 30000, 97)):
        rows = letters[rng.integers(0, 4, size=(nreads, 150))].copy()
        for r in range(nreads):
            rows[r, int(rng.integers(0, step))::step] = 78
        bases = rows.reshape(-1).copy()
        offsets = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(150)
        for canon in (True, False):
            want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_EXPAND, nthreads=8)
            for grid in (0, 2):
                with gpu_engine_cls(k, canonicalize=canon, n_mode=1) as eng:
                    if grid:
                        eng.set_option("sc_grid", grid)                 # many tiles per workgroup: the allowance runs out
                    eng.submit(bases, offsets)
                    if k <= 13:
                        got, total, _ = eng.finish()
                        assert total == want_total and np.array_equal(got, want), (name, k, canon, grid)
                    else:
                        _, total, unique = eng.finish(copy=False)
                        nz = np.flatnonzero(want)
                        got = _sparse_got(eng, nz.astype(np.uint64))
                        assert total == want_total and unique == nz.size and np.array_equal(got, want[nz]), (name, k, canon, grid)


def test_iupac_codes_next_to_n_follow_the_reference(gpu_engine_cls, oracle, golden_dir):
    """The reference's kmer_to_id returns None for a window that holds an N before it meets another IUPAC code (kmer.py:287-289):
    with replace_with_none=True a record whose codes are all shielded by N's is accepted and those windows are dropped
    (kmer.py:541-544); a code in a window without N raises (kmer.py:309).  With replace_with_none=False every such record raises
    (the reference's substitution code: kmer.py:545-555, :612) -- except the shape named in tests/test_oracle_golden.py, where the
    reference returns counts and this engine raises.  Vectors: tests/golden/iupac_next_to_n.json, made by the
    reference's own kmer.shred / parse.parsefile."""
    import json
    from kmerdb_amd import kmer, parse
    from test_oracle_golden import IUPAC_EXPAND_EXCEPTIONS
    g = json.load(open(os.path.join(golden_dir, "iupac_next_to_n.json")))
    checked = 0
    for c in g["shred"]:
        seq, k, rwn, canon = c["seq"], c["k"], c["replace_with_none"], c["canonicalize"]
        bases = np.frombuffer(seq.encode(), dtype=np.uint8)
        offsets = np.array([0, len(seq)], dtype=np.uint64)
        reference_returns = c["raises"] is None and not (not rwn and (seq, k) in IUPAC_EXPAND_EXCEPTIONS)
        for algo in ALGOS:
            with gpu_engine_cls(k, canonicalize=canon, n_mode=0 if rwn else 1, algo=algo) as eng:
                eng.submit(bases, offsets)
                if reference_returns:
                    got, total, _ = eng.finish()
                    assert [int(x) for x in got] == c["counts"] and total == len(c["ids"]), (seq, k, rwn, canon, algo)
                else:
                    with pytest.raises(ValueError):
                        eng.finish()
        if reference_returns:
            ids, _, pos = kmer.shred(seq, k, replace_with_none=rwn, canonicalize=canon)
            assert (ids, pos) == (c["ids"], c["pos"]), (seq, k, rwn, canon)
        else:
            with pytest.raises(ValueError):
                kmer.shred(seq, k, replace_with_none=rwn, canonicalize=canon)
        checked += 1
    assert checked == len(g["shred"])
    for c in g["parsefile"]:
        path = os.path.join(golden_dir, c["file"])
        if c["raises"]:
            with pytest.raises(ValueError):
                parse.parsefile(path, c["k"], replace_with_none=c["replace_with_none"], canonicalize=c["canonicalize"])
        else:
            got, meta, _ = parse.parsefile(path, c["k"], replace_with_none=c["replace_with_none"], canonicalize=c["canonicalize"])
            assert [int(x) for x in got] == c["counts"]
            assert {kk: meta[kk] for kk in ("total_reads", "total_kmers", "unique_kmers", "nullomers")} == {kk: c["metadata"][kk] for kk in ("total_reads", "total_kmers", "unique_kmers", "nullomers")}


@pytest.mark.parametrize("k", [5, 8, 12, 13, 15])
def test_shielded_iupac_codes_in_large_batches(gpu_engine_cls, oracle, k):
    """The same rule through every counting path at scale (the oracle, which restates it, is the checker): reads with IUPAC codes whose
    every window also holds an N -- uniform and ragged batches, codes at record starts and ends and across chunk and tile borders --
    are counted with their N-windows dropped; one code that an N does not shield (k - 1 N-free residues on one side... and the rest
    on the other) makes the whole job raise."""
    rng = np.random.Generator(np.random.PCG64(4100 + k))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    codes = np.frombuffer(b"RYSWKMBDHV", dtype=np.uint8)

    def make(nreads, ragged):
        recs = []
        for r in range(nreads):
            L = int(rng.integers(k + 20, 260)) if ragged else 151
            row = letters[rng.integers(0, 4, size=L)].copy()
            for _ in range(int(rng.integers(0, 3))):                       # up to two shielded codes per read
                p = int(rng.integers(0, L))
                row[p] = codes[rng.integers(0, 10)]
                # an N within k - 1 on either side so that no N-free window of k holds p: gaps a + b < k - 1 ... a + b + 1 <= k - 1
                a = int(rng.integers(0, k - 1))
                b = k - 2 - a
                lo, hi = p - 1 - int(rng.integers(0, a + 1)), p + 1 + int(rng.integers(0, b + 1))
                if lo >= 0:
                    row[lo] = 78
                if hi < L:
                    row[hi] = 78
                # (a code within reach of a record end needs no N on that side)
                if lo < 0 and p >= k - 1 - (hi - p - 1 if hi < L else 0):
                    row[max(p - 1, 0)] = 78 if p > 0 else row[0]
            recs.append(row)
        return recs

    def shielded(rec):
        s = bytes(rec).decode()
        return all(("N" in s[i:i + k]) or not (set(s[i:i + k]) - set("ACGT")) for i in range(len(s) - k + 1))

    for ragged in (False, True):
        recs = [r for r in make(6000, ragged) if shielded(r)]
        assert len(recs) > 4000 and sum(1 for r in recs if set(bytes(r)) - set(b"ACGTN")) > 1000
        bases = np.concatenate(recs)
        offsets = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
        for canon in (True, False):
            want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_DROP)
            for algo in ALGOS:
                with gpu_engine_cls(k, canonicalize=canon, algo=algo) as eng:
                    if k >= 8 and algo == 2:
                        eng.set_option("sc_grid", 3)                    # several tiles per workgroup
                    eng.submit(bases, offsets)
                    got, total, _ = eng.finish() if k <= 13 else (None,) + eng.finish(copy=False)[1:]
                    assert total == want_total, (k, ragged, canon, algo)
                    if got is not None:
                        assert np.array_equal(got, want), (k, ragged, canon, algo)
                # N-expansion mode refuses every such record (the reference raises there, too)
                with gpu_engine_cls(k, canonicalize=canon, n_mode=1, algo=algo) as eng:
                    eng.submit(bases, offsets)
                    with pytest.raises(ValueError):
                        eng.finish(copy=False)
        # one unshielded code
        bad = [r.copy() for r in recs[:50]]
        victim = bad[17]
        victim[:] = letters[rng.integers(0, 4, size=len(victim))]
        victim[len(victim) // 2] = 82                                   # an R with k N-free residues around it
        b2 = np.concatenate(bad)
        o2 = np.concatenate([[0], np.cumsum([len(r) for r in bad])]).astype(np.uint64)
        for algo in ALGOS:
            with gpu_engine_cls(k, algo=algo) as eng:
                eng.submit(b2, o2)
                with pytest.raises(ValueError):
                    eng.finish(copy=False)


@pytest.mark.parametrize("k", [2, 3, 4, 9])
def test_reads_of_a_few_bases_get_their_record_starts(gpu_engine_cls, oracle, k):
    """Ragged batches of reads only k .. k + 4 bases long, several tiles of them: a tile's walk through the offsets begins at the record
    that holds the 4 KiB boundary below it, and a whole round of records can end before the tile begins (round 4: the test for
    "nothing behind this round starts inside the tile" wrapped there and the tile got no record starts -- every window across a
    record boundary was counted; found by tests/fuzz_gpu.py at k = 2)."""
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for seed, nreads in ((0, 5000), (3, 5000), (6, 30000)):
        rng = np.random.Generator(np.random.PCG64(1000 * k + seed))
        lens = rng.integers(k, k + 5, size=nreads)
        lens[::97] = k + 40                                   # (never all of one length)
        p_n = 0.002
        bases = letters[rng.choice(5, size=int(lens.sum()), p=[(1 - p_n) / 4] * 4 + [p_n])].copy()
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        for canon, omode, gmode in ((True, oracle.N_EXPAND, 1), (False, oracle.N_DROP, 0)):
            want, want_total = oracle.c_count(bases, offsets, k, canon, omode)
            for algo in ALGOS:
                got, total, _ = _count(gpu_engine_cls, bases, offsets, k, canon, gmode, algo)
                assert total == want_total and np.array_equal(got, want), (k, seed, canon, algo)


@pytest.mark.parametrize("k", [12, 13, 15, 16])
def test_batches_without_a_countable_window(gpu_engine_cls, oracle, k):
    """Reads that are long enough but hold an N in every window (drop mode): nothing is counted, nothing fails -- also when the device has
    already told the host that the batch took no page of the arena by the time the histogram pass is due (k >= 14: round 4 launched an
    empty grid there; found by tests/fuzz_gpu.py), and a batch that does count afterwards is counted."""
    import torch
    unit = "ACGTACG"[: min(7, k - 1)] + "N"
    recs = [unit * 12, unit * 9 + unit[:3], "N" * (k + 5)]
    assert all(len(r) >= k for r in recs)
    bases, offsets = oracle.pack_records(recs)
    good = ["ACGTTGCAAGGCTTAACCGGTTAAGGCC" * 3, "TTGACCAGTAGGATCCAGTACCAGATTACA" * 2]
    gb, go = oracle.pack_records(good)
    want, want_total = oracle.c_count(gb, go, k, True, oracle.N_DROP) if k <= 13 else (None, sum(len(r) - k + 1 for r in good))
    d_b = torch.from_numpy(bases.copy()).cuda()
    d_o = torch.from_numpy(offsets.view(np.int64).copy()).cuda()
    for device in (True, False):
        with gpu_engine_cls(k, n_mode=0) as eng:
            if device:
                eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(recs))
                torch.cuda.synchronize()                  # (the device's account of the batch has reached the host before the sync below)
            else:
                eng.submit(bases, offsets)
            eng.sync()
            _, total, unique = eng.table_stats(copy=False)
            assert (total, unique) == (0, 0), (k, device)
            eng.submit(gb, go)
            got, total, unique = eng.finish(copy=k <= 13)
            assert total == want_total, (k, device)
            if k <= 13:
                assert np.array_equal(got, want)


def test_scratch_that_does_not_fit_falls_back_to_direct_atomics(gpu_engine_cls, oracle):
    """No room in HBM for the scatter scratch: the batch is counted with direct atomics instead (same vector), the engine
    says so (`oom_fallbacks`), and the next batch goes through the LDS-histogram path again once memory is back."""
    import torch
    from kmerdb_amd import synth
    k = 13
    bases, offsets = synth.reads(200000, 100, seed=77)                  # 20 MB of residues; level 1 alone wants > 200 MB of pages
    want, want_total = oracle.c_count(bases, offsets, k, True, oracle.N_DROP)
    d_b = torch.from_numpy(bases).cuda()
    d_o = torch.from_numpy(offsets.view(np.int64).copy()).cuda()
    with gpu_engine_cls(k, algo=2) as eng:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
        # fill the device: one large block, then smaller and smaller ones until none fits (what the runtime reports as free is not
        # always all it will hand out), then give 32 MiB back -- room for the engine's small per-batch arrays, not for 300 MB of pages
        hog = [torch.empty(free - (1 << 30), dtype=torch.uint8, device="cuda")]
        for size in (256 << 20, 16 << 20, 1 << 20):
            while True:
                try:
                    hog.append(torch.empty(size, dtype=torch.uint8, device="cuda"))
                except torch.OutOfMemoryError:
                    break
        given_back = 0
        while given_back < (32 << 20):
            i = next((j for j in range(len(hog) - 1, -1, -1) if hog[j].numel() <= (16 << 20)), None)
            if i is None:
                break                # (some boxes hand out nothing behind the large block: then there is nothing to give back, and no room for pages either)
            given_back += hog.pop(i).numel()
        torch.cuda.empty_cache()
        try:
            eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
            eng.sync()
            assert eng.get_option("oom_fallbacks") == 1, (free, torch.cuda.mem_get_info(), given_back, len(hog))
        finally:
            del hog
            torch.cuda.empty_cache()
        eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
        got, total, unique = eng.finish()
        assert eng.get_option("oom_fallbacks") == 1
    assert total == 2 * want_total and unique == int(np.count_nonzero(want))
    assert np.array_equal(got, want * np.uint64(2))


@pytest.mark.parametrize("k", [1, 2, 3, 6, 8, 12, 13, 14])
def test_nullomers_are_compacted_on_the_device(gpu_engine_cls, k):
    """nullomer_array of parse.py:139-140 (the ids whose count is zero, ascending) from kdb_nullomers == np.flatnonzero of the
    copied-back vector: dense vectors (no nullomer at all), sparse ones, several ranges of tiles (k >= 13), the count-only call,
    a capacity that is too small, and the samplesheet accumulator."""
    import ctypes
    import kmerdb_amd
    from kmerdb_amd import synth
    for n_reads, canon in ((2000, False), (3, True), (10000, True))[:(2 if k >= 14 else 3)]:
        bases, offsets = synth.reads(n_reads, 60, seed=900 + k + n_reads)
        with gpu_engine_cls(k, canonicalize=canon) as eng:
            eng.submit(bases, offsets)
            counts, total, unique = eng.finish()
            want = np.flatnonzero(counts == 0).astype(np.uint64)
            assert want.size == 4 ** k - unique
            got = eng.nullomers()                                   # asks the device for the number first
            assert got.dtype == np.uint64 and np.array_equal(got, want)
            assert np.array_equal(eng.nullomers(n=want.size), want)
            if want.size:
                small = np.empty(max(want.size - 1, 1), dtype=np.uint64)
                n = ctypes.c_uint64(0)
                rc = eng._lib.kdb_nullomers(eng._h, 0, small.ctypes.data, want.size - 1, ctypes.byref(n))
                assert rc == kmerdb_amd._abi.KDB_ERR_ARG and n.value == want.size
            with pytest.raises(kmerdb_amd._abi.KdbHipError):
                eng.nullomers(folded=True)                          # nothing was folded yet
            eng.fold_file()
            eng.submit(bases[:offsets[1]], offsets[:2])             # one more record into the (cleared) file vector
            eng.fold_file()
            acc, _, _ = eng.finish_folded()
            assert np.array_equal(eng.nullomers(folded=True), np.flatnonzero(acc == 0).astype(np.uint64))


def test_parsefile_keeps_its_engine_between_calls(gpu_engine_cls, oracle, golden_dir):
    """parse.parsefile called in a loop (kmerdb/__init__.py:1888-1891) reuses one engine per parameter set instead of creating and
    destroying one per file; results are those of fresh engines, an error does not poison the pool, release_engines() frees it."""
    from kmerdb_amd import parse
    parse.release_engines()
    a = os.path.join(golden_dir, "inputs", "reads150.fq")
    b = os.path.join(golden_dir, "inputs", "ragged_n.fq")
    first = parse.parsefile(a, 9)
    assert len(parse._pool) == 1
    eng = next(iter(parse._pool.values()))
    second = parse.parsefile(a, 9)
    assert next(iter(parse._pool.values())) is eng                 # the same engine served both calls
    other = parse.parsefile(b, 9, replace_with_none=False)         # other parameters: a second engine
    assert len(parse._pool) == 2
    again = parse.parsefile(a, 9)
    for x in (second, again):
        assert np.array_equal(x[0], first[0]) and x[1] == first[1] and np.array_equal(x[2], first[2])
    recs = [s for _, s in oracle.read_records(b)]
    bb, oo = oracle.pack_records(recs)
    w, wt = oracle.c_count(bb, oo, 9, True, oracle.N_EXPAND)
    assert np.array_equal(other[0], w) and other[1]["total_kmers"] == wt and np.array_equal(other[2], np.flatnonzero(w == 0).astype(np.uint64))
    with pytest.raises(ValueError):
        parse.parsefile(b, 15, replace_with_none=True)                       # ragged_n.fq holds a 14-residue record: shorter than k
    assert np.array_equal(parse.parsefile(a, 9)[0], first[0])      # the pool still works after a failed call
    parse.release_engines()
    assert not parse._pool


@pytest.mark.parametrize("k,algo", [(12, 2), (12, 1), (6, 2), (15, 2)])
def test_more_shielded_iupac_codes_than_the_suspects_list_holds(gpu_engine_cls, oracle, k, algo):
    """A masked assembly: more than 65 536 IUPAC codes in one batch, each next to an N (the list of suspects holds
    65 536; what did not fit used to be an error at once, the reference -- kmer.py:287-289, :541-544 -- drops those windows and counts).
    The overflow makes resolve_suspects_kernel judge every residue of the batch itself: same counts as the oracle; and ONE code that no N
    shields among them still raises."""
    rng = np.random.Generator(np.random.PCG64(77 + k))
    n_reads, L = 8000, 400
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n_reads * L)].copy()
    codes = np.frombuffer(b"RYSWKMBDHV", dtype=np.uint8)
    rows = bases.reshape(n_reads, L)
    rows[:, 5::40] = codes[rng.integers(0, 10, size=rows[:, 5::40].shape)]       # a code every 40 residues ...
    rows[:, 6::40] = 78                                                           # ... between two N's: every window that holds the code holds one of them
    rows[:, 4::40] = 78
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
    n_codes = int(np.isin(bases, codes).sum())
    assert n_codes > 65536 + 1000
    want, want_total = oracle.c_count(bases, offsets, k, True, oracle.N_DROP)
    with gpu_engine_cls(k, algo=algo) as eng:
        eng.submit(bases, offsets)
        got, total, _ = eng.finish() if k <= 13 else (None,) + eng.finish(copy=False)[1:]
        assert total == want_total
        if got is not None:
            assert np.array_equal(got, want)
    bad = bases.copy()
    row = bad.reshape(n_reads, L)[n_reads // 2]
    row[100:160] = 65
    row[130] = 82                                                                 # an R with 30 A's on either side: no N in reach
    with gpu_engine_cls(k, algo=algo) as eng:
        eng.submit(bad, offsets)
        with pytest.raises(ValueError, match="outside ACGTN"):
            eng.finish(copy=False)
        _, n_bad = _error_counts(eng)
        assert n_bad == 1


def _error_counts(eng):
    import ctypes
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert eng._lib.kdb_error_counts(eng._h, ctypes.byref(a), ctypes.byref(b)) == 0
    return a.value, b.value


@pytest.mark.parametrize("k", [5, 8, 12, 14])
@pytest.mark.parametrize("n_mode", [0, 1])
def test_a_bad_residue_in_a_tiles_neighbour_chunk_counts_once(gpu_engine_cls, k, n_mode):
    """The last thread of a scatter workgroup stages the chunk BEHIND its tile (the neighbour of the tile's last chunk); the next tile
    stages it again as its chunk 0.  A residue outside ACGTN there is one error, listed once (it was deferred, and
    counted, twice).  8176-position tiles (k >= 9) and the 16 368-position tiles of the k <= 8 kernel: every chunk border from 500 to 1030."""
    from kmerdb_amd import synth
    bases, offsets = synth.reads(1, 40000, seed=5)
    for chunk in (510, 511, 512, 1022, 1023, 1024):
        for i in (0, 7, 15):
            b = bases.copy()
            b[16 * chunk + i] = ord("R")
            with gpu_engine_cls(k, n_mode=n_mode, algo=2) as eng:
                eng.submit(b, offsets)
                with pytest.raises(ValueError, match="outside ACGTN"):
                    eng.sync()
                assert _error_counts(eng) == (0, 1), (chunk, i)


def test_the_arena_leaves_reserved_memory_free(gpu_engine_cls, oracle):
    """Engine option "reserve_bytes" (VERDICT round 4, item 4): the page arena of k >= 14, which sizes itself on 85 % of the free device
    memory, never grows into the room a later allocation needs (RCCL's buffers and the reduce's scratch, kmerdb_amd/distributed.py).  With
    all but ~7 GiB of the device taken and 4 GiB reserved, the budget is what is left, the arena stays within it over many batches with
    arena_grow=2, and the counts are the oracle's."""
    import torch
    from kmerdb_amd import synth
    k = 14
    bases, offsets = synth.reads(60000, 150, seed=321)
    want_ids = np.concatenate([oracle.c_shred(bytes(bases[int(offsets[r]):int(offsets[r + 1])]).decode(), k, True, oracle.N_DROP)[0] for r in range(2000)])
    d_b = torch.from_numpy(bases).cuda()
    d_o = torch.from_numpy(offsets.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    with gpu_engine_cls(k) as eng:
        free, _ = torch.cuda.mem_get_info()
        hog = torch.empty(max(free - (7 << 30), 1 << 20), dtype=torch.uint8, device="cuda")
        try:
            reserve = 4 << 30
            eng.set_option("reserve_bytes", reserve)
            eng.set_option("arena_grow", 2)
            eng.set_option("arena_batches", 1)
            assert eng.get_option("reserve_bytes") == reserve
            n = 40
            for _ in range(n):
                eng.submit_device(d_b.data_ptr(), bases.size, d_o.data_ptr(), len(offsets) - 1)
            eng.sync()
            sized_on, budget = eng.get_option("free_at_sizing"), eng.get_option("arena_budget_bytes")
            assert sized_on > 0 and budget + reserve <= max(sized_on, reserve + (1 << 30))
            arena_bytes = eng.get_option("arena_pages") * (1024 + 4 + 8)            # pages + tags + list entries
            assert arena_bytes <= budget + (64 << 20), (arena_bytes, budget)
            assert eng.get_option("free_hbm") >= reserve - (512 << 20)              # (the reserve is still there for whoever comes next)
            _, total, _ = eng.finish(copy=False)
            assert total == n * 60000 * (150 - k + 1)
            tab = eng.table_tensor()
            u, c = np.unique(want_ids, return_counts=True)
            got = tab[torch.from_numpy(u.astype(np.int64)).cuda()].cpu().numpy()
            assert np.all(got >= (c * n)) and int(tab.sum().item()) == total
        finally:
            del hog
            torch.cuda.empty_cache()


@pytest.mark.gpu
def test_line_width_options_give_the_same_vector_and_engine_opts_come_from_the_environment(gpu_engine_cls, oracle, monkeypatch):
    """Round 5: the scatter kernels write 128-byte pieces by default (sc_wide_lines / l1_wide_lines / l2_wide_lines, l1_one_round); the 64-byte-line
    forms stay behind the options.  Every combination counts the same vector, and KDB_ENGINE_OPTS sets options for every engine a process creates
    (an unknown name fails the creation, as kdb_set_option would)."""
    from kmerdb_amd import synth
    bases, offsets = synth.reads(30000, 150, seed=4242)
    for k, names in ((12, ("sc_wide_lines",)), (15, ("l1_wide_lines", "l2_wide_lines", "l1_one_round")), (17, ("l1_wide_lines", "l2_wide_lines"))):
        want = None
        for mask in (range(1 << len(names)) if k < 17 else (0, 3)):          # (k = 17: a 128 GiB vector per engine -- all off, all on)
            with gpu_engine_cls(k, algo=2) as eng:
                for i, n in enumerate(names):
                    eng.set_option(n, (mask >> i) & 1)
                    assert eng.get_option(n) == (mask >> i) & 1
                eng.submit(bases, offsets)
                _, total, unique = eng.finish(copy=False)
                t = eng.table_tensor()
                got = (total, unique, _table_checksum(t))
            if want is None:
                want = got
                assert total == 30000 * (151 - k)
            assert got == want, (k, names, mask)
    monkeypatch.setenv("KDB_ENGINE_OPTS", "sc_wide_lines=0,l2_wide_lines=0")
    with gpu_engine_cls(12, algo=2) as eng:
        assert eng.get_option("sc_wide_lines") == 0 and eng.get_option("l2_wide_lines") == 0 and eng.get_option("l1_wide_lines") == 1
    monkeypatch.setenv("KDB_ENGINE_OPTS", "no_such_option=1")
    with pytest.raises(ValueError):
        gpu_engine_cls(12)
    monkeypatch.delenv("KDB_ENGINE_OPTS")


@pytest.mark.gpu
def test_hbm_pattern_probe_reports_every_pattern(gpu_engine_cls):
    """kdb_hbm_pattern_probe (bench.py: roofline.pattern_ceilings): every pattern gets a plausible rate, and the finding the 128-byte pieces rest on holds
    on this box too -- random 128-byte pieces are written faster than random 64-byte lines."""
    import ctypes
    import kmerdb_amd
    L = kmerdb_amd._abi.lib()
    n = L.kdb_hbm_pattern_count()
    names = [L.kdb_hbm_pattern_name(i).decode() for i in range(n)]
    assert n >= 8 and len(set(names)) == n and "pieces_128_write" in names and "lines_64_write" in names
    out = (ctypes.c_double * n)()
    kmerdb_amd._abi.check(L.kdb_hbm_pattern_probe(0, out, n))
    gbs = dict(zip(names, out))
    assert all(500.0 < v < 12000.0 for v in gbs.values()), gbs
    assert gbs["pieces_128_write"] > 1.05 * gbs["lines_64_write"], gbs
    assert gbs["scatter_128"] > gbs["scatter_64"] and gbs["level2_128"] > gbs["level2_64"], gbs
    with pytest.raises(ValueError):
        kmerdb_amd._abi.check(L.kdb_hbm_pattern_probe(0, out, n - 1))
