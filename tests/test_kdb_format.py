"""The .kdb writer / reader against the reference's own fixture file. CPU for the format,
-m gpu for the profile driver end to end."""
import ctypes
import gzip
import os
import struct

import numpy as np
import pytest

from tests.test_oracle_golden import read_kdb_counts


def _members(path):
    """Split a BGZF file into (uncompressed_size, is_bgzf_header) per member."""
    raw = open(path, "rb").read()
    out, pos = [], 0
    while pos < len(raw):
        assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and raw[pos + 12:pos + 14] == b"BC"
        bsize = struct.unpack("<H", raw[pos + 16:pos + 18])[0] + 1
        isize = struct.unpack("<I", raw[pos + bsize - 4:pos + bsize])[0]
        out.append(isize)
        pos += bsize
    assert pos == len(raw)
    return out


def test_frequency_formatting_matches_python_repr():
    import kmerdb_amd
    L = kmerdb_amd._abi.lib()
    buf = ctypes.create_string_buffer(64)
    rng = np.random.Generator(np.random.PCG64(3))
    cases = [(0, 7), (7, 7), (1, 3), (572, 4132866), (187, 4132866), (1, 10 ** 15), (123456789, 1), (10 ** 16, 1), (10 ** 17 + 5, 3),
             (1, 10000), (1, 100000), (99999, 10 ** 9)]
    cases += [(int(a), int(b)) for a, b in zip(rng.integers(0, 10 ** 6, 3000), rng.integers(1, 10 ** 9, 3000))]
    for c, t in cases:
        assert L.kdb_format_frequency(c, t, buf, 64) == 0
        assert buf.value.decode() == str(np.float64(c) / np.float64(t)), (c, t)


@pytest.mark.parametrize("encoder,threads", [("rows", 4), ("rows", 1), ("zlib", 3)])
def test_writer_reproduces_reference_fixture_rows_and_blocks(tmp_path, golden_dir, encoder, threads):
    """Rows (incl. every frequency string) and the 65536-byte block structure equal the reference's own .kdb -- through the
    row-aware deflate encoder (the default) and through zlib."""
    from kmerdb_amd import fileutil
    fixture = os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb")
    _, counts = read_kdb_counts(fixture)
    with gzip.open(fixture, "rt") as f:
        ref_text = f.read()
    ref_body = ref_text.split(fileutil.header_delimiter, 1)[1]
    md = {"version": fileutil.VERSION, "metadata_blocks": 1, "k": 8, "total_kmers": 4132866, "unique_kmers": 64103,
          "unique_nullomers": 1433, "sorted": False, "tags": [],
          "files": [{"filename": "test/data/Cacetobutylicum_ATCC824.fasta.gz", "md5": "0a0f73e1c8b8285703e29279bafaabef",
                     "sha256": "f9081291b62ff3387f1ca6ee2484669c849ed1840fdf2dd9dc3a0c93e9e87951", "total_reads": 2,
                     "total_kmers": 4132866, "unique_kmers": 64103, "nullomers": 1433, "min_read_length": 192000,
                     "max_read_length": 3940880, "avg_read_length": 2066440}]}
    out = str(tmp_path / "x.8.kdb")
    fileutil.write_kdb(out, md, counts, nthreads=threads, encoder=encoder)
    with gzip.open(out, "rt") as f:
        text = f.read()
    header, body = text.split(fileutil.header_delimiter, 1)
    assert body == ref_body                                   # all 65536 rows, byte for byte
    ours, theirs = _members(out), _members(fixture)
    assert ours[1:] == theirs[1:]                             # row blocks: 65536, 65536, ..., last partial; no EOF marker
    assert all(x == 65536 for x in ours[1:-1]) and 0 < ours[-1] <= 65536
    k = fileutil.read_kdb(out)
    assert k.k == 8 and np.array_equal(k.counts, counts) and np.array_equal(k.kmer_ids, np.arange(65536, dtype=np.uint64))
    assert np.array_equal(k.file_frequencies, counts / np.float64(4132866))
    assert np.array_equal(k.frequencies, counts / np.float64(4 ** 8))          # what KDBReader._slurp computes: float(count) / N, fileutil.py:363
    assert k.metadata["files"][0]["sha256"] == md["files"][0]["sha256"]
    ref = fileutil.read_kdb(fixture)                          # the v0.8.15 fixture itself reads too
    assert np.array_equal(ref.counts, counts) and ref.metadata["total_kmers"] == 4132866


def test_row_aware_encoder_on_hard_vectors(tmp_path):
    """The row-aware encoder on vectors that leave its fast paths: counts beyond its 65536-entry string table, 20-digit counts,
    all-zero vectors, k = 1 (one short member), and a k = 9 vector whose text ends a few bytes into its last member.  Every
    file must gunzip (Python's zlib) to the rows Python itself formats, in members of exactly 65536 bytes."""
    from kmerdb_amd import fileutil
    rng = np.random.Generator(np.random.PCG64(11))
    cases = []
    for k in (1, 2, 5, 8, 9):
        n = 4 ** k
        cases.append((k, rng.integers(0, 200, n).astype(np.uint64)))
        cases.append((k, rng.integers(0, 2 ** 40, n).astype(np.uint64) * (rng.integers(0, 4, n) == 0)))
        big = rng.integers(60000, 70000, n).astype(np.uint64)
        big[rng.integers(0, n, max(1, n // 50))] = np.uint64(2 ** 63 + 12345)
        cases.append((k, big))
    z = np.zeros(4 ** 7, np.uint64)
    z[-1] = 1
    cases.append((7, z))
    for i, (k, counts) in enumerate(cases):
        total = int(counts.sum(dtype=object)) if counts.max() > 2 ** 40 else int(counts.sum())
        total = min(total, 2 ** 64 - 1) or 1
        md = {"version": fileutil.VERSION, "metadata_blocks": 1, "k": k, "total_kmers": total, "unique_kmers": int(np.count_nonzero(counts)),
              "unique_nullomers": 0, "sorted": False, "tags": [], "files": []}
        out = str(tmp_path / ("h%d.kdb" % i))
        nblocks = fileutil.write_kdb(out, md, counts, nthreads=1 + i % 5, encoder="rows")
        with gzip.open(out, "rt") as f:
            body = f.read().split(fileutil.header_delimiter, 1)[1]
        freqs = counts.astype(np.float64) / np.float64(total) if total < 2 ** 63 else np.array([np.float64(int(c)) / np.float64(total) for c in counts])
        want = "".join("{0}\t{0}\t{1}\t{2}\n".format(j, int(c), f) for j, (c, f) in enumerate(zip(counts, freqs)))
        assert body == want, (k, i)
        sizes = _members(out)[1:]
        assert len(sizes) == nblocks == -(-len(want) // 65536) and all(x == 65536 for x in sizes[:-1]) and sizes[-1] == len(want) - 65536 * (len(sizes) - 1)
    with pytest.raises(ValueError):
        fileutil.write_kdb(str(tmp_path / "e.kdb"), md, counts, encoder="lz4")


def test_native_reader_equals_the_gzip_and_pandas_reader(tmp_path, golden_dir):
    """read_kdb through kdb_read_kdb_rows (BGZF members inflated and parsed in parallel) == the same file read as one gzip stream through
    pandas: ids, counts, the file's frequency column -- for the reference's own fixture, for files of this writer (both encoders, rows
    that straddle members and member groups) and for 1 .. 7 threads; malformed rows are refused with ValueError by both."""
    from kmerdb_amd import fileutil, _abi
    rng = np.random.Generator(np.random.PCG64(5))

    def pandas_read(path):
        saved = _abi.lib
        _abi.lib = lambda: (_ for _ in ()).throw(_abi.KdbHipError("no native reader in this test"))
        try:
            return fileutil.read_kdb(path)
        finally:
            _abi.lib = saved

    paths = [os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb")]
    for i, (k, enc) in enumerate(((1, "rows"), (5, "zlib"), (9, "rows"), (10, "rows"))):
        counts = rng.integers(0, 3000, 4 ** k).astype(np.uint64) * (rng.integers(0, 3, 4 ** k) > 0)
        md = {"version": fileutil.VERSION, "metadata_blocks": 1, "k": k, "total_kmers": int(counts.sum()) or 1, "unique_kmers": int(np.count_nonzero(counts)),
              "unique_nullomers": 0, "sorted": False, "tags": [], "files": []}
        p = str(tmp_path / ("r%d.kdb" % i))
        fileutil.write_kdb(p, md, counts, nthreads=3, encoder=enc)
        paths.append(p)
    for p in paths:
        want = pandas_read(p)
        for threads in (1, 2, 7):
            got = fileutil.read_kdb(p, nthreads=threads)
            assert got.metadata == want.metadata and got.k == want.k
            for name in ("kmer_ids", "counts", "file_frequencies", "frequencies"):
                assert np.array_equal(getattr(got, name), getattr(want, name)), (p, threads, name)
    # malformed bodies: a missing column, a row index that is not the line number, a row too many
    good_rows = ["{0}\t{0}\t{1}\t{2}\n".format(i, i % 5, (i % 5) / 40.0) for i in range(16)]
    header = (yaml_header := "version: 0.9.6\nk: 2\nfiles: []\n") + fileutil.header_delimiter
    for bad in (good_rows[:7] + ["7\t7\t1\n"] + good_rows[8:], good_rows[:7] + ["9\t7\t1\t0.1\n"] + good_rows[8:], good_rows + ["16\t3\t1\t0.1\n"], good_rows[:15]):
        p = str(tmp_path / "bad.kdb")
        with open(p, "wb") as f:
            f.write(fileutil._bgzf_member(header.encode()))
            f.write(fileutil._bgzf_member("".join(bad).encode()))
        with pytest.raises(ValueError):
            fileutil.read_kdb(p)
        with pytest.raises(ValueError):
            pandas_read(p)
    p = str(tmp_path / "good.kdb")
    with open(p, "wb") as f:
        f.write(fileutil._bgzf_member(header.encode()))
        f.write(fileutil._bgzf_member("".join(good_rows[:9]).encode()[:-3]))           # a row cut in the middle of a member boundary
        f.write(fileutil._bgzf_member("".join(good_rows[:9]).encode()[-3:] + "".join(good_rows[9:]).encode()))
    assert np.array_equal(fileutil.read_kdb(p).counts, np.arange(16, dtype=np.uint64) % 5) and np.array_equal(pandas_read(p).counts, fileutil.read_kdb(p).counts)


def test_open_gives_the_references_reader_and_writer_objects(tmp_path, golden_dir):
    """fileutil.open (kmerdb/fileutil.py:46-105) with the reference's argument checks; the writer object runs _profile's own loop
    (kmerdb/__init__.py:1980-1998: write() per row, _write_block(_buffer) at the end, no close()) and leaves the fixture's rows in the
    fixture's members; the reader object has the attributes of KDBReader before and after slurp() (fileutil.py:229-241, :308-466)."""
    from kmerdb_amd import fileutil
    fixture = os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb")
    rd = fileutil.open(fixture, "r")
    assert rd.k == 8 and rd.metadata["total_kmers"] == 4132866 and not rd.completed and not rd.counts.any() and rd.counts.size == 65536
    counts = rd.slurp()
    assert rd.completed and counts is rd.counts and int(counts.sum()) == 4132866
    assert np.array_equal(rd.kmer_ids, np.arange(65536, dtype=np.uint64)) and np.array_equal(rd.frequencies, counts / np.float64(65536))
    rd2 = fileutil.open(fixture, mode="r", slurp=True)
    assert np.array_equal(rd2.counts, counts)
    md = dict(rd.metadata)
    out = str(tmp_path / "w.8.kdb")
    w = fileutil.open(out, "wb", metadata=md)
    freqs = counts / np.float64(4132866)
    for i in range(65536):
        w.write("{0}\t{1}\t{2}\t{3}\n".format(i, i, counts[i], freqs[i]))
    w._write_block(w._buffer)
    w._handle.flush()
    w._handle.close()
    with gzip.open(fixture, "rt") as f:
        ref_body = f.read().split(fileutil.header_delimiter, 1)[1]
    with gzip.open(out, "rt") as f:
        body = f.read().split(fileutil.header_delimiter, 1)[1]
    assert body == ref_body and _members(out)[1:] == _members(fixture)[1:]
    assert np.array_equal(fileutil.open(out, "r", slurp=True).counts, counts)
    for bad, exc in (((None,), TypeError), ((fixture, 3), TypeError), ((out, "w"), TypeError), ((fixture, "r", None, 1), TypeError),
                     ((fixture, "rz"), ValueError), ((fixture, "rr"), ValueError), ((fixture, "bt"), ValueError), ((fixture, "b"), ValueError),
                     ((str(tmp_path / "missing.kdb"), "r"), IOError)):
        with pytest.raises(exc):
            fileutil.open(*bad)
    with pytest.raises(TypeError):
        fileutil.KDBWriter(None, filename=out)


def test_reader_rejects_invalid_files(golden_dir, tmp_path):
    from kmerdb_amd import fileutil
    p = tmp_path / "bad.kdb"
    p.write_bytes(b"not a kdb\n")                             # reference test_fileutil.py:97-113 -> ValueError
    with pytest.raises(ValueError):
        fileutil.read_kdb(str(p))
    with pytest.raises(TypeError):
        fileutil.read_kdb(None)
    with pytest.raises(ValueError):
        fileutil.write_kdb(str(tmp_path / "y.kdb"), {"k": 3}, np.zeros(64, np.uint64))


@pytest.mark.gpu
def test_profile_driver_writes_the_reference_fixture(gpu_engine_cls, golden_dir, tmp_path):
    """`kmerdb profile -k 8 --do-not-canonicalize` on the Cac genome == the reference's test_Cac_ATCC824.8.kdb rows."""
    from kmerdb_amd import fileutil, profile
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        counts, md, out = profile.profile(["ref_data/Cacetobutylicum_ATCC824.fasta.gz"], 8, str(tmp_path / "cac"),
                                          do_not_canonicalize=True)
    finally:
        os.chdir(cwd)
    fixture = os.path.join(golden_dir, "ref_data", "test_Cac_ATCC824.8.kdb")
    with gzip.open(fixture, "rt") as f:
        ref_body = f.read().split(fileutil.header_delimiter, 1)[1]
    with gzip.open(out, "rt") as f:
        header, body = f.read().split(fileutil.header_delimiter, 1)
    assert body == ref_body
    assert md["total_kmers"] == 4132866 and md["unique_kmers"] == 64103 and md["unique_nullomers"] == 1433
    assert md["files"][0]["md5"] == "0a0f73e1c8b8285703e29279bafaabef"
    k = fileutil.read_kdb(out)
    assert np.array_equal(k.counts, counts)


@pytest.mark.gpu
@pytest.mark.parametrize("k,nfiles", [(9, 1), (13, 1), (12, 3)])
def test_copy_back_beside_the_row_writer(gpu_engine_cls, golden_dir, tmp_path, k, nfiles):
    """profile(write=True) copies the vector back in pieces while the row writer is already at work on the rows that have arrived
    (kdb_copy_back_and_write_kdb_rows): the counts it returns, the file it writes and the file read back are those of the plain path
    (finish + write_kdb); one file (the engine's own vector) and several (the samplesheet accumulator)."""
    from kmerdb_amd import fileutil, profile
    files = [os.path.join(golden_dir, "inputs", f) for f in ("reads150.fq", "contigs.fa", "reads150.fq.gz")][:nfiles]
    inp = files
    if nfiles > 1:
        sheet = tmp_path / "sheet.txt"
        sheet.write_text("\n".join(files) + "\n")
        inp = [str(sheet)]
    tm = {}
    counts, md, out = profile.profile(inp, k, str(tmp_path / "a"), no_ambiguous=True, timings=tm)
    assert "copy_back_and_write_kdb_s" in tm and md["total_kmers"] == int(counts.sum()) and md["unique_kmers"] == int(np.count_nonzero(counts))
    plain_counts, md2, _ = profile.profile(inp, k, str(tmp_path / "b"), no_ambiguous=True, write=False)
    assert np.array_equal(counts, plain_counts) and md2["total_kmers"] == md["total_kmers"]
    ref = str(tmp_path / "ref.kdb")
    fileutil.write_kdb(ref, dict(md), plain_counts)
    with gzip.open(out, "rb") as f, gzip.open(ref, "rb") as g:
        assert f.read() == g.read()
    assert _members(out) == _members(ref)
    assert np.array_equal(fileutil.read_kdb(out).counts, counts)


@pytest.mark.gpu
def test_profile_cli_and_samplesheet(gpu_engine_cls, golden_dir, tmp_path):
    from kmerdb_amd import fileutil, profile
    sheet = tmp_path / "inputs.txt"
    a = os.path.join(golden_dir, "inputs", "reads150.fq")
    b = os.path.join(golden_dir, "inputs", "contigs.fa")
    sheet.write_text(a + "\n" + b + "\n")
    rc = profile.main(["profile", "-k", "9", "-o", str(tmp_path / "two"), "--quiet", str(sheet)])
    assert rc == 0
    k = fileutil.read_kdb(str(tmp_path / "two.9.kdb"))
    from kmerdb_amd import parse
    ca, _, _ = parse.parsefile(a, 9, replace_with_none=False)
    cb, _, _ = parse.parsefile(b, 9, replace_with_none=False)
    assert np.array_equal(k.counts, ca + cb) and len(k.metadata["files"]) == 2
    with pytest.raises(ValueError):
        profile.profile([a, b], 9, str(tmp_path / "z"))        # reference: exactly one positional input
