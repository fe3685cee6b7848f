"""CPU: the host feed (kmerdb/parse.py:50-85 on this path) -- streamed FASTA with pieces and statistics, BGZF inflate."""
import gzip
import os

import numpy as np
import pytest


def _mkfa(path, lens, wrap, rng):
    recs, out = [], []
    for i, L in enumerate(lens):
        s = "".join(rng.choice(list("ACGT"), size=L)) if L else ""
        recs.append(s)
        if wrap:
            out.append(f">c{i} d\n" + "\n".join(s[j:j + wrap] for j in range(0, L, wrap)) + ("\n" if L else ""))
        else:
            out.append(f">c{i} d\n{s}\n")
    open(path, "w").write("junk before the first header\n" + "".join(out))
    return recs


def _join(blocks, ov):
    """re-assemble the records from (possibly continued) blocks: a continuation piece repeats the last min(ov, so far) residues"""
    recs, prefix = [], []
    for blk in blocks:
        b, o, _ = blk
        o = o.astype(np.int64)
        for r in range(len(o) - 1):
            seq = bytes(b[o[r]:o[r + 1]]).decode()
            if r == 0 and blk.cont:
                p = min(ov, len(recs[-1]))
                assert p == 0 or recs[-1][-p:] == seq[:p]
                assert len(seq) > p                       # a piece always brings something new
                prefix.append(p)
                recs[-1] += seq[p:]
            else:
                assert not (r == 0 and blk.cont)
                recs.append(seq)
    return recs, prefix


@pytest.mark.parametrize("wrap", [60, 0])
def test_fasta_streaming_reassembles_every_record(tmp_path, wrap):
    from kmerdb_amd import reader
    rng = np.random.Generator(np.random.PCG64(2))
    for lens in ([10, 5000, 3, 0, 70000, 12, 1], [200000], [1, 1, 1], [9000] * 5):
        p = str(tmp_path / "s.fa")
        recs = _mkfa(p, lens, wrap, rng)
        for B in (1000, 4096, 65536, 1 << 20):
            for ov in (0, 11, 16):
                rd = reader.BlockReader(p, want_ids=True, block_bytes=B, overlap=ov)
                got, prefix = _join(rd, ov)
                assert got == recs, (wrap, lens, B, ov)
                L = [len(r) for r in recs]
                assert (rd.total_reads, rd.min_len, rd.max_len, rd.sum_len) == (len(L), min(L), max(L), sum(L)), (wrap, lens, B, ov)
    # whole-file mode (overlap=None) reports the same statistics
    rd = reader.BlockReader(p, want_ids=True)
    got, _ = _join(rd, 0)
    assert got == recs and rd.total_reads == len(recs)


def test_fastq_reader_statistics(golden_dir):
    from kmerdb_amd import reader
    for f in ("inputs/ragged_n.fq", "inputs/reads150.fq.gz"):
        rd = reader.BlockReader(os.path.join(golden_dir, f), block_bytes=3000)
        lens = []
        for b, o, _ in rd:
            lens += np.diff(o.astype(np.int64)).tolist()
        assert (rd.total_reads, rd.min_len, rd.max_len, rd.sum_len) == (len(lens), min(lens), max(lens), sum(lens))


def test_bgzf_block_parallel_inflate(tmp_path):
    from kmerdb_amd import fileutil, reader
    rng = np.random.Generator(np.random.PCG64(3))
    data = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes() + b"".join(
        b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), size=150).tolist()), b"I" * 150) for i in range(30000))
    p = str(tmp_path / "t.fq.gz")
    with open(p, "wb") as f:
        for i in range(0, len(data), 65280):
            f.write(fileutil._bgzf_member(data[i:i + 65280]))
        f.write(fileutil._bgzf_member(b""))                    # the BGZF EOF marker (an empty member)
    assert reader.is_bgzf(p)
    f = reader._open(p)
    assert isinstance(f, reader._BgzfFile)
    parts = []
    while True:
        x = f.read(1_000_003)
        if not x:
            break
        parts.append(x)
    assert b"".join(parts) == data == gzip.open(p, "rb").read()
    assert reader._BgzfFile(p).read() == data
    raw = bytearray(open(p, "rb").read())
    raw[5000] ^= 0xFF
    bad = str(tmp_path / "bad.fq.gz")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError):
        reader._BgzfFile(bad).read()
    # a plain gzip file is not BGZF: ordinary gzip.open is used
    g = str(tmp_path / "plain.fq.gz")
    gzip.open(g, "wb").write(data)
    assert not reader.is_bgzf(g) and not isinstance(reader._open(g), reader._BgzfFile)


def test_sharded_reader_inflates_only_its_own_bgzf_members(tmp_path):
    """Multi-GPU reading of a BGZF file: every rank finds the members that hold its blocks from the members'
    headers (kdb_bgzf_scan) and inflates those -- together the ranks inflate the file about once, not once per rank --
    and the records are partitioned exactly as for the plain file (the reference reads one stream: parse.py:64-72)."""
    from kmerdb_amd import fileutil, reader
    rng = np.random.Generator(np.random.PCG64(11))
    recs = []
    for i in range(20000):
        n = int(rng.integers(30, 200))
        recs.append(b"@r%d x\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), size=n).tolist()), bytes(rng.choice(list(b"@+I#"), size=n).tolist())))
    data = b"".join(recs)
    p, plain = str(tmp_path / "t.fq.gz"), str(tmp_path / "t.fq")
    open(plain, "wb").write(data)
    with open(p, "wb") as f:
        for i in range(0, len(data), 65280):
            f.write(fileutil._bgzf_member(data[i:i + 65280]))
        f.write(fileutil._bgzf_member(b""))
    src = reader._forward_source(p)
    assert isinstance(src, reader._BgzfShardSource) and src.nmem == (len(data) + 65279) // 65280 + 1
    assert int(src.uoff[-1]) == len(data) and int(src.coff[-1]) == os.path.getsize(p)
    src.close()

    def collect(path, rank, world, B):
        rd = reader.ShardedBlockReader(path, rank, world, want_ids=True, block_bytes=B)
        out = []
        for b, o, ids in rd:
            o = o.astype(np.int64)
            out += [(ids[r], bytes(b[o[r]:o[r + 1]])) for r in range(len(o) - 1)]
        return out, getattr(rd._src, "inflated", None)

    want, _ = collect(plain, 0, 1, 1 << 27)
    assert len(want) == 20000
    for world, B in ((1, 1 << 27), (2, 300000), (4, 100000), (8, 70001), (3, 5000)):
        got, inflated = [], 0
        for r in range(world):
            g, n = collect(p, r, world, B)
            got += g
            inflated += n
        assert sorted(got) == sorted(want), (world, B)
        # each rank inflates its blocks plus, per block, the members around its two ends -- not the whole stream
        nblocks = (len(data) + B - 1) // B
        assert inflated <= len(data) + nblocks * 4 * 65280 + world * (4 << 20), (world, B, inflated, len(data))
    # a corrupt member is an error on the rank that owns it
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    bad = str(tmp_path / "bad.fq.gz")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError):
        for r in range(2):
            collect(bad, r, 2, 200000)


def test_native_gzip_stream_reader(tmp_path):
    """One gzip stream is inflated by a native thread that runs ahead of the reader (kdb_gz_open): same bytes as gzip.open
    (what the reference uses, parse.py:64-72) for any read size, members concatenated; corrupt and truncated streams raise."""
    from kmerdb_amd import reader
    rng = np.random.Generator(np.random.PCG64(17))
    data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), size=120).tolist()), b"F" * 120) for i in range(40000))
    p = str(tmp_path / "t.fq.gz")
    with gzip.open(p, "wb", compresslevel=1) as f:
        f.write(data[:len(data) // 3])
    with gzip.open(p, "ab", compresslevel=9) as f:                 # a second member
        f.write(data[len(data) // 3:])
    f = reader._open(p)
    assert isinstance(f, reader._GzFile)
    assert f.read() == data and f.read(10) == b""
    for n in (1, 4097, 1 << 20, (4 << 20) + 3):
        g = reader._open(p)
        parts = []
        while True:
            x = g.read(n if n > 1 else 65537)
            if not x:
                break
            parts.append(x)
        g.close()
        assert b"".join(parts) == data, n
    # through the block reader: same records as from the plain file
    plain = str(tmp_path / "t.fq")
    open(plain, "wb").write(data)
    a = [(bytes(b), o.tolist()) for b, o, _ in reader.BlockReader(p, block_bytes=300000)]
    b = [(bytes(b), o.tolist()) for b, o, _ in reader.BlockReader(plain, block_bytes=300000)]
    assert a == b and len(a) > 10
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 4] ^= 0x55
    bad = str(tmp_path / "bad.fq.gz")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError):
        reader._open(bad).read()
    trunc = str(tmp_path / "trunc.fq.gz")
    open(trunc, "wb").write(bytes(raw[:50000]))
    with pytest.raises(ValueError):
        reader._open(trunc).read()
    with pytest.raises(ValueError):
        reader._GzFile(str(tmp_path / "missing.gz"))


def _fastq_general(text):
    """Bio.SeqIO.QualityIO.FastqGeneralIterator's grammar restated (what kmerdb/parse.py:70-72 reads FASTQ with): title line, sequence
    lines up to a '+' line, quality lines until they hold as many characters as the sequence.  -> [(id, seq)]; ValueError if malformed."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    out, i = [], 0
    while i < len(lines):
        if not lines[i] or lines[i][0] != "@":                   # (also a blank first line: blank lines behind a record are eaten by the quality loop below)
            raise ValueError("title")
        title = lines[i][1:].rstrip()
        i += 1
        seq = ""
        while True:
            if i >= len(lines):
                raise ValueError("eof")
            if lines[i][:1] == "+":
                second = lines[i][1:].rstrip()
                if second and second != title:
                    raise ValueError("captions")
                i += 1
                break
            seq += lines[i].rstrip()
            i += 1
        if " " in seq or "\t" in seq:
            raise ValueError("whitespace")
        if i >= len(lines):
            raise ValueError("eof")
        qual = lines[i].rstrip()
        i += 1
        while i < len(lines):
            if lines[i][:1] == "@" and len(qual) >= len(seq):
                break
            if not lines[i].strip() and len(qual) >= len(seq):
                i += 1
                continue
            qual += lines[i].rstrip()
            i += 1
        if len(qual) != len(seq) or any(not 33 <= ord(c) <= 126 for c in qual):
            raise ValueError("quality")
        out.append((title.split()[0] if title.split() else "", seq))
    return out


def test_wrapped_fastq_follows_biopythons_grammar(tmp_path):
    """a wrapped (multi-line) FASTQ raised 'third line does not start with +'; Bio.SeqIO.parse(handle, 'fastq') -- the
    reference's reader, kmerdb/parse.py:70-72 -- accepts it.  Random files of four-line and wrapped records, quality lines that start
    with '@', '+' lines that repeat the title, CR LF, blank lines between records; small blocks so that records straddle them; and the
    malformed shapes Biopython refuses, which must raise ValueError here too.  (A blank inside a four-line record's sequence is not
    the splitter's business: the blank reaches the counting kernels as a residue outside ACGTN and the job raises there.)"""
    from kmerdb_amd import reader
    rng = np.random.Generator(np.random.PCG64(23))

    def make(nrec, p_wrap):
        parts = []
        for r in range(nrec):
            L = int(rng.integers(1, 200))
            seq = "".join(rng.choice(list("ACGTN"), size=L))
            q = "".join(chr(c) for c in rng.integers(33, 127, size=L))
            if rng.integers(0, 3) == 0:
                q = "@" + q[1:]
            nl = "\r\n" if rng.integers(0, 10) == 0 else "\n"
            title = "r%d some description" % r
            if rng.random() < p_wrap:
                w = int(rng.integers(1, 70))
                plus = "+" + (title if rng.integers(0, 2) else "")
                parts.append("@" + title + nl + "".join(seq[i:i + w] + nl for i in range(0, L, w)) + plus + nl + "".join(q[i:i + w] + nl for i in range(0, L, w)))
            else:
                parts.append("@" + title + nl + seq + nl + "+" + nl + q + nl)
            if rng.integers(0, 25) == 0:
                parts.append("\n")
        return "".join(parts)

    for p_wrap, B in ((0.0, 1 << 20), (0.3, 1 << 20), (1.0, 5000), (0.05, 3000), (0.5, 70000)):
        text = make(1500, p_wrap)
        want = _fastq_general(text)
        p = str(tmp_path / "w.fq")
        open(p, "w", newline="").write(text)
        got = []
        for b, o, ids in reader.BlockReader(p, want_ids=True, block_bytes=B):
            o = o.astype(np.int64)
            got += [(ids[r], bytes(b[o[r]:o[r + 1]]).decode()) for r in range(len(o) - 1)]
        assert got == want, (p_wrap, B)
    good = "@a\nACGT\n+\nIIII\n"
    for bad in ("@a\nACGT\n+\nII I\n",                  # a blank in the quality string
                "@a\nACGT\n+\nIII\x7f\n",               # DEL
                "@a\nACGT\n+b\nIIII\n",                 # captions differ
                "@a\nACGT\n+\nIII\n",                   # lengths differ
                "@a\nACGT\n+\nIIIII\n",
                "@a\nACGT\nACGT\n",                     # no quality at all
                "@a\nACGTACGT\n+\nIIII\n",              # end of file inside the quality string
                "ACGT\n+\nIIII\n"):                     # no title
        for text in (bad, good + bad, good * 3 + bad + good) + ((("\n" + good),) if bad.startswith("ACGT\n+") else ()):      # (and, once: a blank line in front of the first record)
            with pytest.raises(ValueError):
                _fastq_general(text)
            p = str(tmp_path / "bad.fq")
            open(p, "w", newline="").write(text)
            with pytest.raises(ValueError):
                list(reader.BlockReader(p, want_ids=True))


def test_fastq_is_split_on_several_threads_like_on_one(tmp_path, monkeypatch):
    """kdb_parse_fastq_mt (the text cut at record starts, every piece counted, then split into its final place) gives the blocks the
    one-thread splitter gives: residues, offsets, ids -- for ragged records with '@' quality lines, CR LF, and a wrapped record in the
    middle (which sends that block through the general grammar on one thread)."""
    from kmerdb_amd import reader
    rng = np.random.Generator(np.random.PCG64(29))
    recs = []
    for r in range(60000):
        L = int(rng.integers(20, 260))
        seq = bytes(rng.choice(list(b"ACGTN"), size=L).tolist())
        q = bytes(rng.choice(list(b"@+I#F"), size=L).tolist())
        recs.append(b"@r%d x\n%s\n+\n%s%s" % (r, seq, q, b"\r\n" if r % 97 == 0 else b"\n"))
    for wrapped in (False, True):
        if wrapped:
            recs[30000] = b"@w y\nACGT\nACGTAC\n+w y\nIIII\n@IIIII\n"
        p = str(tmp_path / "m.fq")
        open(p, "wb").write(b"".join(recs))
        res = {}
        for threads in (1, 5):
            monkeypatch.setattr(reader, "_split_threads", lambda t=threads: t)
            res[threads] = [(bytes(b), o.tolist(), ids) for b, o, ids in reader.BlockReader(p, want_ids=True, block_bytes=12 << 20)]
        assert res[1] == res[5] and sum(len(x[2]) for x in res[1]) == 60000
        assert res[1][0][2][:2] == ["r0", "r1"]


def test_truncated_and_empty_gz_files(tmp_path):
    """util.is_gz_file sniffs the magic bytes where the reference tries gzip.open().readline() (kmerdb/util.py:80-88): the same answer on
    gzip files, plain files and empty files that matter here, and a truncated or corrupt .gz -- which the reference lets fail with
    EOFError / zlib.error from readline or from the parser -- raises ValueError when it is read (never a silent short count)."""
    from kmerdb_amd import reader, util
    data = b"".join(b"@r%d\nACGTACGTACGTACGTTTGA\n+\nIIIIIIIIIIIIIIIIIIII\n" % i for i in range(20000))
    p = str(tmp_path / "t.fq.gz")
    with gzip.open(p, "wb") as f:
        f.write(data)
    raw = open(p, "rb").read()
    assert util.is_gz_file(p)
    plain = str(tmp_path / "t.fq")
    open(plain, "wb").write(data)
    assert not util.is_gz_file(plain)
    for cut in (len(raw) // 2, len(raw) - 5, 30):
        t = str(tmp_path / "cut.fq.gz")
        open(t, "wb").write(raw[:cut])
        assert util.is_gz_file(t)
        with pytest.raises(ValueError):
            for _ in reader.BlockReader(t):
                pass
    empty = str(tmp_path / "empty.fq")
    open(empty, "wb").close()
    assert not util.is_gz_file(empty)
    assert list(reader.BlockReader(empty)) == []
