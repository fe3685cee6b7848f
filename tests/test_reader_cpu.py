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
