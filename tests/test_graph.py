"""De Bruijn edge list: oracle vs the reference's rows (CPU), GPU path vs both (-m gpu)."""
import json
import os
from collections import Counter

import numpy as np
import pytest


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "graph_edges.json")))


def test_oracle_rows_equal_reference_rows(oracle, golden_dir):
    cases = _cases(golden_dir)
    assert len(cases) == 12 and all(c["raises"] is None for c in cases)
    for c in cases:
        recs = list(oracle.read_records(os.path.join(golden_dir, c["file"])))
        rows = oracle.py_make_edges(recs, c["k"], canonicalize=c["canonicalize"])
        assert [list(r) for r in rows] == c["rows"], (c["file"], c["k"], c["canonicalize"])
        # the forward (k+1)-mer histogram (C oracle) carries the same multiset of edges
        bases, offsets = oracle.pack_records([s for _, s in recs])
        ev, n_edges = oracle.c_count_edges(bases, offsets, c["k"])
        assert n_edges == len(rows)
        if not c["canonicalize"]:
            want = Counter((r[2] << 2) | (r[4] & 3) for r in c["rows"])
            got = {int(i): int(ev[i]) for i in np.flatnonzero(ev)}
            assert got == dict(want)


@pytest.mark.gpu
def test_gpu_rows_metadata_counts_equal_reference(gpu_engine_cls, golden_dir):
    from kmerdb_amd import graph
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        for c in _cases(golden_dir):
            rows, meta, counts = graph.make_edges_from_fasta(c["file"], c["k"], quiet=True, canonicalize=c["canonicalize"])
            assert [list(r) for r in rows] == c["rows"], (c["file"], c["k"], c["canonicalize"])
            assert meta == c["metadata"]
            assert {int(i): int(counts[i]) for i in np.flatnonzero(counts)} == {int(a): b for a, b in c["counts_nonzero"].items()}
    finally:
        os.chdir(cwd)


@pytest.mark.gpu
def test_gpu_weighted_edges_equal_aggregated_reference_rows(gpu_engine_cls, golden_dir):
    from kmerdb_amd import graph
    for c in _cases(golden_dir):
        ev, counts, n_edges = graph.edge_counts(os.path.join(golden_dir, c["file"]), c["k"], canonicalize=c["canonicalize"])
        assert n_edges == len(c["rows"]) == int(ev.sum())
        id1, id2, w = graph.weighted_edges(ev, c["k"], canonicalize=c["canonicalize"])
        want = Counter((r[2], r[4]) for r in c["rows"])
        assert {(int(a), int(b)): int(x) for a, b, x in zip(id1, id2, w)} == dict(want)


@pytest.mark.gpu
def test_gpu_edge_histogram_k12_vs_oracle(gpu_engine_cls, oracle, tmp_path):
    """BASELINE config 5 shape (k=12 -> 13-mer histogram, 512 MiB vector) on 20 k reads, both algorithms."""
    from kmerdb_amd import graph, synth
    bases, offsets = synth.reads(20000, 150, seed=synth.SEED0 + 5)
    path = str(tmp_path / "g.fq")
    open(path, "wb").write(synth.fastq_text(bases, offsets))
    want, n_edges = oracle.c_count_edges(bases, offsets, 12)
    for algo in (1, 2):
        ev, counts, n = graph.edge_counts(path, 12, canonicalize=True, engine_opts={"algo": algo})
        assert n == n_edges == 20000 * 138 and np.array_equal(ev, want)
    wc, _ = oracle.c_count(bases, offsets, 12, True, oracle.N_DROP)
    assert np.array_equal(counts, wc)


@pytest.mark.gpu
def test_gpu_graph_errors(gpu_engine_cls, golden_dir):
    from kmerdb_amd import graph
    with pytest.raises(ValueError):
        graph.make_edges_from_fasta(os.path.join(golden_dir, "inputs/ragged_n.fq"), 4)      # N present
    with pytest.raises(TypeError):
        graph.make_edges_from_fasta(None, 4)
    with pytest.raises(TypeError):
        graph.make_edges_from_fasta(os.path.join(golden_dir, "inputs/graph_small.fa"), "4")


def _bgzf_blocks(raw):
    import struct
    sizes, p = [], 0
    while p < len(raw):
        bsize = struct.unpack("<H", raw[p + 16:p + 18])[0] + 1
        sizes.append(struct.unpack("<I", raw[p + bsize - 4:p + bsize])[0])
        p += bsize
    return sizes


def _check_kdbg(path, want):
    import gzip
    import hashlib
    raw = open(path, "rb").read()
    text = gzip.decompress(raw)
    assert text[:1500].decode("latin-1") == want["head"] and text[-300:].decode("latin-1") == want["tail"]
    assert hashlib.sha256(text).hexdigest() == want["sha256_decompressed"]
    assert _bgzf_blocks(raw) == want["block_sizes"]


def test_kdbg_writer_equals_reference_writer_output(golden_dir, tmp_path):
    """write_kdbg on the reference's own rows / metadata -> the stream and block boundaries the reference's KDBGWriter
    produced (tests/golden/graph_kdbg.json, generated through kmerdb/graph.py:376-474 and kmer.id_to_kmer)."""
    from collections import OrderedDict
    from kmerdb_amd import fileutil, graph
    edges = {(c["file"], c["k"], c["canonicalize"]): c for c in _cases(golden_dir)}
    done = 0
    for want in json.load(open(os.path.join(golden_dir, "graph_kdbg.json"))):
        c = edges.get((want["file"], want["k"], want["canonicalize"]))
        if c is None:
            continue                        # (the k = 12 case has no stored rows: covered by the GPU test below)
        N = 4 ** c["k"]
        uniq = len(c["counts_nonzero"])
        md = OrderedDict({"version": fileutil.VERSION, "metadata_blocks": 1, "k": c["k"], "total_kmers": c["metadata"]["total_kmers"],
                          "unique_kmers": uniq, "unique_nullomers": N - uniq if not c["canonicalize"] else int((N / 2) - uniq),
                          "sorted": False, "tags": [], "files": [c["metadata"]]})
        out = str(tmp_path / "g{0}.kdbg".format(done))
        graph.write_kdbg(out, md, [tuple(r) for r in c["rows"]], c["k"])
        _check_kdbg(out, want)
        done += 1
    assert done == 2
    with pytest.raises(IOError):
        graph.write_kdbg(str(tmp_path / "g.txt"), md, [], 4)
    with pytest.raises(TypeError):
        graph.write_kdbg(str(tmp_path / "g.kdbg"), None, [], 4)


@pytest.mark.gpu
def test_gpu_make_graph_writes_the_reference_kdbg(gpu_engine_cls, golden_dir, tmp_path):
    from kmerdb_amd import graph
    cwd = os.getcwd()
    os.chdir(golden_dir)                    # the golden header holds the relative filename
    try:
        for i, want in enumerate(json.load(open(os.path.join(golden_dir, "graph_kdbg.json")))):
            out = str(tmp_path / "m{0}.kdbg".format(i))
            md, n = graph.make_graph([want["file"]], want["k"], out, do_not_canonicalize=not want["canonicalize"])
            assert n == want["n_rows"]
            _check_kdbg(out, want)
    finally:
        os.chdir(cwd)
