#!/usr/bin/env python3
"""Golden vectors for the De Bruijn edge list: runs the REFERENCE's
kmerdb/graph.py make_edges_from_fasta (unmodified, loaded by path) on small N-free inputs and records
the per-occurrence rows it returns.  Same stand-ins as make_golden.py plus a no-op `jsonschema`.
Run in the build container only:  python tests/golden/make_golden_graph.py"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFPKG = "/root/reference/kmerdb"
sys.path.insert(0, os.path.join(HERE, "bio_standin"))
pkg = types.ModuleType("kmerdb")
pkg.__path__ = [REFPKG]
sys.modules["kmerdb"] = pkg
mods = {}
for name in ("config", "util", "kmer", "parse", "appmap", "fileutil", "graph"):
    spec = importlib.util.spec_from_file_location(f"kmerdb.{name}", os.path.join(REFPKG, f"{name}.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"kmerdb.{name}"] = m
    setattr(pkg, name, m)
    spec.loader.exec_module(m)
    mods[name] = m
graph = mods["graph"]

import numpy as np  # noqa: E402
rng = np.random.Generator(np.random.PCG64(99))
L = np.array(list("ACGT"))
ind = os.path.join(HERE, "inputs")
recs = ["ACGTTGCAAC", "".join(L[rng.integers(0, 4, size=60)]), "".join(L[rng.integers(0, 4, size=23)]), "AAAAAAAAAAAA", "ACACACACACAC"]
with open(os.path.join(ind, "graph_small.fa"), "w") as f:
    for i, s in enumerate(recs):
        f.write(f">g{i}\n{s}\n")
recs2 = ["".join(L[rng.integers(0, 4, size=50)]) for _ in range(40)]
with open(os.path.join(ind, "graph_reads50.fq"), "w") as f:
    for i, s in enumerate(recs2):
        f.write(f"@e{i}\n{s}\n+\n{'I' * 50}\n")

out = []
for fname in ("inputs/graph_small.fa", "inputs/graph_reads50.fq"):
    for k in (3, 4, 6):
        for canon in (True, False):
            os.chdir(HERE)
            try:
                data, meta, counts = graph.make_edges_from_fasta(fname, k, quiet=True, canonicalize=canon, replace_with_none=False)
                entry = {"file": fname, "k": k, "canonicalize": canon, "raises": None,
                         "rows": [[r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4])] for r in data],
                         "counts_nonzero": {int(i): int(counts[i]) for i in np.flatnonzero(counts)},
                         "metadata": {kk: (int(v) if isinstance(v, (int, np.integer)) else v) for kk, v in meta.items()}}
            except BaseException as e:  # noqa: BLE001
                entry = {"file": fname, "k": k, "canonicalize": canon, "raises": type(e).__name__, "message": str(e)[:200]}
            out.append(entry)
            print(fname, k, canon, entry.get("raises"), len(entry.get("rows", [])))
json.dump(out, open(os.path.join(HERE, "graph_edges.json"), "w"))

# ---- the .kdbg file the reference's driver writes (kmerdb/__init__.py:1635-1788), through the reference's own
# KDBGWriter (graph.py:376-474) and kmer.id_to_kmer; expected = the decompressed stream and the block sizes
import gzip  # noqa: E402
import hashlib  # noqa: E402
import struct  # noqa: E402
import tempfile  # noqa: E402
from collections import OrderedDict  # noqa: E402

kmer = mods["kmer"]
config = mods["config"]


def bgzf_block_sizes(raw):
    sizes, p = [], 0
    while p < len(raw):
        bsize = struct.unpack("<H", raw[p + 16:p + 18])[0] + 1
        sizes.append(struct.unpack("<I", raw[p + bsize - 4:p + bsize])[0])
        p += bsize
    return sizes


kdbg = []
for fname, k, canon in (("inputs/graph_small.fa", 4, True), ("inputs/graph_reads50.fq", 6, False), ("inputs/graph_reads50.fq", 12, True)):
    os.chdir(HERE)
    data, f_meta, counts = graph.make_edges_from_fasta(fname, k, quiet=True, canonicalize=canon, replace_with_none=False)
    N = 4 ** k
    unique_kmers = int(np.count_nonzero(counts))
    metadata = OrderedDict({                                   # __init__.py:1713-1723
        "version": config.VERSION, "metadata_blocks": 1, "k": k,
        "total_kmers": f_meta["total_kmers"], "unique_kmers": unique_kmers,
        "unique_nullomers": N - unique_kmers if not canon else int((N / 2) - unique_kmers),
        "sorted": False, "tags": [], "files": [f_meta]})
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "g.kdbg")
        out_ = graph.open(path, mode="w", metadata=metadata)
        try:
            for i in range(len(data)):                         # __init__.py:1756-1762
                seq_id, pos1, kmerid1, pos2, kmerid2 = data[i]
                tupley = (i, seq_id, pos1, kmerid1, kmer.id_to_kmer(kmerid1, k), pos2, kmerid2, kmer.id_to_kmer(kmerid2, k))
                out_.write("\t".join(list(map(str, tupley))) + "\n")
        finally:
            out_._write_block(out_._buffer)
            out_._handle.flush()
            out_._handle.close()
        raw = open(path, "rb").read()
    text = gzip.decompress(raw)
    kdbg.append({"file": fname, "k": k, "canonicalize": canon, "n_rows": len(data), "block_sizes": bgzf_block_sizes(raw),
                 "sha256_decompressed": hashlib.sha256(text).hexdigest(),
                 "head": text[:1500].decode("latin-1"), "tail": text[-300:].decode("latin-1")})
    print("kdbg", fname, k, canon, len(data), len(text))
json.dump(kdbg, open(os.path.join(HERE, "graph_kdbg.json"), "w"))
