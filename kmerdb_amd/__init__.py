"""kmerdb_amd -- MI355X-native k-mer counting engine behind kmerdb's `profile` hot path.

    from kmerdb_amd import parse
    counts, file_metadata, nullomers = parse.parsefile("reads.fq.gz", 12)

is a drop-in for kmerdb.parse.parsefile (reference kmerdb/parse.py:90); the work
runs in hand-written gfx950 HIP kernels behind the C ABI of include/kdbhip.h.
"""
VERSION = "0.1.0"

from . import _abi, util, reader, kmer, parse, synth, graph, fileutil, profile  # noqa: E402,F401
from .engine import Engine, KDB_N_DROP, KDB_N_EXPAND, device_count, pinned_empty  # noqa: E402,F401
