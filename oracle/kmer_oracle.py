"""oracle/kmer_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two CPU restatements of the reference's k-mer counting semantics
(MatthewRalston/kmerdb v0.9.6, paths relative to /root/reference/):

* pure-Python functions (``py_kmer_to_id``, ``py_shred``, ``py_count``) that
  follow ``kmerdb/kmer.py:234-317`` / ``:489-577`` / ``kmerdb/parse.py:117-137``
  statement by statement -- for small cases only;
* a ctypes binding of ``oracle/kmer_oracle.c`` (``c_count`` ...), the plain-C
  restatement used at sizes Python loops cannot reach and as bench.py's
  ``cpu_baseline`` ("port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  Nothing under ``kmerdb_amd/`` does.

Pinning (see tests/test_oracle_golden.py): both restatements reproduce the
reference's own fixture pair ``test/data/Cacetobutylicum_ATCC824.fasta.gz`` ->
``test/data/test_Cac_ATCC824.8.kdb`` (forward, k=8, every bin) and the vectors
in ``tests/golden/`` that were produced by the reference's own ``kmer.py`` /
``parse.py`` (script: ``tests/golden/make_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkmer_oracle.so")

N_DROP = 0      # replace_with_none=True   (kmer.py:541-544)
N_EXPAND = 1    # replace_with_none=False  (kmer.py:545-565)

OK, SHORT_READ, BAD_RESIDUE, BAD_ARG = 0, 1, 2, 3

# kmer.py:44-49
letterToBinaryNA = {65: 0, 67: 1, 71: 2, 84: 3}
_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


class OracleError(ValueError):
    def __init__(self, status, read_index):
        self.status = status
        self.read_index = read_index
        what = {SHORT_READ: "sequence shorter than k (kmer.py:461-463)",
                BAD_RESIDUE: "residue outside ACGTN (kmer.py:309 / :170) that no N shields (kmer.py:287-289)",
                BAD_ARG: "bad argument"}.get(status, "?")
        super().__init__(f"oracle: {what} in record {read_index}")


# ----------------------------------------------------------------------------
# pure-Python restatement (small inputs)
# ----------------------------------------------------------------------------

def py_kmer_to_id(s, canonicalize=True):
    """kmer.py:234-317 (nucleic-acid branch). Returns None if 'N' in s."""
    if type(s) is not str:
        raise TypeError("kmer_to_id expects a str")          # :274-276
    if s.find("N") != -1:                                     # :287-289
        return None
    idx1 = 0
    idx2 = 0
    for c in bytes(s, "UTF-8"):                               # :307-309
        idx1 = idx1 << 2
        idx1 = idx1 | letterToBinaryNA[c]                     # KeyError on non-ACGT, as the reference
    rc = "".join(_COMPLEMENT[c] for c in reversed(s))         # Bio.Seq.reverse_complement, :310
    for c in bytes(rc, "UTF-8"):                              # :310-312
        idx2 = idx2 << 2
        idx2 = idx2 | letterToBinaryNA[c]
    return min(idx1, idx2) if canonicalize else idx1          # :314-317


def py_id_to_kmer(id, k):
    """kmer.py:320-363 (nucleic-acid branch)."""
    kmer = ""
    for _ in range(k):
        kmer += "ACGT"[id & 0x03]
        id = id >> 2
    return kmer[::-1]


def py_shred(seq, k, replace_with_none=False, canonicalize=True):
    """kmer.py:489-577 for one record given as str. Returns (ids, positions)."""
    from itertools import product
    if len(seq) < k:                                          # :461-463
        raise ValueError("sequence shorter than k")
    # :519-521: letters outside the IUPAC nucleotide alphabet raise for the whole record (NameError :170 / ValueError :473); the ten
    # IUPAC codes besides N pass, and what becomes of them is decided window by window: kmer_to_id returns None for a window that
    # holds an N BEFORE it looks at any other letter (:287-289), so such a window is dropped with replace_with_none=True even if
    # it also holds an R; without an N the code meets letterToBinaryNA (KeyError :309).
    if set(seq) - set("ACGTN" + "RYSWKMBDHV"):
        raise ValueError("residue outside the IUPAC nucleotide alphabet")
    if not replace_with_none and set(seq) - set("ACGTN"):
        # replace_with_none=False hands a window with such a code to _substitute_na_doublets / _triplets (:545-555, :630-851),
        # which raise (KeyError / NameError) for every shape but one (each code at least twice in every window that holds it:
        # tests/golden/iupac_next_to_n.json, "ANRRNA"); this restatement raises for all of them
        raise ValueError("IUPAC code other than N with replace_with_none=False")
    ids, pos = [], []
    for i in range(len(seq) - k + 1):                         # :526
        kmer = seq[i:i + k]
        try:
            kmer_id = py_kmer_to_id(kmer, canonicalize=canonicalize)   # :528
        except KeyError:
            raise ValueError("IUPAC code other than N in a window without N (kmer.py:309)")
        if kmer_id is not None:                               # :537-540
            ids.append(kmer_id)
            pos.append(i)
            continue
        if replace_with_none:                                 # :541-544
            continue
        m = kmer.count("N")                                   # :559, :586-621
        for fill in product("ACGT", repeat=m):
            _kmer = kmer
            for c in fill:
                _kmer = _kmer.replace("N", c, 1)
            ids.append(py_kmer_to_id(_kmer, canonicalize=canonicalize))   # :562-565
            pos.append(i)
    return ids, pos


def py_count(records, k, replace_with_none=True, canonicalize=True):
    """parse.py:117-137 over an iterable of str records. Returns (counts, total_kmers)."""
    counts = np.zeros(4 ** k, dtype="uint64")                 # :120
    total_kmers = 0
    for seq in records:
        ids, _ = py_shred(seq, k, replace_with_none=replace_with_none, canonicalize=canonicalize)
        for kmer_id in ids:                                   # :133-136
            counts[kmer_id] += 1
            total_kmers += 1
    return counts, total_kmers


def py_make_edges(records, k, canonicalize=True):
    """graph.py:108-216 + :261-283 for N-free records [(seq_id, seq), ...]:
    rows (seq_id, j-1, id[j-1], j, id[j]) for j = 1..n-1 in read order."""
    rows = []
    for seq_id, seq in records:
        ids, pos = py_shred(seq, k, replace_with_none=False, canonicalize=canonicalize)
        for i in range(1, len(ids)):                     # graph.py:151-153 skips position 0; every other k-mer pairs with its predecessor
            rows.append((seq_id, pos[i - 1], ids[i - 1], pos[i], ids[i]))
    return rows


# ----------------------------------------------------------------------------
# C restatement (ctypes)
# ----------------------------------------------------------------------------

def build(force=False):
    """Compile oracle/kmer_oracle.c -> oracle/libkmer_oracle.so with gcc."""
    src = os.path.join(_HERE, "kmer_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra",
                           "-o", _LIB_PATH, src])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.kdbo_kmer_to_id.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u64p]
        L.kdbo_kmer_to_id.restype = ctypes.c_int
        L.kdbo_count.argtypes = [u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 u64p, u64p, u64p]
        L.kdbo_count.restype = ctypes.c_int
        L.kdbo_count_mt.argtypes = [u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, u64p, u64p, u64p]
        L.kdbo_count_mt.restype = ctypes.c_int
        L.kdbo_shred.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 u64p, u64p, ctypes.c_uint64, u64p]
        L.kdbo_shred.restype = ctypes.c_int
        L.kdbo_count_edges.argtypes = [u8p, u64p, ctypes.c_uint64, ctypes.c_int, u64p, u64p, u64p]
        L.kdbo_count_edges.restype = ctypes.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _u64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def pack_records(records):
    """list of str/bytes -> (bases uint8[], offsets uint64[n+1])."""
    bs = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in records]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return bases, offsets


def c_count(bases, offsets, k, canonicalize=True, n_mode=N_DROP, counts=None, nthreads=1):
    """parse.py:117-137 on (bases, offsets). Returns (counts uint64[4^k], total_kmers)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nreads = len(offsets) - 1
    if counts is None:
        counts = np.zeros(4 ** k, dtype=np.uint64)
    total = ctypes.c_uint64(0)
    err = ctypes.c_uint64(0)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    if nthreads > 1:
        rc = lib().kdbo_count_mt(_u8(bases), _u64(offsets), nreads, k, int(canonicalize), n_mode,
                                 nthreads, _u64(counts), ctypes.byref(total), ctypes.byref(err))
    else:
        rc = lib().kdbo_count(_u8(bases), _u64(offsets), nreads, k, int(canonicalize), n_mode,
                              _u64(counts), ctypes.byref(total), ctypes.byref(err))
    if rc != OK:
        raise OracleError(rc, err.value)
    return counts, total.value


def c_shred(seq, k, canonicalize=True, n_mode=N_EXPAND):
    """kmer.py:489-577 for one record (bytes/str). Returns (ids uint64[], positions uint64[])."""
    b = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
    arr = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, np.uint8)
    n = ctypes.c_uint64(0)
    ids = np.zeros(1, np.uint64)
    pos = np.zeros(1, np.uint64)
    rc = lib().kdbo_shred(_u8(arr), len(b), k, int(canonicalize), n_mode, _u64(ids), _u64(pos), 0,
                          ctypes.byref(n))
    if rc != OK:
        raise OracleError(rc, 0)
    ids = np.zeros(max(n.value, 1), np.uint64)
    pos = np.zeros(max(n.value, 1), np.uint64)
    lib().kdbo_shred(_u8(arr), len(b), k, int(canonicalize), n_mode, _u64(ids), _u64(pos), n.value,
                     ctypes.byref(n))
    return ids[:n.value], pos[:n.value]


def c_count_edges(bases, offsets, k, edge_counts=None):
    """graph.py:108-216 on N-free records, aggregated by forward (k+1)-mer id."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    if edge_counts is None:
        edge_counts = np.zeros(4 ** (k + 1), dtype=np.uint64)
    total = ctypes.c_uint64(0)
    err = ctypes.c_uint64(0)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    rc = lib().kdbo_count_edges(_u8(bases), _u64(offsets), len(offsets) - 1, k, _u64(edge_counts),
                                ctypes.byref(total), ctypes.byref(err))
    if rc != OK:
        raise OracleError(rc, err.value)
    return edge_counts, total.value


# ----------------------------------------------------------------------------
# host-side record reader for the oracle (parse.py:50-85 restated; plain/gzip,
# FASTA/FASTQ by suffix).  Separate from kmerdb_amd's reader on purpose.
# ----------------------------------------------------------------------------

def read_records(path):
    """Yield (id, seq) like parse.parse_sequence_file (parse.py:50-85)."""
    import gzip
    with open(path, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"                         # util.py:80-88 (content sniff)
    opener = gzip.open if gz else open
    fasta = path.endswith((".fna", ".fna.gz", ".fa.gz", ".fa", ".fasta", ".fasta.gz"))   # util.py:120-126
    fastq = path.endswith((".fastq", ".fastq.gz", ".fq.gz", ".fq"))                      # util.py:128-134
    if not (fasta or fastq):
        raise ValueError(f"Could not determine the format of file '{path}'")            # parse.py:74
    with opener(path, "rt", encoding="UTF-8") as h:
        if fasta:
            name, chunks = None, []
            for line in h:
                line = line.rstrip("\r\n")
                if line.startswith(">"):
                    if name is not None:
                        yield name, "".join(chunks)
                    name = line[1:].split()[0] if len(line) > 1 and line[1:].split() else ""
                    chunks = []
                elif name is not None:
                    chunks.append(line.strip())
            if name is not None:
                yield name, "".join(chunks)
        else:
            while True:
                head = h.readline()
                if not head:
                    break
                if not head.strip():
                    continue
                seq = h.readline().rstrip("\r\n")
                h.readline()
                h.readline()
                toks = head[1:].split()
                yield (toks[0] if toks else ""), seq
