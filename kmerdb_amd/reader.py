"""Host feeder: FASTA / FASTQ (plain or gzip) -> flat residue buffers for the engine.

Takes over kmerdb/parse.py:50-85 (parse_sequence_file over Bio.SeqIO): gzip is
sniffed by content, the format is chosen by filename suffix, record ids are the
first whitespace token of the header, FASTA lines are concatenated, case is
preserved.  Instead of one SeqRecord object per read it emits, per block,
    bases   uint8[nbytes]      residues of all records of the block, concatenated
    offsets uint64[nreads+1]   record r = bases[offsets[r]:offsets[r+1]]
which is exactly what kdb_submit() takes.  All splitting is vectorised numpy;
nothing here touches residues one at a time.
"""
import gzip
import os

import numpy as np

from . import util

BLOCK_BYTES = 128 << 20


def _open(path):
    return gzip.open(path, "rb") if util.is_gz_file(path) else open(path, "rb")


def _keep_ranges(data, starts, ends):
    """Concatenate data[starts[i]:ends[i]] for all i (vectorised)."""
    n = data.size
    delta = np.zeros(n + 1, dtype=np.int8)
    np.add.at(delta, starts, 1)
    np.add.at(delta, ends, -1)
    keep = np.cumsum(delta[:-1], dtype=np.int8) > 0
    return data[keep]


def _lines(data):
    """-> (starts, ends) of every line of a uint8 buffer; '\\r' before '\\n' is excluded."""
    nl = np.flatnonzero(data == 10)
    starts = np.empty(nl.size + 1, dtype=np.int64)
    starts[0] = 0
    starts[1:] = nl + 1
    ends = np.empty(nl.size + 1, dtype=np.int64)
    ends[:-1] = nl
    ends[-1] = data.size
    if starts[-1] >= data.size:          # buffer ends with '\n': no trailing partial line
        starts, ends = starts[:-1], ends[:-1]
    cr = (ends > starts) & (data[np.maximum(ends - 1, 0)] == 13)
    ends = ends - cr
    return starts, ends


def _ids(data, starts, ends):
    out = []
    for s, e in zip(starts.tolist(), ends.tolist()):
        toks = bytes(data[s + 1:e]).split()
        out.append(toks[0].decode("utf-8", "replace") if toks else "")
    return out


def parse_fastq_block(data, want_ids=False):
    """data: uint8 buffer holding whole 4-line FASTQ records -> (bases, offsets, ids|None)."""
    starts, ends = _lines(data)
    # drop blank lines at the end (Biopython tolerates trailing blank lines)
    while starts.size and ends[-1] == starts[-1]:
        starts, ends = starts[:-1], ends[:-1]
    if starts.size % 4 != 0:
        raise ValueError("FASTQ block does not hold a whole number of 4-line records")
    if starts.size == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64), ([] if want_ids else None)
    h = starts[0::4]
    p = starts[2::4]
    if not (np.all(data[h] == 64) and np.all(data[p] == 43)):   # '@' and '+'
        raise ValueError("FASTQ records must be 4 lines: @id / sequence / + / quality")
    s, e = starts[1::4], ends[1::4]
    lens = e - s
    if not np.array_equal(lens, ends[3::4] - starts[3::4]):
        raise ValueError("FASTQ sequence and quality lengths differ")
    offsets = np.zeros(lens.size + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = _keep_ranges(data, s, e)
    return bases, offsets, (_ids(data, h, ends[0::4]) if want_ids else None)


def parse_fasta(data, want_ids=False):
    """data: whole FASTA text as uint8 -> (bases, offsets, ids|None). Lines before the first '>' are ignored."""
    starts, ends = _lines(data)
    nonempty = ends > starts
    is_h = np.zeros(starts.size, dtype=bool)
    is_h[nonempty] = data[starts[nonempty]] == 62                  # '>'
    rec = np.cumsum(is_h) - 1                                      # record index of every line
    seq = (~is_h) & (rec >= 0)
    nrec = int(is_h.sum())
    if nrec == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64), ([] if want_ids else None)
    bases = _keep_ranges(data, starts[seq], ends[seq])
    # Biopython's FASTA parser strips spaces and '\r' inside sequence lines
    ws = (bases == 32) | (bases == 13) | (bases == 9)
    if ws.any():
        # recompute per-record lengths after stripping: count kept bytes per line
        keep_line_len = np.add.reduceat(np.concatenate([(~ws).astype(np.int64), [0]]),
                                        np.concatenate([[0], np.cumsum(ends[seq] - starts[seq])[:-1]]))
        keep_line_len = np.where(ends[seq] - starts[seq] > 0, keep_line_len, 0)
        lens = np.bincount(rec[seq], weights=keep_line_len, minlength=nrec).astype(np.int64)
        bases = bases[~ws]
    else:
        lens = np.bincount(rec[seq], weights=(ends[seq] - starts[seq]), minlength=nrec).astype(np.int64)
    offsets = np.zeros(nrec + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return bases, offsets, (_ids(data, starts[is_h], ends[is_h]) if want_ids else None)


def _fastq_cut(buf):
    """Largest prefix of `buf` (bytes) made of whole 4-line records; -> cut index."""
    nl = np.flatnonzero(np.frombuffer(buf, dtype=np.uint8) == 10)
    whole = (nl.size // 4) * 4
    return 0 if whole == 0 else int(nl[whole - 1]) + 1


def iter_blocks(path, want_ids=False, block_bytes=BLOCK_BYTES):
    """Yield (bases, offsets, ids|None) blocks of a FASTA/FASTQ file (parse.py:50-85)."""
    if type(path) is not str:
        raise TypeError("iter_blocks expects a fasta/fastq filepath as a str")
    if not os.path.exists(path) or not os.access(path, os.R_OK):
        raise ValueError("the filepath must be readable on the filesystem")       # parse.py:57-58
    if util.is_fasta(path):
        with _open(path) as f:
            data = np.frombuffer(f.read(), dtype=np.uint8)
        yield parse_fasta(data, want_ids)
    elif util.is_fastq(path):
        with _open(path) as f:
            carry = b""
            while True:
                chunk = f.read(block_bytes)
                if not chunk:
                    break
                buf = carry + chunk
                cut = _fastq_cut(buf)
                if cut:
                    yield parse_fastq_block(np.frombuffer(buf[:cut], dtype=np.uint8), want_ids)
                carry = buf[cut:]
            if carry.strip():
                yield parse_fastq_block(np.frombuffer(carry, dtype=np.uint8), want_ids)
    else:
        raise ValueError("Could not determine the format of file '{0}'".format(path))   # parse.py:74
