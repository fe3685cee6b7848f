"""kmerdb_amd.kmer -- host-side mirror of the reference's kmerdb/kmer.py surface
used around the hot path: kmer_to_id (kmer.py:234-317), id_to_kmer (kmer.py:320-363)
and shred (kmer.py:489-577).

kmer_to_id / id_to_kmer on ONE k-mer string are plain host integer arithmetic
(callers such as minimizer.py:79 or codons.py:208 pass single strings); shred on
a whole record runs on the GPU through libkdbhip (kdb_shred) and raises if the
engine is unavailable.  Nothing here imports oracle/.
"""
from . import _abi
from .engine import ids_engine

letterToBinaryNA = {65: 0, 67: 1, 71: 2, 84: 3}       # kmer.py:44-49
binaryToLetterNA = ["A", "C", "G", "T"]
standard_lettersNA = set("ACTG")
_RC = str.maketrans("ACGT", "TGCA")


def kmer_to_id(s, is_aa=False, canonicalize=True):
    """kmer.py:234-317. str k-mer -> int id (min of forward / reverse-complement id when
    canonicalize); None if the k-mer contains N; TypeError for non-str; ValueError for
    residues outside ACGTN (the reference raises KeyError/NameError there)."""
    if type(s) is not str:
        raise TypeError("kmerdb_amd.kmer.kmer_to_id expects a str as its positional argument")
    if is_aa:
        raise NotImplementedError("amino-acid k-mers are outside the profile hot path")
    if s.find("N") != -1:
        return None
    idx1 = 0
    idx2 = 0
    try:
        for c in bytes(s, "UTF-8"):
            idx1 = (idx1 << 2) | letterToBinaryNA[c]
        for c in bytes(s.translate(_RC)[::-1], "UTF-8"):
            idx2 = (idx2 << 2) | letterToBinaryNA[c]
    except KeyError as e:
        raise ValueError("kmer_to_id: residue outside ACGTN in '{0}'".format(s)) from e
    return min(idx1, idx2) if canonicalize is True else idx1


def id_to_kmer(id, k, is_aa=False):
    """kmer.py:320-363 (nucleic acids)."""
    if type(id) is not int:
        raise TypeError("kmerdb_amd.kmer.id_to_kmer expects an int as its first positional argument")
    elif type(k) is not int:
        raise TypeError("kmerdb_amd.kmer.id_to_kmer expects an int as its second positional argument")
    if is_aa:
        raise NotImplementedError("amino-acid k-mers are outside the profile hot path")
    kmer = []
    for _ in range(k):
        kmer.append(binaryToLetterNA[id & 0x03])
        id = id >> 2
    kmer.reverse()
    return "".join(kmer)


def shred(seqRecord, k, replace_with_none=False, canonicalize=True, quiet_iupac_warning=True, device=0):
    """kmer.py:489-577 on the GPU: -> (kmer_ids, seq_ids, positions) python lists.

    `seqRecord` is a str or any object with .seq / .id (e.g. Bio.SeqRecord).  Windows
    containing N are dropped (replace_with_none=True) -- with replace_with_none=False
    the reference emits all 4^m fills of such windows (kmer.py:545-565); that expansion
    is done here on the host from the device's ids only for the N windows."""
    if type(k) is not int:
        raise TypeError("kmerdb_amd.kmer.shred() expects an int as its second positional argument")
    if isinstance(seqRecord, str):
        seq, seq_id = seqRecord, "Untitled_sequence"
    elif hasattr(seqRecord, "seq"):
        seq, seq_id = str(seqRecord.seq), getattr(seqRecord, "id", "Untitled_sequence")
    else:
        raise TypeError("kmerdb_amd.kmer.shred() expects a str or SeqRecord as its first positional argument")
    ids, pos = ids_engine(k, canonicalize is True, device).shred(seq)      # no 4^k vector is allocated for this
    ids, pos = ids.tolist(), pos.tolist()
    if replace_with_none is False and "N" in seq:
        ids, pos = _merge_n_expansions(seq, k, canonicalize, ids, pos)
    return ids, [seq_id] * len(ids), pos


def _merge_n_expansions(seq, k, canonicalize, ids, pos):
    """Insert, in window order, the 4^m fills of every window containing N (kmer.py:559-565)."""
    from itertools import product
    clean = dict(zip(pos, ids))
    out_ids, out_pos = [], []
    for i in range(len(seq) - k + 1):
        if i in clean:
            out_ids.append(clean[i])
            out_pos.append(i)
            continue
        w = seq[i:i + k]
        m = w.count("N")
        if m == 0:
            continue
        for fill in product("ACGT", repeat=m):
            f = w
            for c in fill:
                f = f.replace("N", c, 1)
            out_ids.append(kmer_to_id(f, canonicalize=canonicalize))
            out_pos.append(i)
    return out_ids, out_pos
