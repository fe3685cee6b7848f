"""kmerdb_amd.graph -- the k -> k+1 adjacency path of `kmerdb graph` (reference kmerdb/graph.py:108-332,
driver kmerdb/__init__.py:1635-1788) on the GPU.

For an N-free record with n = L-k+1 k-mers the reference emits, in read order, the rows
    (seq_id, j-1, id[j-1], j, id[j])     j = 1 .. n-1          (graph.py:108-216)
and also accumulates the k-mer count vector (graph.py:281-283).  The consecutive pair (id[j-1], id[j]) is
a function of the forward (k+1)-mer starting at j-1, so the WEIGHTED edge list is a dense 4^(k+1) histogram
of forward (k+1)-mers -- exactly the `profile` hot path at k+1 -- followed by a fold onto (id1, id2).

    make_edges_from_fasta(filename, k, ...)   drop-in: per-occurrence rows + metadata + counts
    edge_counts(filename, k)                  device histogram of (k+1)-mers (uint64[4^(k+1)]) + k-mer counts
    weighted_edges(edge_vector, k, canon)     fold to (id1, id2, weight) arrays
    write_kdbg / make_graph                   the .kdbg file the reference's `graph` driver writes

Records containing N: the reference raises ValueError with its default N-expansion (graph.py:351-354) and
emits a spurious gap-bridging edge with --replace-with-none; only N-free input is in scope,
so any N raises ValueError here.
"""
import os
from collections import OrderedDict

import numpy as np

from . import fileutil, reader, util
from .engine import Engine, KDB_N_DROP, NO_WINDOW


def _rc_ids(ids, k):
    """reverse-complement id of each k-mer id (vectorised numpy; host-side fold of the device histogram)."""
    x = ids.astype(np.uint64).copy()
    rc = np.zeros_like(x)
    for _ in range(k):
        rc = (rc << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return rc


def make_edges_from_fasta(filename, k, quiet=True, canonicalize=True, replace_with_none=False, device=0):
    """kmerdb/graph.py:219-332.  -> (rows, file_metadata, counts)
    rows: list of (seq_id, pos1, kmer_id1, pos2, kmer_id2) in the reference's order; counts: uint64[4**k]."""
    if type(filename) is not str:
        raise TypeError("kmerdb_amd.graph.make_edges_from_fasta() expects a fasta/fastq sequence filepath as a str")
    elif type(k) is not int:
        raise TypeError("kmerdb_amd.graph.make_edges_from_fasta() expects an int for k as the second positional argument")
    elif type(quiet) is not bool:
        raise TypeError("kmerdb_amd.graph.make_edges_from_fasta() expects the keyword argument 'quiet' to be a bool")
    if os.path.exists(filename) is False or os.access(filename, os.R_OK) is False:
        raise ValueError("kmerdb_amd.graph.make_edges_from_fasta() expects the filepath to be be readable on the filesystem")
    N = 4 ** k
    rows = []
    lens_all = []
    from .engine import ids_engine
    ids_eng = ids_engine(k, canonicalize is True, device)             # window ids need no 4^k vector
    with Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP, device=device) as eng:
        for bases, offsets, ids_ in reader.iter_blocks(filename, want_ids=True, block_bytes=32 << 20):
            if len(offsets) < 2:
                continue
            if np.any(bases == ord("N")):
                raise ValueError("kmerdb_amd.graph: records containing N are outside the edge-list path "
                                 "(the reference raises ValueError at graph.py:351-354)")
            wid = ids_eng.window_ids(bases, offsets)      # also raises on short records / bad residues
            eng.submit(bases, offsets)                    # the k-mer count vector, graph.py:281-283
            o = offsets.astype(np.int64)
            lens = np.diff(o)
            lens_all.append(lens)
            # rows (seq_id, j-1, id[j-1], j, id[j]) for j = 1 .. n-1 of every record, in read order: built as arrays
            # (one pass of numpy per block), turned into the reference's tuples at the end
            n = lens - (k - 1)                            # k-mers per record (>= 1: shorter records raised above)
            ne = n - 1                                    # rows per record
            tot = int(ne.sum())
            if tot:
                rec = np.repeat(np.arange(len(lens)), ne)
                first = np.cumsum(ne) - ne                # index of each record's first row
                j = np.arange(tot, dtype=np.int64) - np.repeat(first, ne) + 1
                p1 = o[rec] + j - 1                       # residue position of k-mer j-1
                id1, id2 = wid[p1], wid[p1 + 1]
                assert not (np.any(id1 == NO_WINDOW) or np.any(id2 == NO_WINDOW))
                sid = np.asarray(ids_, dtype=object)[rec]
                rows.extend(zip(sid.tolist(), (j - 1).tolist(), id1.tolist(), j.tolist(), id2.tolist()))
        if not lens_all:
            raise ValueError("no sequence records found in '{0}'".format(filename))
        counts, total_kmers, unique_kmers = eng.finish()
    lens = np.concatenate(lens_all)
    md5, sha256 = util.checksum(filename)
    file_metadata = {                                        # graph.py:305-330
        "filename": filename, "md5": md5, "sha256": sha256,
        "total_reads": int(len(lens)), "total_kmers": int(total_kmers), "unique_kmers": int(unique_kmers),
        "nullomers": int(N - unique_kmers) if canonicalize is False else int((N / 2) - unique_kmers),
        "num_reads": int(len(lens)),
        "min_read_length": int(lens.min()), "max_read_length": int(lens.max()), "avg_read_length": int(np.mean(lens)),
    }
    return rows, file_metadata, counts


def edge_counts(filename, k, canonicalize=True, device=0, engine_opts=None):
    """Weighted adjacency on the device: -> (edge_vector uint64[4**(k+1)] keyed by forward (k+1)-mer id,
    counts uint64[4**k], n_edges).  Both histograms run through the profile hot path (at k+1 and k)."""
    if type(filename) is not str:
        raise TypeError("edge_counts expects a filepath str")
    if type(k) is not int:
        raise TypeError("edge_counts expects an int k")
    with Engine(k + 1, canonicalize=False, n_mode=KDB_N_DROP, device=device) as e1, \
            Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP, device=device) as e0:
        e1.set_option("min_len", k)                # a record of exactly k residues has one k-mer and no edge
        for name, v in (engine_opts or {}).items():
            e1.set_option(name, v)
        any_rec = False
        for bases, offsets, _ in reader.iter_blocks(filename):
            if len(offsets) < 2:
                continue
            if np.any(bases == ord("N")):
                raise ValueError("kmerdb_amd.graph: records containing N are outside the edge-list path")
            any_rec = True
            e1.submit(bases, offsets)
            e0.submit(bases, offsets)
        if not any_rec:
            raise ValueError("no sequence records found in '{0}'".format(filename))
        edges, n_edges, _ = e1.finish()
        counts, _, _ = e0.finish()
    return edges, counts, n_edges


def weighted_edges(edge_vector, k, canonicalize=True):
    """Fold the forward (k+1)-mer histogram onto the reference's (kmer_id1, kmer_id2) pairs.
    -> (id1 uint64[], id2 uint64[], weight uint64[]) sorted by (id1, id2); Sum(weight) == number of rows."""
    e = np.flatnonzero(edge_vector).astype(np.uint64)
    w = edge_vector[e.astype(np.int64)]
    mask = np.uint64(4 ** k - 1)
    id1 = e >> np.uint64(2)
    id2 = e & mask
    if canonicalize:
        id1 = np.minimum(id1, _rc_ids(id1, k))
        id2 = np.minimum(id2, _rc_ids(id2, k))
        key = id1 * np.uint64(4 ** k) + id2
        uk, inv = np.unique(key, return_inverse=True)
        ww = np.zeros(len(uk), dtype=np.uint64)
        np.add.at(ww, inv, w)
        return uk // np.uint64(4 ** k), uk % np.uint64(4 ** k), ww
    return id1, id2, w


def _ids_to_kmers(ids, k):
    """kmer.id_to_kmer (kmer.py:320-363) for an array of ids -> list of str."""
    ids = np.asarray(ids, dtype=np.uint64)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((ids.size, k), dtype=np.uint8)
    for j in range(k):
        out[:, k - 1 - j] = letters[((ids >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
    return [row.tobytes().decode("ascii") for row in out]


def write_kdbg(path, metadata, rows, k, compresslevel=6):
    """The .kdbg file of `kmerdb graph` (driver kmerdb/__init__.py:1745-1768, writer graph.py:376-474): BGZF member(s)
    with the YAML header + delimiter, then the rows
        i \t seq_id \t pos1 \t kmer_id1 \t kmer1 \t pos2 \t kmer_id2 \t kmer2
    cut into 65536-byte BGZF members, the last one holding whatever remains (no EOF marker, like the reference).
    `rows` are the tuples make_edges_from_fasta returns.  -> number of row blocks written."""
    if type(path) is not str:
        raise TypeError("kmerdb_amd.graph.write_kdbg expects the filename to be a str")
    if os.path.splitext(path)[-1] != ".kdbg":
        raise IOError("Destination .kdbg filepath does not end in '.kdbg'")                 # __init__.py:1662-1663
    if metadata is None or (type(metadata) is not OrderedDict and type(metadata) is not dict):
        raise TypeError("kmerdb_amd.graph.write_kdbg - invalid metadata argument")           # graph.py:413-414
    for key in ("version", "metadata_blocks", "k", "tags", "files", "total_kmers", "unique_kmers", "unique_nullomers"):
        if key not in metadata:                                                              # config.graph_schema "required"
            raise ValueError("kdbg metadata is missing the key '{0}'".format(key))
    for f in metadata["files"]:
        for key in ("filename", "sha256", "md5", "total_reads", "total_kmers", "unique_kmers", "nullomers"):
            if key not in f:
                raise ValueError("kdbg file metadata is missing the key '{0}'".format(key))
    hb, nblocks = fileutil.header_bytes(metadata)                                            # same construction as KDBWriter
    nrow_blocks = 0
    with open(path, "wb") as f:
        for _ in range(nblocks):
            f.write(fileutil._bgzf_member(hb[:65536], compresslevel))
            hb = hb[65536:]
        buf = b""
        step = 1 << 16
        for s0 in range(0, len(rows), step):
            part = rows[s0:s0 + step]
            k1 = _ids_to_kmers([r[2] for r in part], k)
            k2 = _ids_to_kmers([r[4] for r in part], k)
            buf += "".join("{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7}\n".format(s0 + i, r[0], r[1], r[2], k1[i], r[3], r[4], k2[i])
                           for i, r in enumerate(part)).encode("latin-1")
            while len(buf) >= 65536:
                f.write(fileutil._bgzf_member(buf[:65536], compresslevel))
                buf = buf[65536:]
                nrow_blocks += 1
        f.write(fileutil._bgzf_member(buf, compresslevel))                                   # __init__.py:1765: _write_block(_buffer)
        nrow_blocks += 1
    return nrow_blocks


def make_graph(inputs, k, kdbg, quiet=True, do_not_canonicalize=False, replace_with_none=False, sorted=False, device=0):
    """`kmerdb graph` (kmerdb/__init__.py:1635-1788): edges of every input, summed k-mer counts, header, .kdbg file.
    -> (metadata OrderedDict, number of rows)."""
    if os.path.splitext(kdbg)[-1] != ".kdbg":
        raise IOError("Destination .kdbg filepath does not end in '.kdbg'")
    N = 4 ** k
    counts = np.zeros(N, dtype="uint64")
    file_metadata, data = [], []
    for f in inputs:                                                                          # :1676-1682
        data_, f_metadata, counts_ = make_edges_from_fasta(f, k, quiet=quiet, canonicalize=not do_not_canonicalize,
                                                           replace_with_none=replace_with_none, device=device)
        data += data_
        counts = counts + counts_
        file_metadata.append(f_metadata)
    all_observed_kmers = sum(fm["total_kmers"] for fm in file_metadata)                       # :1708-1710
    unique_kmers = int(np.count_nonzero(counts))
    unique_nullomers = N - unique_kmers if do_not_canonicalize is True else int((N / 2) - unique_kmers)
    metadata = OrderedDict({
        "version": fileutil.VERSION, "metadata_blocks": 1, "k": k, "total_kmers": all_observed_kmers,
        "unique_kmers": unique_kmers, "unique_nullomers": unique_nullomers, "sorted": sorted, "tags": [],
        "files": file_metadata,
    })
    write_kdbg(kdbg, metadata, data, k)
    if not quiet:
        import sys
        sys.stderr.write("Edges in file:  {0}\n".format(len(data)))
    return metadata, len(data)
