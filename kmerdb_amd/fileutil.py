"""kmerdb_amd.fileutil -- the .kdb file format (BGZF-framed YAML header + TSV rows) of the reference's
kmerdb/fileutil.py (KDBWriter :486-565, KDBReader :115-480), without Biopython and without per-row Python:

  write_kdb(path, metadata, counts)   header member(s) from Python, rows by the native writer
                                      (kdb_write_kdb_rows: 65536-byte BGZF members, Python float repr)
  read_kdb(path)                      -> KDB object with .metadata .k .kmer_ids .counts .frequencies (numpy),
                                      the attribute surface of KDBReader after slurp() (fileutil.py:229-241, :308-466)
"""
import ctypes
import gzip
import io
import math
import os
import struct
import sys
import zlib
from collections import OrderedDict

import numpy as np
import yaml

from . import _abi

VERSION = "0.9.6"                                   # kmerdb/config.py:20: the format version this writer emits
header_delimiter = "\n" + ("=" * 24) + "\n"        # kmerdb/config.py:22
KDB_COLUMN_NUMBER = 4                               # kmerdb/config.py:38

_FILE_KEYS = ("filename", "md5", "sha256", "total_reads", "total_kmers", "unique_kmers", "nullomers",
              "min_read_length", "max_read_length", "avg_read_length")


def _bgzf_member(data, compresslevel=6):
    """One BGZF block, as Bio.bgzf.BgzfWriter._write_block frames it."""
    assert len(data) <= 65536
    c = zlib.compressobj(compresslevel, zlib.DEFLATED, -15, zlib.DEF_MEM_LEVEL, 0)
    compressed = c.compress(data) + c.flush()
    if len(compressed) > 65536 - 26:
        c = zlib.compressobj(0, zlib.DEFLATED, -15, zlib.DEF_MEM_LEVEL, 0)
        compressed = c.compress(data) + c.flush()
    bsize = struct.pack("<H", len(compressed) + 25)
    crc = struct.pack("<I", zlib.crc32(data) & 0xFFFFFFFF)
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00\x42\x43\x02\x00" + bsize + compressed + crc
            + struct.pack("<I", len(data)))


def validate_metadata(metadata):
    """The checks config.kdb_metadata_schema (kmerdb/config.py:88-136) makes on a .kdb header."""
    if type(metadata) is not dict and not isinstance(metadata, OrderedDict):
        raise TypeError("kmerdb_amd.fileutil expects a valid metadata dictionary")
    for key in ("version", "metadata_blocks", "k", "total_kmers", "unique_kmers", "unique_nullomers", "sorted", "tags", "files"):
        if key not in metadata:
            raise ValueError("kdb metadata is missing the key '{0}'".format(key))
    for f in metadata["files"]:
        for key in _FILE_KEYS:
            if key not in f:
                raise ValueError("kdb file metadata is missing the key '{0}'".format(key))
        if len(f["md5"]) != 32 or len(f["sha256"]) != 64:
            raise ValueError("kdb file metadata has a malformed checksum")


def header_bytes(metadata):
    """YAML + delimiter exactly as KDBWriter.__init__ builds it (fileutil.py:530-538), incl. its
    metadata_blocks estimate from sys.getsizeof."""
    md = dict(metadata)
    b = bytes(yaml.dump(md, sort_keys=False), "utf-8") + bytes(header_delimiter, "utf-8")
    md["metadata_blocks"] = math.ceil(sys.getsizeof(b) / (2 ** 16))
    b = bytes(yaml.dump(md, sort_keys=False), "utf-8") + bytes(header_delimiter, "utf-8")
    md["metadata_blocks"] = math.ceil(sys.getsizeof(b) / (2 ** 16))
    b = bytes(yaml.dump(md, sort_keys=False), "utf-8") + bytes(header_delimiter, "utf-8")
    return b, md["metadata_blocks"]


def default_writer_threads():
    """Threads the row writer formats and deflates with: the CPUs this process can really use (its affinity mask cut down to its
    cgroup's CPU quota, util.effective_cpus), at most 64 (KDB_WRITER_THREADS overrides).  More threads than the quota pays for are
    slower, not faster: on a box with 256 visible CPUs and ~16 CPUs of quota the k = 15 rows took 2.3 s with 16 threads and 5.1 s with 64."""
    env = os.environ.get("KDB_WRITER_THREADS")
    if env:
        return max(1, int(env))
    from . import util
    return max(1, min(64, util.effective_cpus()))


ENCODERS = {None: -1, "default": -1, "rows": 0, "zlib": 1}


def write_kdb(path, metadata, counts, compresslevel=6, nthreads=None, encoder=None):
    """Write `<path>`: header member(s), then the 4^k rows  i \\t kmer_id \\t count \\t frequency  in kmer-id order
    (kmerdb/__init__.py:1939-1998; the unsorted branch -- the reference's --sorted branch raises NameError).
    `encoder`: "rows" (default: the row-aware deflate encoder of kdb_write_kdb_rows_ex) or "zlib" (zlib at `compresslevel`,
    as Bio.bgzf does); the decompressed file is the same.  -> number of row members written."""
    validate_metadata(metadata)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    k = int(metadata["k"])
    if counts.size != 4 ** k:
        raise ValueError("counts has {0} entries, expected 4^{1}".format(counts.size, k))
    hb, nblocks = header_bytes(metadata)
    with io.open(path, "wb") as f:
        for _ in range(nblocks):                             # fileutil.py:551-556
            f.write(_bgzf_member(hb[:65536], compresslevel))
            hb = hb[65536:]
    if nthreads is None:
        nthreads = default_writer_threads()
    nb = ctypes.c_uint64(0)
    if encoder not in ENCODERS:
        raise ValueError("unknown encoder '{0}' (rows, zlib)".format(encoder))
    _abi.check(_abi.lib().kdb_write_kdb_rows_ex(path.encode(), counts.ctypes.data, counts.size, int(metadata["total_kmers"]),
                                                int(compresslevel), int(nthreads), ENCODERS[encoder], ctypes.byref(nb)))
    return nb.value


def write_kdb_from_engine(path, metadata, engine, folded=False, compresslevel=6, nthreads=None, encoder=None):
    """write_kdb for a vector that is still on the device: header member(s), then kdb_copy_back_and_write_kdb_rows -- the copy-back runs in
    pieces beside the row writer, which starts on the rows that have arrived.  `metadata` holds the totals already (Engine.finish(copy=False) /
    finish_folded(copy=False)).  -> (counts uint64[4^k], number of row members written)."""
    validate_metadata(metadata)
    k = int(metadata["k"])
    if engine.nbins != 4 ** k:
        raise ValueError("the engine counts k = {0}, the metadata says k = {1}".format(engine.k, k))
    hb, nblocks = header_bytes(metadata)
    with io.open(path, "wb") as f:
        for _ in range(nblocks):                             # fileutil.py:551-556
            f.write(_bgzf_member(hb[:65536], compresslevel))
            hb = hb[65536:]
    if nthreads is None:
        nthreads = default_writer_threads()
    if encoder not in ENCODERS:
        raise ValueError("unknown encoder '{0}' (rows, zlib)".format(encoder))
    counts = np.empty(4 ** k, dtype=np.uint64)
    nb = ctypes.c_uint64(0)
    _abi.check(_abi.lib().kdb_copy_back_and_write_kdb_rows(engine._h, 1 if folded else 0, counts.ctypes.data, path.encode(), int(metadata["total_kmers"]),
                                                           int(compresslevel), int(nthreads), ENCODERS[encoder], ctypes.byref(nb)))
    return counts, nb.value


class KDB:
    """What the reference's KDBReader exposes after slurp(): fileutil.py:229-241.  `.frequencies` is what the reference's reader computes
    for an unsorted file -- count / 4^k, fileutil.py:363 (NOT the file's fourth column, which holds count / total_kmers); the file's own
    column is kept as `.file_frequencies`."""

    def __init__(self, metadata, kmer_ids, counts, file_frequencies):
        self.metadata = metadata
        self.k = metadata["k"]
        self.kmer_ids = kmer_ids
        self.counts = counts
        self.file_frequencies = file_frequencies
        self.sorted = metadata.get("sorted", False)
        if self.sorted:
            self.frequencies = file_frequencies                                                # fileutil.py:407-417: the sorted branch keeps the column
        else:
            self.frequencies = counts.astype(np.float64) / np.float64(4 ** int(self.k))        # fileutil.py:363: float(count) / N


def _read_header(path):
    """The YAML header of a .kdb: the text in front of the delimiter line, from the first gzip member(s)."""
    delim = header_delimiter.encode()
    try:
        with gzip.open(path, "rb") as f:
            head = b""
            while True:
                chunk = f.read(1 << 16)
                head += chunk
                at = head.find(delim)
                if at >= 0 or not chunk or len(head) > (64 << 20):
                    break
    except (OSError, EOFError, zlib.error) as e:
        raise ValueError("'{0}' is not a valid .kdb file: {1}".format(path, e)) from e   # reference: ValueError (test_fileutil.py:97-113)
    if at < 0:
        raise ValueError("'{0}' has no .kdb header delimiter".format(path))
    metadata = yaml.safe_load(head[:at].decode("utf-8"))
    if not isinstance(metadata, dict) or "k" not in metadata:
        raise ValueError("'{0}' has no valid .kdb YAML header".format(path))
    return metadata


def read_kdb(path, nthreads=None):
    """Read a .kdb into numpy arrays (KDBReader + slurp, fileutil.py:128-297, :308-466).  A file of BGZF members -- what the reference
    and write_kdb produce -- is inflated and parsed by the native kdb_read_kdb_rows on `nthreads` threads (at k = 15 the rows are
    37 GB of text: one gzip stream through pandas does not get there); any other concatenation of gzip members goes through gzip +
    pandas as before.  Same checks as the reference: four columns (:354), row index == line number (:361), 4^k rows."""
    if type(path) is not str:
        raise TypeError("kmerdb_amd.fileutil.read_kdb expects a str filepath")
    metadata = _read_header(path)
    N = 4 ** int(metadata["k"])
    kmer_ids = np.zeros(N, dtype=np.uint64)
    counts = np.zeros(N, dtype=np.uint64)
    freqs = np.zeros(N, dtype=np.float64)
    try:
        lib = _abi.lib()
    except _abi.KdbHipError:
        lib = None
    if lib is not None:
        if nthreads is None:
            nthreads = default_writer_threads()
        nrows = ctypes.c_uint64(0)
        rc = lib.kdb_read_kdb_rows(path.encode(), N, kmer_ids.ctypes.data, counts.ctypes.data, freqs.ctypes.data, int(nthreads), ctypes.byref(nrows))
        if rc == _abi.KDB_OK:
            return KDB(metadata, kmer_ids, counts, freqs)
        if rc != _abi.KDB_ERR_STATE:
            raise ValueError("'{0}' is not a valid .kdb file: {1}".format(path, _abi.last_error()))
        kmer_ids[:] = 0
        counts[:] = 0
        freqs[:] = 0
    try:
        with gzip.open(path, "rb") as f:
            raw = f.read()
    except (OSError, EOFError, zlib.error) as e:
        raise ValueError("'{0}' is not a valid .kdb file: {1}".format(path, e)) from e
    delim = header_delimiter.encode()
    body = raw[raw.find(delim) + len(delim):]
    import pandas as pd
    df = pd.read_csv(io.BytesIO(body), sep="\t", header=None, dtype={0: np.uint64, 1: np.uint64, 2: np.uint64, 3: np.float64},
                     float_precision="round_trip")
    if df.shape[1] != KDB_COLUMN_NUMBER:                                                   # fileutil.py:354
        raise ValueError("'{0}': expected {1} columns, found {2}".format(path, KDB_COLUMN_NUMBER, df.shape[1]))
    if df.shape[0] != N:
        raise ValueError("'{0}': expected 4^k = {1} rows, found {2}".format(path, N, df.shape[0]))
    if bool(df[3].isna().any()):
        raise ValueError("'{0}': a row has fewer than {1} columns".format(path, KDB_COLUMN_NUMBER))
    if not np.array_equal(df[0].to_numpy(dtype=np.uint64), np.arange(N, dtype=np.uint64)):  # fileutil.py:361
        raise ValueError("'{0}': a row's index does not match its line number".format(path))
    ids = df[1].to_numpy(dtype=np.uint64)
    idx = ids.astype(np.int64)
    kmer_ids[:] = ids                                                                      # fileutil.py:367-369
    counts[idx] = df[2].to_numpy(dtype=np.uint64)
    freqs[idx] = df[3].to_numpy(dtype=np.float64)
    return KDB(metadata, kmer_ids, counts, freqs)


# ---------------------------------------------------------------------------------------------------------------
# The reference's object surface: fileutil.open -> KDBReader / KDBWriter (kmerdb/fileutil.py:46-105, :115-480, :486-565).
# The fast paths are write_kdb / read_kdb above; these classes give a caller written against the reference's API the same
# names, argument checks and attributes.
# ---------------------------------------------------------------------------------------------------------------
class KDBWriter:
    """fileutil.KDBWriter (kmerdb/fileutil.py:486-565) without Bio.bgzf: the header member(s) at construction, then whatever .write()
    is given, cut into 65536-byte BGZF members (Bio.bgzf.BgzfWriter.write / _write_block).  _profile's own loop
    (kmerdb/__init__.py:1980-1998: write() per row, then `_write_block(_buffer)`, no close()) runs on it unchanged -- one Python call per
    row; write_kdb() formats and deflates the same rows natively."""

    def __init__(self, metadata, filename=None, mode="w", fileobj=None, compresslevel=6):
        if type(metadata) is not dict and not isinstance(metadata, OrderedDict):
            raise TypeError("kmerdb_amd.fileutil.KDBWriter expects a valid metadata dictionary as its first positional argument")
        validate_metadata(metadata)
        self.metadata = metadata
        self.k = metadata["k"]
        if fileobj:
            assert filename is None
            handle = fileobj
        else:
            if "w" not in mode.lower() and "a" not in mode.lower():
                raise ValueError("Must use write or append mode, not %r" % mode)
            if "a" in mode.lower():
                raise NotImplementedError("Append mode is not implemented yet")
            handle = io.open(filename, "wb")
        self._text = "b" not in mode.lower()
        self._handle = handle
        self._buffer = b""
        self.compresslevel = compresslevel
        hb, nblocks = header_bytes(metadata)
        self.metadata["metadata_blocks"] = nblocks
        if "b" in mode.lower():
            for _ in range(nblocks):                             # fileutil.py:551-556
                self._write_block(hb[:65536])
                hb = hb[65536:]
                self._handle.flush()
        else:                                                    # fileutil.py:557-563: text mode writes the YAML through the buffer, without the delimiter
            self.write(bytes(yaml.dump(metadata, sort_keys=False), "utf-8"))
            self._handle.flush()

    def _write_block(self, block):
        self._handle.write(_bgzf_member(bytes(block), self.compresslevel))

    def write(self, data):
        if isinstance(data, str):
            data = data.encode("latin-1")                        # (Bio.bgzf.BgzfWriter.write)
        self._buffer += data
        while len(self._buffer) >= 65536:
            self._write_block(self._buffer[:65536])
            self._buffer = self._buffer[65536:]

    def flush(self):
        while len(self._buffer) >= 65536:
            self._write_block(self._buffer[:65536])
            self._buffer = self._buffer[65536:]
        self._write_block(self._buffer)
        self._buffer = b""
        self._handle.flush()

    def close(self):
        if self._buffer:
            self.flush()
        self._write_block(b"")                                   # BGZF end-of-file marker (Bio.bgzf.BgzfWriter.close)
        self._handle.flush()
        self._handle.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class KDBReader:
    """fileutil.KDBReader (kmerdb/fileutil.py:115-480): the header at construction (.metadata, .k, zeroed .kmer_ids / .counts /
    .frequencies of length 4^k, :229-241), the rows with slurp() (:308-466; here through read_kdb: BGZF members inflated and parsed in
    parallel).  sort=True and with_index=True are not offered (the reference's sorted branch re-reads the file after an np.lexsort;
    nothing on this path uses either)."""

    def __init__(self, filename, mode="r", sort=False, slurp=False, with_index=False):
        if type(filename) is not str:
            raise TypeError("kmerdb_amd.fileutil.KDBReader expects a str as its first positional argument")
        if "w" in mode.lower() or "a" in mode.lower():
            raise ValueError("Must use read mode (default), not write or append mode")       # fileutil.py:151-153
        if sort or with_index:
            raise ValueError("kmerdb_amd.fileutil.KDBReader offers neither sort=True nor with_index=True")
        if not os.path.exists(filename):
            raise IOError("kmerdb_amd.fileutil.KDBReader could not find '{0}' on the filesystem".format(filename))
        self._filepath = filename
        self.metadata = _read_header(filename)
        self.k = self.metadata["k"]
        self.sorted = self.metadata.get("sorted", False)
        N = 4 ** int(self.k)
        self.kmer_ids = np.zeros(N, dtype="uint64")
        self.counts = np.zeros(N, dtype="uint64")
        self.frequencies = np.zeros(N, dtype="float64")
        self.completed = False
        self.index = None
        if slurp:
            self.slurp()

    def slurp(self, sort=False, with_index=False):
        if type(sort) is not bool or type(with_index) is not bool:
            raise TypeError("kmerdb_amd.fileutil.KDBReader.slurp expects bools")
        if sort or with_index:
            raise ValueError("kmerdb_amd.fileutil.KDBReader offers neither sort=True nor with_index=True")
        kdb = read_kdb(self._filepath)
        self.kmer_ids, self.counts, self.frequencies = kdb.kmer_ids, kdb.counts, kdb.frequencies
        self.file_frequencies = kdb.file_frequencies
        self.completed = True
        return self.counts                                        # fileutil.py:466

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None


def open(filepath, mode="r", metadata=None, sort=False, slurp=False, with_index=False):       # noqa: A001 - the reference's name
    """fileutil.open (kmerdb/fileutil.py:46-105): same argument checks, a KDBReader for 'r' modes, a KDBWriter for 'w' / 'x' modes."""
    if type(filepath) is not str:
        raise TypeError("kmerdb_amd.fileutil.open expects a str as its first positional argument")
    elif type(mode) is not str:
        raise TypeError("kmerdb_amd.fileutil.open expects the keyword argument 'mode' to be a str")
    if (mode == "w" or mode == "x") and (metadata is not None and (isinstance(metadata, OrderedDict) or type(metadata) is dict)):
        pass
    elif mode == "w" or mode == "x":
        raise TypeError("kmerdb_amd.fileutil.open expects an additional metadata dictionary")
    elif sort is None or type(sort) is not bool:
        raise TypeError("kmerdb_amd.fileutil.open expects a boolean for the keyword argument 'sort'")
    elif slurp is None or type(slurp) is not bool:
        raise TypeError("kmerdb_amd.fileutil.open expects a boolean for the keyword argument 'slurp'")
    elif with_index is None or type(with_index) is not bool:
        raise TypeError("kmerdb_amd.fileutil.open expects a boolean for the keyword argument 'with_index'")
    modes = set(mode)
    if modes - set("xrwbt") or len(mode) > len(modes):
        raise ValueError("invalid mode: {}".format(mode))
    if "t" in modes and "b" in modes:
        raise ValueError("can't have text and binary mode at once")
    elif not ("x" in modes or "r" in modes or "w" in modes):
        raise ValueError("must have exactly one or read/write")
    if "r" in mode.lower():
        return KDBReader(filename=filepath, mode=mode, sort=sort, slurp=slurp, with_index=with_index)
    return KDBWriter(metadata, filename=filepath, mode=mode)
