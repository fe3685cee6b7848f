"""Engine: the Python handle on one libkdbhip engine (one GPU, one 4^k uint64 count vector in HBM).

Takes over  counts = np.zeros(4**k) ... counts[kmer_id] += 1  of the reference
(kmerdb/parse.py:117-137).  numpy arrays / raw pointers in, numpy arrays out;
torch is only used by callers that want the count vector as a tensor for RCCL.
"""
import collections
import ctypes
import threading

import numpy as np

from . import _abi
from ._abi import KDB_N_DROP, KDB_N_EXPAND  # noqa: F401  (re-exported)


NO_WINDOW = np.uint64(0xFFFFFFFFFFFFFFFF)


class _DeviceArray:
    """Expose a raw device pointer through __cuda_array_interface__ (zero-copy torch.as_tensor)."""

    def __init__(self, ptr, n, typestr, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class Engine:
    def __init__(self, k, canonicalize=True, n_mode=KDB_N_DROP, device=0, table_ptr=None, algo=None):
        if type(k) is not int:
            raise TypeError("k must be an int")
        self._h = ctypes.c_void_p()
        self._lib = _abi.lib()
        _abi.check(self._lib.kdb_create(k, 1 if canonicalize else 0, int(n_mode), int(device),
                                        ctypes.c_void_p(table_ptr) if table_ptr else None, ctypes.byref(self._h)))
        self.k = k
        self.nbins = 4 ** k
        self.canonicalize = bool(canonicalize)
        self.n_mode = int(n_mode)
        self.device = int(device)
        if algo is not None:
            self.set_option("algo", algo)

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.kdb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- counting ---------------------------------------------------------------------
    def reset(self):
        _abi.check(self._lib.kdb_reset(self._h))

    def submit(self, bases, offsets, continues=False):
        """bases: uint8[nbytes] raw ASCII; offsets: uint64[nreads+1]. Asynchronous.
        continues: record 0 is the next piece of the previous submit's last record and starts with its last k-1 residues."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nreads = len(offsets) - 1
        if nreads <= 0:
            return
        if int(offsets[-1]) > bases.size:
            raise ValueError("offsets exceed the residue buffer")
        _abi.check(self._lib.kdb_submit_ex(self._h, bases.ctypes.data, bases.size, offsets.ctypes.data, nreads,
                                           _abi.KDB_SUBMIT_CONTINUES if continues else 0))

    def submit_pinned(self, bases, offsets, continues=False):
        """Like submit, for `bases` allocated with pinned_empty(): no staging copy; keep `bases` alive until sync()."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nreads = len(offsets) - 1
        if nreads <= 0:
            return
        if bases.dtype != np.uint8 or not bases.flags.c_contiguous or int(offsets[-1]) > bases.size:
            raise ValueError("bases must be a contiguous uint8 array covering the offsets")
        self._keep = getattr(self, "_keep", [])
        self._keep.append(bases)
        _abi.check(self._lib.kdb_submit_ex(self._h, bases.ctypes.data, bases.size, offsets.ctypes.data, nreads,
                                           _abi.KDB_SUBMIT_PINNED | (_abi.KDB_SUBMIT_CONTINUES if continues else 0)))

    def submit_device(self, bases_ptr, nbytes, offsets_ptr, nreads):
        """Inputs already in HBM (raw device pointers, e.g. tensor.data_ptr())."""
        _abi.check(self._lib.kdb_submit_device(self._h, ctypes.c_void_p(bases_ptr), int(nbytes),
                                               ctypes.c_void_p(offsets_ptr), int(nreads)))

    def submit_device_const(self, bases_ptr, nbytes, offsets_ptr, nreads):
        """submit_device for a buffer the engine must not write: records must all have one length (else sync raises)."""
        _abi.check(self._lib.kdb_submit_device_const(self._h, ctypes.c_void_p(bases_ptr), int(nbytes),
                                                     ctypes.c_void_p(offsets_ptr), int(nreads)))

    def sync(self):
        _abi.check(self._lib.kdb_sync(self._h))
        self._keep = []

    def table_stats(self, copy=True):
        """-> (counts or None, Sum(counts), count_nonzero(counts)) of the vector as it is -- for a vector that other
        ranks' counts were reduced into, where finish()'s Sum == emitted check does not apply."""
        counts = np.empty(self.nbins, dtype=np.uint64) if copy else None
        s = ctypes.c_uint64(0)
        u = ctypes.c_uint64(0)
        _abi.check(self._lib.kdb_table_stats(self._h, counts.ctypes.data if copy else None, ctypes.byref(s), ctypes.byref(u)))
        return counts, s.value, u.value

    def fold_file(self, into=None):
        """End of one input file of a samplesheet: add its vector to the on-device accumulator (this engine's, or that of
        the engine `into` when several engines count files at the same time -- one fold at a time per accumulator),
        clear it.  -> (total_kmers, unique_kmers) of that file (kmerdb/__init__.py:1888-1891 without leaving HBM)."""
        total = ctypes.c_uint64(0)
        unique = ctypes.c_uint64(0)
        acc = into if into is not None else self
        _abi.check(self._lib.kdb_fold_file_into(self._h, acc._h, ctypes.byref(total), ctypes.byref(unique)))
        self._keep = []
        return total.value, unique.value

    def finish_folded(self, copy=True):
        """-> (accumulated counts or None, Sum, count_nonzero) over all folded files: the one device-to-host copy."""
        counts = np.empty(self.nbins, dtype=np.uint64) if copy else None
        total = ctypes.c_uint64(0)
        unique = ctypes.c_uint64(0)
        _abi.check(self._lib.kdb_finish_folded(self._h, counts.ctypes.data if copy else None,
                                               ctypes.byref(total), ctypes.byref(unique)))
        return counts, total.value, unique.value

    def finish(self, copy=True):
        """-> (counts uint64[4^k] or None, total_kmers, unique_kmers)  (parse.py:139-147)."""
        counts = np.empty(self.nbins, dtype=np.uint64) if copy else None
        total = ctypes.c_uint64(0)
        unique = ctypes.c_uint64(0)
        _abi.check(self._lib.kdb_finish(self._h, counts.ctypes.data if copy else None,
                                        ctypes.byref(total), ctypes.byref(unique)))
        return counts, total.value, unique.value

    def nullomers(self, n=None, folded=False):
        """-> uint64[] of the ids whose count is zero, ascending: nullomer_array of parse.py:139-140, compacted on the device.
        `n`: their number if the caller knows it (4^k - unique_kmers of finish()); asked of the device otherwise."""
        if n is None:
            c = ctypes.c_uint64(0)
            _abi.check(self._lib.kdb_nullomers(self._h, 1 if folded else 0, None, 0, ctypes.byref(c)))
            n = c.value
        ids = np.empty(int(n), dtype=np.uint64)
        got = ctypes.c_uint64(0)
        _abi.check(self._lib.kdb_nullomers(self._h, 1 if folded else 0, ids.ctypes.data if n else None, int(n), ctypes.byref(got)))
        if got.value != n:
            raise _abi.KdbHipError("kdb_nullomers: %d ids, %d expected" % (got.value, n))
        return ids

    def shred(self, seq):
        """kmer.shred for one record (N-free windows): -> (ids uint64[], positions uint64[])."""
        b = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
        arr = np.frombuffer(b, dtype=np.uint8)
        cap = max(len(b) - self.k + 1, 1)
        ids = np.empty(cap, dtype=np.uint64)
        pos = np.empty(cap, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        _abi.check(self._lib.kdb_shred(self._h, arr.ctypes.data if len(b) else None, len(b), ids.ctypes.data,
                                       pos.ctypes.data, cap, ctypes.byref(n)))
        return ids[:n.value], pos[:n.value]

    def window_ids(self, bases, offsets):
        """ids[p] = id of the window starting at residue p (uint64; NO_WINDOW where none). Synchronous."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.empty(bases.size, dtype=np.uint64)
        if bases.size and len(offsets) > 1:
            _abi.check(self._lib.kdb_window_ids(self._h, bases.ctypes.data, bases.size, offsets.ctypes.data,
                                                len(offsets) - 1, ids.ctypes.data))
        return ids

    # -- the count vector in HBM --------------------------------------------------------
    def table_ptr(self):
        p = ctypes.c_void_p()
        n = ctypes.c_uint64(0)
        _abi.check(self._lib.kdb_table(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def table_tensor(self):
        """The count vector as a torch int64 CUDA tensor (same bits as uint64; sums are identical mod 2^64).
        Syncs first: submits are asynchronous and, for k >= 14, partitioned batches are only added to the vector at a sync."""
        import torch
        self.sync()
        p, n = self.table_ptr()
        return torch.as_tensor(_DeviceArray(p, n, "<i8", self), device=f"cuda:{self.device}")

    # -- options / profiling --------------------------------------------------------------
    def set_option(self, name, value):
        _abi.check(self._lib.kdb_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int64(0)
        _abi.check(self._lib.kdb_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def traffic_counters(self):
        """HBM traffic of the LDS-histogram paths by the engine's own account, cumulative since reset() (synchronises):
        residue bytes handed to the kernels; pages holding elements and lines written (counted in 64-byte units, whatever the piece size), per scatter kernel; bytes
        of the count vector read + written by the histogram pass."""
        return {n: self.get_option(n) for n in ("bytes_in", "pages_bases", "lines_bases", "pages_ids", "lines_ids", "table_bytes")}

    def prof_enable(self, on=True):
        _abi.check(self._lib.kdb_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        _abi.check(self._lib.kdb_prof_reset(self._h))

    def prof(self):
        """-> {kernel_name: (total_ms, launches)} measured with HIP events on the engine's compute stream."""
        out = {}
        for i in range(_abi.KDB_N_KERNELS):
            ms = ctypes.c_double(0)
            n = ctypes.c_uint64(0)
            _abi.check(self._lib.kdb_prof_get(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            out[self._lib.kdb_prof_kernel_name(i).decode()] = (ms.value, n.value)
        return out


class IdsEngine(Engine):
    """An engine without a count vector: shred() / window_ids() only (kdb_create_ids)."""

    def __init__(self, k, canonicalize=True, device=0):
        if type(k) is not int:
            raise TypeError("k must be an int")
        self._h = ctypes.c_void_p()
        self._lib = _abi.lib()
        _abi.check(self._lib.kdb_create_ids(k, 1 if canonicalize else 0, int(device), ctypes.byref(self._h)))
        self.k = k
        self.nbins = 0
        self.canonicalize = bool(canonicalize)
        self.n_mode = KDB_N_DROP
        self.device = int(device)


def reduce_engines(engines, root=0):
    """kdb_reduce: sum the HBM count vectors of `engines` (one per device of this process, same k) into engines[root]'s,
    which then also carries everybody's emitted k-mers: engines[root].finish() reports the whole job (SURVEY 8(e),
    single-process form).  The other engines' vectors are partly overwritten: reset() them before further use."""
    arr = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    _abi.check(_abi.lib().kdb_reduce(arr, len(engines), int(root)))
    for e in engines:
        e._keep = []


_ids_tls = threading.local()


def ids_engine(k, canonicalize=True, device=0):
    """IdsEngine per (k, strand mode, device), cached PER THREAD: kmer.shred is called once per record by the reference's
    callers, and an engine's scratch and stream serve one caller at a time.  At most eight per thread; the least recently
    used one is dropped from the cache, not closed -- a caller that still holds it (graph.py keeps one across its block
    loop) goes on using it, and it closes itself when the last reference dies."""
    cache = getattr(_ids_tls, "engines", None)
    if cache is None:
        cache = _ids_tls.engines = collections.OrderedDict()
    key = (k, bool(canonicalize), int(device))
    e = cache.get(key)
    if e is not None:
        cache.move_to_end(key)
        return e
    while len(cache) >= 8:
        cache.popitem(last=False)
    e = cache[key] = IdsEngine(k, canonicalize, device)
    return e


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self._lib = _abi.lib()
        _abi.check(self._lib.kdb_host_alloc(ctypes.byref(self.ptr), nbytes))

    def __del__(self):
        try:
            if self.ptr:
                self._lib.kdb_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(nbytes):
    """uint8 numpy array of `nbytes` in pinned host memory (freed when the array and its views die)."""
    blk = _PinnedBlock(nbytes)
    buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(blk.ptr.value)
    arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes)
    # tie the allocation's lifetime to the array: numpy keeps `buf` alive, and `buf` keeps `blk`
    buf._kdb_block = blk
    return arr


def device_count():
    n = ctypes.c_int(0)
    _abi.check(_abi.lib().kdb_device_count(ctypes.byref(n)))
    return n.value
