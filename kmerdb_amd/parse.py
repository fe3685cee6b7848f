"""kmerdb_amd.parse -- drop-in for the reference's kmerdb/parse.py: same function
names, argument meaning, return values and error behaviour; the per-read /
per-k-mer Python loops (parse.py:128-137 -> kmer.py:489-577 -> kmer.py:234-317)
run as HIP kernels on an MI355X through libkdbhip.so.
"""
import logging
import os

import numpy as np

from . import reader, util
from .engine import Engine, KDB_N_DROP, KDB_N_EXPAND

logger = logging.getLogger(__file__)


def parse_sequence_file(seq_filepath, return_tuple=True):
    """kmerdb/parse.py:50-85: yield (seq_id, seq) string tuples from a fasta/fastq file."""
    if type(seq_filepath) is not str:
        raise TypeError("kmerdb_amd.parse.parse_sequence_file() expects a fasta/fastq sequence filepath as a str")
    for bases, offsets, ids in reader.iter_blocks(seq_filepath, want_ids=True):
        o = offsets.astype(np.int64)
        for r in range(len(o) - 1):
            seq = bytes(bases[o[r]:o[r + 1]]).decode("ascii", "replace")
            yield (ids[r], seq) if return_tuple else _Record(ids[r], seq)


class _Record:
    """The slice of Bio.SeqRecord the callers of parse_sequence_file use (.id, .seq, len)."""
    __slots__ = ("id", "seq")

    def __init__(self, id, seq):
        self.id = id
        self.seq = seq

    def __len__(self):
        return len(self.seq)


def _feed_file(eng, filepath):
    """Split `filepath` into records and submit them all to `eng` (asynchronous).  FASTA files are streamed in blocks
    (a record longer than a block goes in pieces that overlap by k - 1 residues), FASTQ in blocks of whole records.
    -> (total_reads, min_len, max_len, sum_len, reader); the caller calls reader.release() once the engine has synced
    (its DMA reads the reader's pinned ring until then).  ValueError if the file holds no records (reference: max([])
    at parse.py:144)."""
    blocks = reader.BlockReader(filepath, pinned=True, overlap=eng.k - 1, hold_ring=True)    # residues are split straight into pinned memory
    try:
        for blk in blocks:
            bases, offsets, _ = blk
            if len(offsets) < 2:
                continue
            # asynchronous: the next block is parsed while this one is copied and counted
            if blocks.pinned:
                eng.submit_pinned(bases, offsets, continues=blk.cont)
            else:
                eng.submit(bases, offsets, continues=blk.cont)
        if blocks.total_reads == 0:
            raise ValueError("no sequence records found in '{0}'".format(filepath))
    except BaseException:
        try:
            eng.sync()
        except Exception:
            pass
        blocks.release()
        raise
    return blocks.total_reads, blocks.min_len, blocks.max_len, blocks.sum_len, blocks


def _feed_shard(eng, filepath, rank, world, block_bytes=None):
    """Submit shard `rank` of `world` of the file's records to `eng` (reader.ShardedBlockReader: byte ranges
    re-synchronised on record starts, whole records only).  -> (reads, sum_len, min_len, max_len, reader); the shard may
    be empty (reads == 0).  The caller releases the reader after the engine has synced."""
    kw = {} if block_bytes is None else {"block_bytes": block_bytes}
    blocks = reader.ShardedBlockReader(filepath, rank, world, pinned=True, hold_ring=True, **kw)
    reads = sum_len = 0
    min_len, max_len = 1 << 62, 0
    try:
        for bases, offsets, _ in blocks:
            if len(offsets) < 2:
                continue
            lens = np.diff(offsets.astype(np.int64))
            reads += len(lens)
            sum_len += int(lens.sum())
            min_len, max_len = min(min_len, int(lens.min())), max(max_len, int(lens.max()))
            if blocks.pinned:
                eng.submit_pinned(bases, offsets)
            else:
                eng.submit(bases, offsets)
    except BaseException:
        try:
            eng.sync()
        except Exception:
            pass
        blocks.release()
        raise
    return reads, sum_len, min_len, max_len, blocks


def _file_metadata(filepath, k, md5, sha256, total_reads, total_kmers, unique_kmers, min_len, max_len, sum_len):
    """The per-file dict of parse.py:149-160."""
    return {
        "filename": filepath,
        "md5": md5,
        "sha256": sha256,
        "total_reads": total_reads,
        "total_kmers": int(total_kmers),
        "unique_kmers": int(unique_kmers),
        "nullomers": int(4 ** k - unique_kmers),                       # parse.py:143
        "min_read_length": min_len,
        "max_read_length": max_len,
        "avg_read_length": int(sum_len / total_reads),                 # int(np.mean(...)) parse.py:146
    }


def _check_args(filepath, k, replace_with_none):
    if filepath is None or type(filepath) is not str:
        raise TypeError("kmerdb_amd.parse.parsefile expects a str as its first positional argument")
    elif not os.path.exists(filepath):
        raise OSError("kmerdb_amd.parse.parsefile could not find the file '{0}' on the filesystem".format(filepath))
    elif k is None or type(k) is not int:
        raise TypeError("kmerdb_amd.parse.parsefile expects an int as its second positional argument")
    elif type(replace_with_none) is not bool:
        raise TypeError("kmerdb_amd.parse.parsefile expects the keyword argument 'replace_with_none' to be a bool")


_pool = {}                     # (k, canonicalize, n_mode, device) -> an idle Engine kept from the last parsefile call
_pool_lock = __import__("threading").Lock()
POOL_MAX_K = 13                # engines of larger k hold gigabytes of HBM (vector + page arena): those are closed, not kept


def _engine_for(k, canonicalize, n_mode, device):
    """An engine for one parsefile call: the one the last call with the same parameters left behind (reset), or a new one.
    Creating an engine and, above all, destroying it -- freeing the vector and the scatter scratch -- cost 13 ms per file at
    k = 12, seven times the counting of 10 M reads; a library user who calls parsefile in a loop (as _profile does,
    kmerdb/__init__.py:1888-1891) pays it once."""
    key = (k, bool(canonicalize), int(n_mode), int(device))
    with _pool_lock:
        eng = _pool.pop(key, None)
    if eng is not None:
        try:
            eng.reset()
            return key, eng
        except Exception:
            eng.close()
    return key, Engine(k, canonicalize=bool(canonicalize), n_mode=n_mode, device=device)


def _engine_done(key, eng, ok):
    """Back into the pool (one engine per parameter set, k <= POOL_MAX_K, only after a call that went through), else closed."""
    if ok and key[0] <= POOL_MAX_K:
        with _pool_lock:
            old = _pool.pop(key, None)
            _pool[key] = eng
            while len(_pool) > 4:
                _pool.pop(next(iter(_pool))).close()
        if old is not None:
            old.close()
    else:
        eng.close()


def release_engines():
    """Close the engines parsefile keeps between calls (their HBM is freed; also done at interpreter exit)."""
    with _pool_lock:
        engines = list(_pool.values())
        _pool.clear()
    for e in engines:
        e.close()


__import__("atexit").register(release_engines)


def parsefile(filepath, k, replace_with_none=True, canonicalize=True, device=0, engine=None, timings=None):
    """Count all k-mers of one FASTA/FASTQ file -- kmerdb/parse.py:90-163.

    :returns: (counts uint64[4**k], file_metadata dict, nullomer_array uint64[])
    :raises TypeError: filepath not a str / k not an int / replace_with_none not a bool  (parse.py:109-116)
    :raises OSError: file does not exist                                                (parse.py:111-112)
    :raises ValueError: unknown suffix; a record shorter than k; a residue outside ACGTN; no records
                        (the reference raises ValueError / AttributeError / KeyError / NameError there --
                        never a silent skip; see DESIGN.md "Error behaviour")

    `device` / `engine` are additions: which GPU to use, or an existing Engine to accumulate into
    (it is reset first, so the result is this file's vector like the reference's).  `timings`: a dict that receives the
    wall-clock seconds of the stages (read + split + submit; the rest of the counting + copy-back; nullomers; waiting for the
    digests) and the thread time of md5 and sha256, which run beside all of them.

    nullomer_array (parse.py:139-140) is compacted on the device (kdb_nullomers) and copied back next to the vector; the
    statistics of parse.py:141-147 come from the device's sweep of the vector as well -- nothing on the host walks 4^k bins.
    """
    import time
    _check_args(filepath, k, replace_with_none)
    t_start = time.perf_counter()
    sums = util.ChecksumJob(filepath)          # md5 + sha256 of the raw file (util.py:35-50), overlapped with the counting

    own = engine is None
    if own:
        key, eng = _engine_for(k, canonicalize is True, KDB_N_DROP if replace_with_none else KDB_N_EXPAND, device)
    else:
        eng = engine
    ok = False
    try:
        if not own:
            eng.reset()
        t_engine = time.perf_counter()
        total_reads, min_len, max_len, sum_len, blocks = _feed_file(eng, filepath)
        t_fed = time.perf_counter()
        try:
            counts, total_kmers, unique_kmers = eng.finish()
        finally:
            blocks.release()
        t_counted = time.perf_counter()
        nullomer_array = eng.nullomers(n=4 ** k - unique_kmers)            # parse.py:139-140
        t_null = time.perf_counter()
        ok = True
    finally:
        if own:
            _engine_done(key, eng, ok)

    t_closed = time.perf_counter()
    md5, sha256 = sums.result()
    t_sums = time.perf_counter()
    if timings is not None:
        timings.update({"engine_setup_s": t_engine - t_start, "read_split_submit_s": t_fed - t_engine, "count_rest_and_copy_back_s": t_counted - t_fed,
                        "nullomers_s": t_null - t_counted, "engine_close_s": t_closed - t_null, "wait_for_digests_s": t_sums - t_closed,
                        "md5_thread_s": sums.seconds.get("md5"), "sha256_thread_s": sums.seconds.get("sha256")})
    file_metadata = _file_metadata(filepath, k, md5, sha256, total_reads, total_kmers, unique_kmers, min_len, max_len, sum_len)
    assert file_metadata["nullomers"] == len(nullomer_array), "inconsistent nullomer count"
    logger.info("Finished counting k-mers from '{0}'".format(filepath))
    return counts, file_metadata, nullomer_array


def parsefile_devices(filepath, k, devices, replace_with_none=True, canonicalize=True, block_bytes=None):
    """parsefile over several GPUs of THIS process (single-process form): one engine and one reader thread
    per entry of `devices`; engine j counts shard j of the file's records (a record is counted independently of all
    others, parse.py:128-137); kdb_reduce sums the vectors into the first engine's over xGMI peer access; one copy to the
    host.  Same return value and errors as parsefile.  (One process per GPU: distributed.parsefile_distributed.)"""
    from concurrent.futures import ThreadPoolExecutor
    from .engine import reduce_engines
    _check_args(filepath, k, replace_with_none)
    devices = [int(d) for d in devices]
    if len(devices) == 0:
        raise ValueError("parsefile_devices needs at least one device")
    if len(devices) == 1:
        return parsefile(filepath, k, replace_with_none=replace_with_none, canonicalize=canonicalize, device=devices[0])
    sums = util.ChecksumJob(filepath)
    n = len(devices)
    engines, readers = [], [None] * n
    try:
        for d in devices:
            engines.append(Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP if replace_with_none else KDB_N_EXPAND, device=d))

        def work(j):
            reads, sum_len, mn, mx, readers[j] = _feed_shard(engines[j], filepath, j, n, block_bytes)
            engines[j].sync()                      # a short record / bad residue of this shard is raised here
            return reads, sum_len, mn, mx

        with ThreadPoolExecutor(max_workers=n) as ex:
            stats = list(ex.map(work, range(n)))
        total_reads = sum(s[0] for s in stats)
        if total_reads == 0:
            raise ValueError("no sequence records found in '{0}'".format(filepath))
        sum_len = sum(s[1] for s in stats)
        min_len, max_len = min(s[2] for s in stats), max(s[3] for s in stats)
        reduce_engines(engines, root=0)
        counts, total_kmers, unique_kmers = engines[0].finish()
        nullomer_array = engines[0].nullomers(n=4 ** k - unique_kmers)
    finally:
        for e in engines:
            e.close()                              # (syncs: nothing reads the readers' rings any more)
        for b in readers:
            if b is not None:
                b.release()
    md5, sha256 = sums.result()
    return counts, _file_metadata(filepath, k, md5, sha256, total_reads, total_kmers, unique_kmers, min_len, max_len, sum_len), nullomer_array


def parsefile_folded(filepath, k, engine, replace_with_none=True, sums=None, into=None, lock=None, fold=True):
    """One file of a samplesheet: count it into `engine`, fold its vector into the on-device accumulator (the engine's
    own, or that of the engine `into` -- several engines then work on several files at the same time and `lock`
    lets one of them fold at a time) and return only the per-file metadata (parse.py:149-160): the vector never
    leaves HBM (kmerdb/__init__.py:1888-1891 sums vectors).  `sums`: a ChecksumJob started earlier.
    fold=False: the only file of a job -- its vector IS the sum; it stays in the engine's count vector (no second 4^k vector,
    no sweep) and the caller copies it back with engine.finish()."""
    _check_args(filepath, k, replace_with_none)
    if sums is None:
        sums = util.ChecksumJob(filepath)
    total_reads, min_len, max_len, sum_len, blocks = _feed_file(engine, filepath)
    try:
        if not fold:
            _, total_kmers, unique_kmers = engine.finish(copy=False)
        elif lock is not None:
            engine.sync()                               # (wait for the counting outside the lock)
            with lock:
                total_kmers, unique_kmers = engine.fold_file(into=into)
        else:
            total_kmers, unique_kmers = engine.fold_file(into=into)
    finally:
        blocks.release()
    md5, sha256 = sums.result()
    logger.info("Finished counting k-mers from '{0}'".format(filepath))
    return _file_metadata(filepath, k, md5, sha256, total_reads, total_kmers, unique_kmers, min_len, max_len, sum_len)
