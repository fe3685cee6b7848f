"""Multi-GPU: reads shard across ranks, one reduce of the 4^k vector at the end.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  The count
vector is a commutative integer sum over records (kmerdb/parse.py:128-137), so each rank counts its own
blocks into its own engine with no data-path collective, and a single SUM reduce over xGMI yields the
single-GPU vector bit for bit (uint64 viewed as int64: identical bits mod 2^64).

The reduce is issued in chunks (REDUCE_CHUNK_BYTES, in place on the root): RCCL stages a reduce through
internal buffers, and one 128 GiB call (k = 17) beside the 128 GiB vector leaves no room for them; 1 GiB
chunks keep every ring full (>> the 7 x 153 GB/s x latency product) and cost nothing measurable.
"""
import numpy as np

REDUCE_CHUNK_BYTES = 1 << 30


def block_owner(block_index, world_size):
    """Round-robin block -> rank map (each block is a run of whole records)."""
    return block_index % world_size


def shard_bounds(nreads, rank, world_size):
    """Contiguous [r0, r1) shard of `nreads` records for `rank`; shards partition range(nreads)."""
    return nreads * rank // world_size, nreads * (rank + 1) // world_size


def reduce_vector(t, dst=0, group=None, chunk_bytes=REDUCE_CHUNK_BYTES):
    """Sum an int64 count vector (torch tensor, CPU or CUDA) across ranks onto `dst`; in place, chunked.
    -> number of collective calls issued."""
    import torch.distributed as dist
    n = t.numel()
    step = max(1, int(chunk_bytes) // t.element_size())
    calls = 0
    for s in range(0, n, step):
        dist.reduce(t[s:s + step], dst=dst, op=dist.ReduceOp.SUM, group=group)
        calls += 1
    return calls


def reduce_scalars(values, group=None):
    """All-reduce per-file host metadata: (total_reads, sum_len, total_kmers) summed, (-min_len, max_len) maxed."""
    import torch
    import torch.distributed as dist
    s = torch.tensor(values["sum"], dtype=torch.int64)
    m = torch.tensor(values["max"], dtype=torch.int64)
    dev = values.get("device")
    if dev is not None:
        s, m = s.to(dev), m.to(dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return s.tolist(), m.tolist()


def reduce_counts(engine, dst=0, group=None, chunk_bytes=REDUCE_CHUNK_BYTES):
    """Reduce an Engine's HBM count vector onto rank `dst` (in place on the device).  Syncs the engine first.
    After this, rank `dst` must read its vector with Engine.table_stats(), not finish(): the vector now holds
    every rank's counts, which finish()'s Sum(counts) == emitted-by-this-engine check rightly rejects."""
    import torch
    t = engine.table_tensor()          # syncs: submits are asynchronous, and k >= 14 defers its histogram pass
    torch.cuda.synchronize(t.device)
    reduce_vector(t, dst=dst, group=group, chunk_bytes=chunk_bytes)
    torch.cuda.synchronize(t.device)
    return t


class RankFailed(RuntimeError):
    """Another rank raised before the collectives; every rank leaves parsefile_distributed together."""


def _agree_or_raise(local_error, device, group):
    """One small all-reduce every rank reaches whether or not it failed, so that a rank that raised while reading or
    counting its shard cannot leave the others waiting in the vector reduce."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if local_error is not None else 0], dtype=torch.int64)
    if device is not None:
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if local_error is not None:
        raise local_error
    if int(flag.item()):
        raise RankFailed("another rank failed while counting its shard (its own exception says why)")


def parsefile_distributed(filepath, k, replace_with_none=True, canonicalize=True, device=None, group=None,
                          block_bytes=None, engine_opts=None):
    """parse.parsefile over all ranks of the default process group: rank r reads and counts blocks r, r+W, ... of the
    file (reader.iter_blocks_sharded: byte ranges re-synchronised on record starts; a plain file is never read twice,
    a gzip stream is inflated by every rank but split into records only where owned); one chunked SUM reduce.
    Rank 0 returns (counts, file_metadata, nullomer_array) like kmerdb/parse.py:90-163, other ranks (None, None, None)."""
    import torch
    import torch.distributed as dist
    from . import parse, reader, util
    from .engine import Engine, KDB_N_DROP, KDB_N_EXPAND
    parse._check_args(filepath, k, replace_with_none)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    if device is None:
        device = torch.cuda.current_device()
    coll_dev = f"cuda:{device}" if backend == "nccl" else None      # gloo reduces host tensors
    sums = util.ChecksumJob(filepath) if rank == 0 else None        # md5 + sha256 of the raw file, overlapped (util.py:35-50)
    eng = None
    blocks = None
    err = None
    reads = sum_len = total_kmers = 0
    min_len, max_len = 1 << 62, 0
    try:
        eng = Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP if replace_with_none else KDB_N_EXPAND, device=device)
        for name, v in (engine_opts or {}).items():
            eng.set_option(name, v)
        reads, sum_len, min_len, max_len, blocks = parse._feed_shard(eng, filepath, rank, world, block_bytes)
        _, total_kmers, _ = eng.finish(copy=False)                   # this rank's shard: Sum == emitted holds here
    except BaseException as e:  # noqa: BLE001 - re-raised on every rank by _agree_or_raise
        err = e
    try:
        _agree_or_raise(err, coll_dev, group)
        (reads, sum_len, total_kmers), (neg_min, max_len) = reduce_scalars(
            {"sum": [reads, sum_len, total_kmers], "max": [-min_len, max_len], "device": coll_dev}, group)
        if backend == "nccl":
            reduce_counts(eng, dst=0, group=group)
            counts = None
        else:                       # CPU collectives (tests): the vector crosses to the host first
            t = torch.from_numpy(eng.table_stats()[0].view(np.int64))
            reduce_vector(t, dst=0, group=group)
            counts = t.numpy().view(np.uint64)
        if rank != 0:
            return None, None, None
        if reads == 0:
            raise ValueError("no sequence records found in '{0}'".format(filepath))
        if counts is None:
            counts, vec_sum, unique = eng.table_stats()              # rank 0's vector now holds the global sum
        else:
            vec_sum, unique = int(counts.sum()), int(np.count_nonzero(counts))
        if vec_sum != total_kmers:
            raise RuntimeError("reduced vector sums to {0} but the ranks emitted {1} k-mers".format(vec_sum, total_kmers))
    finally:
        if eng is not None:
            eng.close()                                  # (syncs: nothing reads the reader's ring any more)
        if blocks is not None:
            blocks.release()
    md5, sha256 = sums.result()
    nullomers = np.flatnonzero(counts == 0).astype("uint64")
    meta = parse._file_metadata(filepath, k, md5, sha256, int(reads), total_kmers, unique, int(-neg_min), int(max_len), int(sum_len))
    return counts, meta, nullomers
