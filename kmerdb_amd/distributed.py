"""Multi-GPU: reads shard across ranks, one reduce of the 4^k vector at the end.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  The count
vector is a commutative integer sum over records (kmerdb/parse.py:128-137), so each rank counts its own
blocks into its own engine with no data-path collective, and a single SUM reduce over xGMI yields the
single-GPU vector bit for bit (uint64 viewed as int64: identical bits mod 2^64).
"""
import numpy as np


def block_owner(block_index, world_size):
    """Round-robin block -> rank map (each block is a run of whole records)."""
    return block_index % world_size


def shard_bounds(nreads, rank, world_size):
    """Contiguous [r0, r1) shard of `nreads` records for `rank`; shards partition range(nreads)."""
    return nreads * rank // world_size, nreads * (rank + 1) // world_size


def reduce_vector(t, dst=0, group=None):
    """Sum an int64 count vector (torch tensor, CPU or CUDA) across ranks onto `dst`; in place."""
    import torch.distributed as dist
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return t


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


def reduce_counts(engine, dst=0, group=None):
    """Reduce an Engine's HBM count vector onto rank `dst` (in place on the device). Call after engine.sync()."""
    import torch
    engine.sync()
    t = engine.table_tensor()
    torch.cuda.synchronize(t.device)
    reduce_vector(t, dst=dst, group=group)
    torch.cuda.synchronize(t.device)
    return t


def parsefile_distributed(filepath, k, replace_with_none=True, canonicalize=True, device=None, group=None):
    """parse.parsefile over all ranks of the default process group: rank r counts blocks r, r+W, ...;
    rank 0 returns (counts, file_metadata, nullomer_array) like kmerdb/parse.py:90-163, other ranks (None, None, None)."""
    import torch
    import torch.distributed as dist
    from . import reader, util
    from .engine import Engine, KDB_N_DROP, KDB_N_EXPAND
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = torch.cuda.current_device()
    eng = Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP if replace_with_none else KDB_N_EXPAND, device=device)
    try:
        reads = sum_len = 0
        min_len, max_len = 1 << 62, 0
        for i, (bases, offsets, _) in enumerate(reader.iter_blocks(filepath)):
            if block_owner(i, world) != rank or len(offsets) < 2:
                continue
            lens = np.diff(offsets.astype(np.int64))
            reads += len(lens)
            sum_len += int(lens.sum())
            min_len, max_len = min(min_len, int(lens.min())), max(max_len, int(lens.max()))
            eng.submit(bases, offsets)
        _, total_kmers, _ = eng.finish(copy=False)
        (reads, sum_len, total_kmers), (neg_min, max_len) = reduce_scalars(
            {"sum": [reads, sum_len, total_kmers], "max": [-min_len, max_len], "device": f"cuda:{device}"}, group)
        reduce_counts(eng, dst=0, group=group)
        if rank != 0:
            return None, None, None
        if reads == 0:
            raise ValueError("no sequence records found in '{0}'".format(filepath))
        counts, _, unique = eng.finish()          # rank 0's vector now holds the global sum
    finally:
        eng.close()
    md5, sha256 = util.checksum(filepath)
    nullomers = np.flatnonzero(counts == 0).astype("uint64")
    meta = {"filename": filepath, "md5": md5, "sha256": sha256, "total_reads": int(reads), "total_kmers": int(total_kmers),
            "unique_kmers": int(unique), "nullomers": int(4 ** k - unique), "min_read_length": int(-neg_min),
            "max_read_length": int(max_len), "avg_read_length": int(sum_len / reads)}
    return counts, meta, nullomers
