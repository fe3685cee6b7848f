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


RCCL_HEADROOM_BYTES = 2 << 30


def reduce_reserve_bytes(world_size, chunk_bytes=REDUCE_CHUNK_BYTES):
    """Device memory a rank's engine must leave free for the end-of-job reduce (Engine option "reserve_bytes"): the sharded shapes'
    scratch -- a chunk to receive into and a shard of one (reduce_vector) -- and room for what RCCL allocates at the first collective
    of each kind (channel and peer-to-peer buffers; 2 GiB is generous for eight ranks).  The page arena of k >= 14 sizes itself on
    "85 % of what is free": without the reserve the first reduce of a long job could find the device full."""
    if world_size <= 1:
        return 0
    return int(chunk_bytes) + int(chunk_bytes) // int(world_size) + RCCL_HEADROOM_BYTES


def first_contact(device=None, group=None, nbytes=None):
    """Run every reduce shape once on a scratch tensor NOW -- before the engines have grown their arenas into the free memory: RCCL
    creates its connections and buffers at the first call of each collective, not when the communicator is made.  Also settles which
    shape reduce_counts will use for large vectors.  -> (fastest shape, {shape: ms}); a no-op answer for one rank."""
    import torch.distributed as dist
    W = dist.get_world_size(group)
    if W <= 1:
        return "ring", {}
    chosen, ms = probe_reduce_shapes(device, group, nbytes=int(nbytes) if nbytes else REDUCE_CHUNK_BYTES, repeats=1)
    _shape_choice[(W, dist.get_backend(group), str(device))] = chosen
    return chosen, ms


def block_owner(block_index, world_size):
    """Round-robin block -> rank map (each block is a run of whole records)."""
    return block_index % world_size


def shard_bounds(nreads, rank, world_size):
    """Contiguous [r0, r1) shard of `nreads` records for `rank`; shards partition range(nreads)."""
    return nreads * rank // world_size, nreads * (rank + 1) // world_size


REDUCE_SHAPES = ("ring", "rs_gather", "a2a_gather")


def _reduce_chunk_sharded(c, dst, group, shape, scratch):
    """Sum chunk `c` (numel divisible by the world size) onto rank `dst`, every rank reducing one shard of it:
      rs_gather   reduce_scatter_tensor (rank j ends up with the sum of shard j), then the shards are gathered on `dst`
      a2a_gather  all_to_all_single (rank j receives everybody's shard j: every xGMI link of every GPU carries one
                  shard at the same time), a local sum of the W shards, then the same gather
    The gather lands in place in `c` on `dst`.  `scratch` = {} reused across chunks."""
    import torch
    import torch.distributed as dist
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    sh = c.numel() // W
    key = ("shard", sh, c.device)
    mine = scratch.get(key)
    if mine is None:
        mine = scratch[key] = torch.empty(sh, dtype=c.dtype, device=c.device)
    if shape == "rs_gather":
        dist.reduce_scatter_tensor(mine, c, op=dist.ReduceOp.SUM, group=group)
    else:
        key = ("recv", c.numel(), c.device)
        recv = scratch.get(key)
        if recv is None:
            recv = scratch[key] = torch.empty(c.numel(), dtype=c.dtype, device=c.device)
        dist.all_to_all_single(recv, c, group=group)
        torch.sum(recv.view(W, sh), dim=0, out=mine)
    dist.gather(mine, [c[j * sh:(j + 1) * sh] for j in range(W)] if rank == dst else None, dst=dst, group=group)


def reduce_vector(t, dst=0, group=None, chunk_bytes=REDUCE_CHUNK_BYTES, shape="ring"):
    """Sum an int64 count vector (torch tensor, CPU or CUDA) across ranks onto `dst`; in place on `dst`, chunked.
    shape: "ring" = one reduce per chunk (RCCL's ring/tree: bound by one xGMI link); "rs_gather" / "a2a_gather" = every
    rank sums one shard of the chunk and `dst` gathers the shards (all seven links of a GPU carry traffic;
    see _reduce_chunk_sharded).  Same bits whatever the shape (integer sum).  On ranks other than `dst` the vector is
    scratch afterwards.  -> number of chunks."""
    import torch.distributed as dist
    if shape not in REDUCE_SHAPES:
        raise ValueError("reduce shape {0!r} (one of {1})".format(shape, REDUCE_SHAPES))
    n = t.numel()
    W = dist.get_world_size(group)
    step = max(W, int(chunk_bytes) // t.element_size() // W * W)
    calls = 0
    scratch = {}
    for s in range(0, n, step):
        c = t[s:s + step]
        m = c.numel() // W * W if shape != "ring" and W > 1 else 0
        if m:
            _reduce_chunk_sharded(c[:m], dst, group, shape, scratch)
        if m < c.numel():                       # (ring shape, or the few elements a world size that does not divide 4^k leaves over)
            dist.reduce(c[m:], dst=dst, op=dist.ReduceOp.SUM, group=group)
        calls += 1
    return calls


_shape_choice = {}
last_probe_errors = {}       # shape -> text of the exception the last probe_reduce_shapes call caught on this rank


def probe_reduce_shapes(device=None, group=None, nbytes=256 << 20, repeats=2):
    """Time every reduce shape on a scratch vector (not the count vector: a reduce is destructive) and agree on the
    fastest across ranks.  -> (chosen shape, {shape: ms on the slowest rank, or None if the backend lacks a collective})."""
    import time
    import torch
    import torch.distributed as dist
    W = dist.get_world_size(group)
    n = max(W, nbytes // 8 // W * W)
    t = torch.ones(n, dtype=torch.int64, device=device if device is not None else "cpu")
    cuda = t.is_cuda
    out = {}
    last_probe_errors.clear()
    for shape in REDUCE_SHAPES:
        ms, failed = None, 0
        try:
            for rep in range(repeats + 1):                   # (the first pass opens connections and allocates scratch)
                if cuda:
                    torch.cuda.synchronize(t.device)
                dist.barrier(group)
                t0 = time.perf_counter()
                reduce_vector(t, dst=0, group=group, shape=shape)
                if cuda:
                    torch.cuda.synchronize(t.device)
                dt = (time.perf_counter() - t0) * 1e3
                if rep:
                    ms = dt if ms is None else min(ms, dt)
        except (RuntimeError, NotImplementedError) as e:
            failed = 1
            last_probe_errors[shape] = "{0}: {1}".format(type(e).__name__, e)
        v = torch.tensor([ms if ms is not None else 0.0, float(failed)], dtype=torch.float64, device=t.device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
        out[shape] = None if v[1].item() else round(float(v[0].item()), 3)
    ok = {k: v for k, v in out.items() if v is not None}
    chosen = min(ok, key=ok.get) if ok else "ring"
    return chosen, out


def best_reduce_shape(nbytes, device=None, group=None):
    """The shape reduce_counts uses: small vectors (< 256 MiB: latency, not bandwidth) take the plain reduce; larger ones
    the shape a one-off probe found fastest for this process group."""
    import torch.distributed as dist
    W = dist.get_world_size(group)
    if W <= 2 or nbytes < (256 << 20):
        return "ring"
    key = (W, dist.get_backend(group), str(device))
    if key not in _shape_choice:
        _shape_choice[key] = probe_reduce_shapes(device, group)[0]
    return _shape_choice[key]


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


def reduce_counts(engine, dst=0, group=None, chunk_bytes=REDUCE_CHUNK_BYTES, shape="auto"):
    """Reduce an Engine's HBM count vector onto rank `dst` (in place on the device).  Syncs the engine first.
    After this, rank `dst` must read its vector with Engine.table_stats(), not finish(): the vector now holds
    every rank's counts, which finish()'s Sum(counts) == emitted-by-this-engine check rightly rejects."""
    import torch
    t = engine.table_tensor()          # syncs: submits are asynchronous, and k >= 14 defers its histogram pass
    torch.cuda.synchronize(t.device)
    if shape == "auto":
        shape = best_reduce_shape(t.numel() * 8, t.device, group)
    reduce_vector(t, dst=dst, group=group, chunk_bytes=chunk_bytes, shape=shape)
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
                          block_bytes=None, engine_opts=None, reduce_shape="auto"):
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
    sums = None
    eng = None
    blocks = None
    err = None
    reads = sum_len = total_kmers = 0
    min_len, max_len = 1 << 62, 0
    # one try/finally around both phases: whatever leaves this function -- KeyboardInterrupt and SystemExit included -- closes the
    # engine, releases the reader's ring and joins the checksum threads
    try:
        try:
            if rank == 0:
                sums = util.ChecksumJob(filepath)                        # md5 + sha256 of the raw file, overlapped (util.py:35-50)
            if backend == "nccl" and 8 * 4 ** k >= (256 << 20):
                first_contact(coll_dev, group, nbytes=min(8 * 4 ** k, REDUCE_CHUNK_BYTES))     # RCCL's first collectives, before the arena takes the free memory
            eng = Engine(k, canonicalize=canonicalize is True, n_mode=KDB_N_DROP if replace_with_none else KDB_N_EXPAND, device=device)
            eng.set_option("reserve_bytes", reduce_reserve_bytes(world))
            for name, v in (engine_opts or {}).items():
                eng.set_option(name, v)
            reads, sum_len, min_len, max_len, blocks = parse._feed_shard(eng, filepath, rank, world, block_bytes)
            _, total_kmers, _ = eng.finish(copy=False)                   # this rank's shard: Sum == emitted holds here
        except Exception as e:  # noqa: BLE001 - re-raised on every rank by _agree_or_raise (KeyboardInterrupt / SystemExit pass through to the cleanup below)
            err = e
        _agree_or_raise(err, coll_dev, group)
        (reads, sum_len, total_kmers), (neg_min, max_len) = reduce_scalars(
            {"sum": [reads, sum_len, total_kmers], "max": [-min_len, max_len], "device": coll_dev}, group)
        if backend == "nccl":
            reduce_counts(eng, dst=0, group=group, shape=reduce_shape)
            counts = None
        else:                       # CPU collectives (tests): the vector crosses to the host first
            t = torch.from_numpy(eng.table_stats()[0].view(np.int64))
            reduce_vector(t, dst=0, group=group, shape=best_reduce_shape(t.numel() * 8, None, group) if reduce_shape == "auto" else reduce_shape)
            counts = t.numpy().view(np.uint64)
        if rank != 0:
            return None, None, None
        if reads == 0:
            raise ValueError("no sequence records found in '{0}'".format(filepath))
        nullomers = None
        if counts is None:
            counts, vec_sum, unique = eng.table_stats()              # rank 0's vector now holds the global sum
            nullomers = eng.nullomers(n=4 ** k - unique)             # parse.py:139-140, compacted on the device
        else:
            vec_sum, unique = int(counts.sum()), int(np.count_nonzero(counts))
        if vec_sum != total_kmers:
            raise RuntimeError("reduced vector sums to {0} but the ranks emitted {1} k-mers".format(vec_sum, total_kmers))
    except BaseException:
        if sums is not None:                             # leaving with an error: the checksum threads are joined, their result dropped
            try:
                sums.result()
            except Exception:  # noqa: BLE001
                pass
        raise
    finally:
        if eng is not None:
            eng.close()                                  # (syncs: nothing reads the reader's ring any more)
        if blocks is not None:
            blocks.release()
    md5, sha256 = sums.result()
    if nullomers is None:                                # (CPU collectives: the reduced vector lives on the host)
        nullomers = np.flatnonzero(counts == 0).astype("uint64")
    meta = parse._file_metadata(filepath, k, md5, sha256, int(reads), total_kmers, unique, int(-neg_min), int(max_len), int(sum_len))
    return counts, meta, nullomers
