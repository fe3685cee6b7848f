"""CPU, world_size 2, gloo: the N>1 path -- shard, count per rank, one SUM reduce -- gives the single-rank vector."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kmerdb_amd import distributed, synth
    from oracle import kmer_oracle
    bases, offsets = synth.reads(3000, 150, seed=11, p_n=0.002)
    r0, r1 = distributed.shard_bounds(len(offsets) - 1, rank, world)
    o = offsets[r0:r1 + 1]
    # each rank counts ITS shard (the oracle stands in for the GPU engine on this CPU-only box)
    mine, total = kmer_oracle.c_count(bases[int(o[0]):int(o[-1])], o - o[0], k, True, kmer_oracle.N_DROP)
    t = torch.from_numpy(mine.view(np.int64).copy())
    distributed.reduce_vector(t, dst=0)
    (reads, tot), (mx,) = distributed.reduce_scalars({"sum": [r1 - r0, total], "max": [int(np.diff(o.astype(np.int64)).max())]})
    if rank == 0:
        want, want_total = kmer_oracle.c_count(bases, offsets, k, True, kmer_oracle.N_DROP)
        ok = np.array_equal(t.numpy().view(np.uint64), want) and tot == want_total and reads == 3000 and mx == 150
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_rank(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 9, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"


def test_sharding_partitions_everything():
    from kmerdb_amd import distributed
    for n in (0, 1, 7, 10_000_001):
        for w in (1, 2, 3, 8):
            b = [distributed.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    owners = [distributed.block_owner(i, 4) for i in range(16)]
    assert sorted(set(owners)) == [0, 1, 2, 3] and owners.count(0) == 4
