"""CPU, world_size 2, gloo: the N>1 path -- shard the file by record blocks, count per rank, agree on errors, one
chunked SUM reduce -- gives the single-rank vector.  `kmerdb_amd.distributed.parsefile_distributed` runs unchanged;
only the Engine class is replaced by an oracle-backed stand-in (there is no GPU here).  The same function runs
against the real engine in tests/test_distributed_gpu.py."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleEngine:
    """The slice of kmerdb_amd.Engine that parsefile_distributed uses, computed by the CPU oracle (test stand-in)."""

    def __init__(self, k, canonicalize=True, n_mode=0, device=0):
        from oracle import kmer_oracle
        self.o, self.k, self.canon, self.n_mode = kmer_oracle, k, canonicalize, n_mode
        self.counts = np.zeros(4 ** k, dtype=np.uint64)
        self.total = 0
        self.pinned_calls = 0

    def set_option(self, name, v):
        pass

    def submit(self, bases, offsets):
        c, t = self.o.c_count(np.array(bases), np.array(offsets), self.k, self.canon, self.n_mode)
        self.counts += c
        self.total += t

    submit_pinned = submit

    def finish(self, copy=True):
        return (self.counts.copy() if copy else None), self.total, int(np.count_nonzero(self.counts))

    def table_stats(self, copy=True):
        return (self.counts.copy() if copy else None), int(self.counts.sum()), int(np.count_nonzero(self.counts))

    def close(self):
        pass


def _worker(rank, world, port, path, k, rwn, canon, block_bytes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kmerdb_amd.engine
    from kmerdb_amd import distributed
    kmerdb_amd.engine.Engine = OracleEngine
    distributed.REDUCE_CHUNK_BYTES = 1 << 12            # 4^9 * 8 B = 2 MiB vector -> 512 chunked collectives
    res = {"rank": rank}
    try:
        counts, meta, nullomers = distributed.parsefile_distributed(path, k, replace_with_none=rwn, canonicalize=canon,
                                                                    device=0, block_bytes=block_bytes)
        if rank == 0:
            np.save(os.path.join(out_dir, "counts.npy"), counts)
            res["meta"] = meta
            res["nullomers"] = int(len(nullomers))
        else:
            res["none"] = counts is None and meta is None and nullomers is None
    except Exception as e:  # noqa: BLE001 - reported to the parent
        res["error"] = type(e).__name__
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fname,k,rwn,canon,block", [("inputs/reads150.fq", 9, True, True, 4096),
                                                     ("inputs/ragged_n.fq", 7, False, False, 1500),
                                                     ("inputs/reads150.fq.gz", 8, True, True, 10000),
                                                     ("inputs/contigs.fa", 6, True, True, 3000)])
def test_two_rank_parsefile_distributed_equals_single_rank(tmp_path, oracle, fname, k, rwn, canon, block):
    path = os.path.join(GOLDEN, fname)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, path, k, rwn, canon, block, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert "error" not in r0 and "error" not in r1, (r0, r1)
    assert r1["none"] is True
    recs = [s for _, s in oracle.read_records(path)]
    bases, offsets = oracle.pack_records(recs)
    want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_DROP if rwn else oracle.N_EXPAND)
    got = np.load(tmp_path / "counts.npy")
    assert got.dtype == np.uint64 and np.array_equal(got, want)
    meta = r0["meta"]
    lens = [len(r) for r in recs]
    assert meta["total_kmers"] == want_total and meta["total_reads"] == len(recs)
    assert meta["unique_kmers"] == int(np.count_nonzero(want)) and meta["nullomers"] == 4 ** k - meta["unique_kmers"] == r0["nullomers"]
    assert (meta["min_read_length"], meta["max_read_length"], meta["avg_read_length"]) == (min(lens), max(lens), int(sum(lens) / len(lens)))
    # the golden metadata of the reference for the same file (when that case exists) agrees on the shared keys
    for c in json.load(open(os.path.join(GOLDEN, "parsefile.json"))):
        if c["file"] == fname and c["k"] == k and c["replace_with_none"] == rwn and c["canonicalize"] == canon:
            for key in ("md5", "sha256", "total_reads", "total_kmers", "unique_kmers", "nullomers"):
                assert meta[key] == c["metadata"][key], key


def test_a_failing_rank_takes_every_rank_out_instead_of_hanging(tmp_path, oracle):
    """Records shorter than k raise on the rank that owns them; the other rank must leave too (RankFailed), before the
    vector reduce."""
    path = os.path.join(GOLDEN, "inputs", "short_read.fq")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, path, 8, True, True, 1 << 20, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0.get("error") in ("OracleError", "ValueError") and r1.get("error") == "RankFailed", (r0, r1)


def _reduce_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kmerdb_amd import distributed
    res = {}
    n = 4 ** 7 + 5                                   # not a multiple of the world size: the leftover goes through a plain reduce
    for shape in distributed.REDUCE_SHAPES:
        for chunk_bytes in (1 << 30, 4096, 8 * world):
            g = torch.Generator().manual_seed(1000 * rank + 7)
            t = torch.randint(0, 1 << 40, (n,), generator=g, dtype=torch.int64)
            calls = distributed.reduce_vector(t, dst=0, chunk_bytes=chunk_bytes, shape=shape)
            if rank == 0:
                want = sum(torch.randint(0, 1 << 40, (n,), generator=torch.Generator().manual_seed(1000 * r + 7), dtype=torch.int64) for r in range(world))
                res[f"{shape}/{chunk_bytes}"] = bool(torch.equal(t, want)) and calls >= 1
    chosen, ms = distributed.probe_reduce_shapes(None, None, nbytes=1 << 16, repeats=1)
    # first contact (round 5): every shape once before anything else, the choice kept for reduce_counts; and what an engine must leave free
    fc, fc_ms = distributed.first_contact(None, None, nbytes=1 << 16)
    if rank == 0:
        res["probe"] = [chosen, ms]
        res["first_contact"] = [fc, fc_ms, distributed._shape_choice.get((world, "gloo", "None")), distributed.reduce_reserve_bytes(world), distributed.reduce_reserve_bytes(1)]
        json.dump(res, open(os.path.join(out_dir, "reduce.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_every_reduce_shape_gives_the_same_sum(tmp_path, world):
    """The end-of-job reduce in its three shapes (one reduce per chunk; reduce-scatter + gather; all-to-all + local sum +
    gather) over gloo: identical bits, whatever the chunking, also when the world size does not divide the vector."""
    mp.spawn(_reduce_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = json.load(open(tmp_path / "reduce.json"))
    chosen, ms = res.pop("probe")
    fc, fc_ms, kept, reserve, reserve1 = res.pop("first_contact")
    assert fc in fc_ms and kept == fc and set(fc_ms) == {"ring", "rs_gather", "a2a_gather"}
    assert reserve == (1 << 30) + (1 << 30) // world + (2 << 30) and reserve1 == 0
    assert len(res) == 9 and all(res.values()), res
    assert chosen in ms and all(v is not None and v >= 0 for v in ms.values()), (chosen, ms)


def test_sharding_partitions_everything():
    from kmerdb_amd import distributed
    for n in (0, 1, 7, 10_000_001):
        for w in (1, 2, 3, 8):
            b = [distributed.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    owners = [distributed.block_owner(i, 4) for i in range(16)]
    assert sorted(set(owners)) == [0, 1, 2, 3] and owners.count(0) == 4


def _collect(it):
    out = []
    for b, o, ids in it:
        o = o.astype(np.int64)
        out += [(ids[r], bytes(b[o[r]:o[r + 1]])) for r in range(len(o) - 1)]
    return out


@pytest.mark.parametrize("fname", ["inputs/reads150.fq", "inputs/ragged_n.fq", "inputs/reads150.fq.gz", "inputs/contigs.fa",
                                   "ref_data/sample.fa"])
def test_sharded_reader_partitions_the_records_exactly(fname):
    """Every record is read by exactly one rank, whatever the world size and block size; '@' at the start of a quality
    line is not mistaken for a header."""
    from kmerdb_amd import reader
    path = os.path.join(GOLDEN, fname)
    want = _collect(reader.iter_blocks(path, want_ids=True))
    for world in (1, 2, 3, 8):
        for B in (700, 4096, 1 << 16, 1 << 27):
            got = []
            for r in range(world):
                got += _collect(reader.iter_blocks_sharded(path, r, world, want_ids=True, block_bytes=B))
            assert sorted(got) == sorted(want), (fname, world, B)


def test_sharded_reader_on_adversarial_fastq(tmp_path):
    rng = np.random.Generator(np.random.PCG64(5))
    recs = []
    for i in range(2000):
        n = int(rng.integers(1, 200))
        seq = "".join(rng.choice(list("ACGTN"), size=n))
        qual = "".join(rng.choice(list("@+I>#"), size=n))          # quality lines that look like headers / separators
        recs.append(f"@r{i} +x\n{seq}\n+\n{qual}\n")
    p = str(tmp_path / "adv.fq")
    open(p, "w").write("".join(recs)[:-1])                          # no trailing newline
    from kmerdb_amd import reader
    want = _collect(reader.iter_blocks(p, want_ids=True))
    assert len(want) == 2000
    for world in (2, 5):
        for B in (257, 1000, 30000):
            got = []
            for r in range(world):
                got += _collect(reader.iter_blocks_sharded(p, r, world, want_ids=True, block_bytes=B))
            assert sorted(got) == sorted(want), (world, B)
