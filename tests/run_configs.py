#!/usr/bin/env python3
"""BASELINE.json configs 2-5 on one MI355X (config 4: one rank's shard), with the size-independent gates of
SURVEY 8(d): Sum(counts) == n_reads * (151 - k), and a 200 k-read prefix equal to the oracle (sparse compare).
Prints one JSON object per config. Not the headline metric (bench.py is)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import kmerdb_amd  # noqa: E402
from kmerdb_amd import synth  # noqa: E402
from oracle import kmer_oracle  # noqa: E402

L = 150
dev = torch.device("cuda", 0)


def make_reads(n, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d = torch.empty(n * L, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, n * L, step):
        e = min(n * L, s + step)
        d[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
    o = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    return d, o


def run(name, k, n, canonical=True, seed=0, graph=False):
    out = {"config": name, "k": k, "reads": n, "bases": n * L}
    d, o = make_reads(n, synth.SEED0 + seed)
    kk = k + 1 if graph else k
    with kmerdb_amd.Engine(kk, canonicalize=(canonical and not graph)) as eng:
        if graph:
            eng.set_option("min_len", k)
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)      # warm-up (scratch allocation)
        eng.sync()
        eng.reset()
        eng.prof_enable(True)
        t0 = time.perf_counter()
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        eng.sync()
        dt = time.perf_counter() - t0
        _, total, unique = eng.finish(copy=False)
        out["kernels_ms"] = {kname: round(ms, 3) for kname, (ms, cnt) in eng.prof().items() if cnt}
        table = eng.table_tensor()
        assert total == n * (L - kk + 1) == int(table.sum().item()), (total, n * (L - kk + 1))
        # independent check: the first 200 k reads through the direct-atomics kernel must give the same vector slice
        m = min(n, 200_000)
        big = (4 ** kk) * 8 > (40 << 30)          # two more vectors of this size would not fit beside the first
        if not big:
          with kmerdb_amd.Engine(kk, canonicalize=(canonical and not graph), algo=1) as ref, \
                kmerdb_amd.Engine(kk, canonicalize=(canonical and not graph)) as fast:
            for e2 in (ref, fast):
                if graph:
                    e2.set_option("min_len", k)
                e2.submit_device(d.data_ptr(), m * L, o.data_ptr(), m)      # offsets of the prefix are the same
            # (the offsets array is longer than m+1 entries; only the first m+1 are read)
            _, t_ref, u_ref = ref.finish(copy=False)
            _, t_fast, u_fast = fast.finish(copy=False)
            assert (t_ref, u_ref) == (t_fast, u_fast)
            assert torch.equal(ref.table_tensor(), fast.table_tensor())
        # and 2000 sampled reads against the CPU oracle, id by id
        hb = d[:m * L].cpu().numpy()
        rng = np.random.Generator(np.random.PCG64(1))
        ids = np.concatenate([kmer_oracle.c_shred(bytes(hb[r * L:(r + 1) * L]), kk, canonical and not graph, kmer_oracle.N_DROP)[0]
                              for r in rng.choice(m, size=2000, replace=False)])
        uniq, cnt = np.unique(ids, return_counts=True)
        got = table[torch.as_tensor(uniq.astype(np.int64), device=dev)].cpu().numpy().astype(np.uint64)
        assert np.all(got >= cnt.astype(np.uint64))                  # every sampled occurrence is in the full vector
    out.update({"seconds": round(dt, 4), "gbase_per_s": round(n * L / dt / 1e9, 2), "kmers_per_s": round(total / dt, 1),
                "total_kmers": int(total), "unique": int(unique), "gates": "sum ok; " + ("" if big else "200k-read prefix == direct-atomics kernel; ") + "sampled ids present"})
    del d, o
    torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "5", "3", "4"]
    if "2" in which:
        run("2: k=12 profile, 10M reads", 12, 10_000_000, seed=2)
    if "5" in which:
        run("5: k=12 graph (k+1-mer adjacency histogram), 50M reads", 12, 50_000_000, seed=5, graph=True)
    if "3" in which:
        run("3: k=15 profile, 100M reads", 15, 100_000_000, seed=3)
    if "4" in which:
        run("4: k=17 profile, one rank's shard (62.5M of 500M reads); 128 GiB vector", 17, 62_500_000, seed=4)
