"""GPU (-m gpu): BASELINE.json configs 2, 3, 5 (and one rank's shard of config 4) at FULL size on one MI355X, gated as
SURVEY 8(d) prescribes: Sum(counts) == n_reads * (151 - k) for the whole job; the whole vector against the oracle for
config 2; a 1 M-read prefix (counted on its own by the same engine path) against the oracle for configs 3 and 5;
sampled reads of the prefix id by id for config 4 (a 4^17 host vector is 128 GiB).  On top of that every full-size vector
is compared WHOLE with the one the direct-atomics path (algo 1: one 64-bit global atomic per k-mer run, no rings, no pages, no
LDS histogram; pinned against the oracle on its own by test_gpu_parity / test_gpu_fuzz) builds from the same reads: bin by bin on
the device (configs 3 and 5), through a position-weighted checksum (config 4: two 128 GiB vectors do not fit one device)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L = 150


def _reads(n, seed):
    import torch
    from kmerdb_amd import synth
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(synth.SEED0 + seed)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d = torch.empty(n * L, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, n * L, step):
        e = min(n * L, s + step)
        d[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
    o = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    return d, o


def _checksum(t):
    """Sum over i of counts[i] * mix(i) mod 2^64, on the device (int64 arithmetic wraps): equal for equal vectors, and a count that
    moved to another bin, or changed, changes it."""
    import torch
    acc, step = 0, 1 << 27
    for s in range(0, t.numel(), step):
        e = min(t.numel(), s + step)
        idx = torch.arange(s, e, device=t.device, dtype=torch.int64)
        h = (idx * -7046029254386353131 + 7146057691288625177) ^ (idx >> 17)
        acc = (acc + int((t[s:e] * h).sum().item())) & 0xFFFFFFFFFFFFFFFF
        del idx, h
    return acc


def _threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 8


def test_config2_k12_10m_reads_whole_vector_equals_the_oracle(gpu_engine_cls, oracle):
    n, k = 10_000_000, 12
    d, o = _reads(n, 2)
    with gpu_engine_cls(k) as eng:
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        got, total, unique = eng.finish()
    assert total == n * (L - k + 1) == int(got.sum())
    hb = d.cpu().numpy() & 0x7F
    ho = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    want, want_total = oracle.c_count(hb, ho, k, True, oracle.N_DROP, nthreads=_threads())
    assert want_total == total and unique == int(np.count_nonzero(want))
    assert np.array_equal(got, want)


def test_config3_k15_100m_reads(gpu_engine_cls, oracle):
    import torch
    n, k, m = 100_000_000, 15, 1_000_000
    d, o = _reads(n, 3)
    with gpu_engine_cls(k) as eng:
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        _, total, _ = eng.finish(copy=False)
        t = eng.table_tensor()
        assert total == n * (L - k + 1) == int(t.sum().item())
        # the prefix's counts are a lower bound of the whole job's, bin by bin
        hb = d[:m * L].cpu().numpy() & 0x7F
        ho = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
        want, want_total = oracle.c_count(hb, ho, k, True, oracle.N_DROP, nthreads=_threads())
        nz = np.flatnonzero(want)
        full_at = t[torch.as_tensor(nz, device=t.device)].cpu().numpy().view(np.uint64)
        assert np.all(full_at >= want[nz])
        # the whole 8 GiB vector, bin by bin, against the direct-atomics path on the same 100 M reads
        with gpu_engine_cls(k, algo=1) as eng1:
            eng1.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
            _, total1, _ = eng1.finish(copy=False)
            assert total1 == total and torch.equal(eng1.table_tensor(), t)
    with gpu_engine_cls(k) as eng:                      # the prefix on its own: the whole 8 GiB vector equals the oracle's
        eng.submit_device(d.data_ptr(), m * L, o.data_ptr(), m)
        got, total, unique = eng.finish()
    assert total == want_total == m * (L - k + 1) and unique == nz.size
    assert np.array_equal(got, want)


def test_config5_k12_graph_50m_reads_adjacency_histogram(gpu_engine_cls, oracle):
    """The weighted edge list of `kmerdb graph` is the forward 13-mer histogram (graph.py:108-216; kmerdb_amd/graph.py)."""
    n, k, m = 50_000_000, 12, 1_000_000
    d, o = _reads(n, 5)
    with gpu_engine_cls(k + 1, canonicalize=False) as eng:
        eng.set_option("min_len", k)
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        _, total, _ = eng.finish(copy=False)
        assert total == n * (L - k) == int(eng.table_tensor().sum().item())
        import torch
        with gpu_engine_cls(k + 1, canonicalize=False, algo=1) as eng1:     # the whole vector against the direct-atomics path
            eng1.set_option("min_len", k)
            eng1.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
            _, total1, _ = eng1.finish(copy=False)
            assert total1 == total and torch.equal(eng1.table_tensor(), eng.table_tensor())
    hb = d[:m * L].cpu().numpy() & 0x7F
    ho = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    want, want_total = oracle.c_count_edges(hb, ho, k)
    with gpu_engine_cls(k + 1, canonicalize=False) as eng:
        eng.set_option("min_len", k)
        eng.submit_device(d.data_ptr(), m * L, o.data_ptr(), m)
        got, total, _ = eng.finish()
    assert total == want_total == m * (L - k)
    assert np.array_equal(got, want)


def test_config4_k17_one_ranks_shard(gpu_engine_cls, oracle):
    """One rank's share of config 4: 62.5 M of the 500 M reads into a 128 GiB vector (the reduce over 8 ranks is
    covered by tests/test_distributed_gpu.py and bench.py --gpus N)."""
    import torch
    n, k, m = 62_500_000, 17, 3000
    d, o = _reads(n, 4)
    with gpu_engine_cls(k) as eng:
        eng.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        _, total, _ = eng.finish(copy=False)
        assert total == n * (L - k + 1)
        t = eng.table_tensor()
        hb = d[:m * L].cpu().numpy() & 0x7F
        ids = np.concatenate([oracle.c_shred(bytes(hb[r * L:(r + 1) * L]), k, True, oracle.N_DROP)[0] for r in range(m)])
        uniq, cnt = np.unique(ids, return_counts=True)
        at = t[torch.as_tensor(uniq.astype(np.int64), device=t.device)].cpu().numpy().view(np.uint64)
        assert np.all(at >= cnt.astype(np.uint64))
        chk = _checksum(t)
        del t
    with gpu_engine_cls(k, algo=1) as eng1:             # the whole 128 GiB vector of the direct-atomics path: the same checksum
        eng1.submit_device(d.data_ptr(), n * L, o.data_ptr(), n)
        _, total1, _ = eng1.finish(copy=False)
        assert total1 == total and _checksum(eng1.table_tensor()) == chk
    with gpu_engine_cls(k) as eng:                      # those reads alone: exact, including unique
        eng.submit_device(d.data_ptr(), m * L, o.data_ptr(), m)
        _, total, unique = eng.finish(copy=False)
        t = eng.table_tensor()
        at = t[torch.as_tensor(uniq.astype(np.int64), device=t.device)].cpu().numpy().view(np.uint64)
    assert total == ids.size and unique == uniq.size and np.array_equal(at, cnt.astype(np.uint64))


def _ragged(n, lo, hi, p_n, seed):
    """Reads of lengths uniform in lo..hi with a fraction p_n of N's, generated on the device (bench.py's ragged batch)."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (n,), generator=g, device=dev, dtype=torch.int64)
    o = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=o[1:])
    nbytes = int(o[-1].item())
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d = lut[torch.randint(0, 4, (nbytes,), generator=g, device=dev, dtype=torch.uint8).long()]
    d[torch.rand(nbytes, generator=g, device=dev) < p_n] = 78
    torch.cuda.synchronize()
    return d, o, nbytes


@pytest.mark.parametrize("p_n", [0.0005, 0.005, 0.05])
def test_ragged_reads_with_n_at_scale(gpu_engine_cls, oracle, p_n):
    """The shape of a real FASTQ at a size where every scatter workgroup walks dozens of tiles (the seeded and fuzz cases hold a tile or
    two per workgroup): 1.2 M reads of 35..150 bases with N's, N-expansion mode (the reference CLI's default; the tile-level N lists of
    DESIGN.md section 4, at 5 % N the dense path) and N-drop mode.  k = 8, 12, 13: the whole vector against the oracle; k = 15:
    the whole 8 GiB vector against the direct-atomics path's."""
    import torch
    n = 1_200_000 if p_n < 0.01 else 300_000
    d, o, nbytes = _ragged(n, 35, L, p_n, 4242 + int(p_n * 1e4))
    hb = d.cpu().numpy()
    ho = o.cpu().numpy().astype(np.uint64)
    for k in (8, 12, 13):
        for canon, omode, gmode in ((True, oracle.N_EXPAND, 1), (False, oracle.N_DROP, 0)):
            with gpu_engine_cls(k, canonicalize=canon, n_mode=gmode) as eng:
                eng.submit_device(d.data_ptr(), nbytes, o.data_ptr(), n)
                got, total, unique = eng.finish()
            want, want_total = oracle.c_count(hb, ho, k, canon, omode, nthreads=_threads())
            assert total == want_total and unique == int(np.count_nonzero(want)), (k, canon, p_n)
            assert np.array_equal(got, want), (k, canon, p_n)
    k = 15
    with gpu_engine_cls(k, n_mode=1) as eng, gpu_engine_cls(k, n_mode=1, algo=1) as eng1:
        # (one after the other: the direct-atomics kernel marks the record starts of a ragged batch in the caller's buffer while it runs)
        eng.submit_device(d.data_ptr(), nbytes, o.data_ptr(), n)
        _, total, unique = eng.finish(copy=False)
        eng1.submit_device(d.data_ptr(), nbytes, o.data_ptr(), n)
        _, total1, unique1 = eng1.finish(copy=False)
        assert (total, unique) == (total1, unique1) and torch.equal(eng.table_tensor(), eng1.table_tensor())
