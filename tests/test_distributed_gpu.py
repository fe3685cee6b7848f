"""GPU (-m gpu): the N > 1 path with the REAL engine.  Two fresh rank processes share device 0 (a one-GPU box), so the
collectives run on gloo; everything else -- sharded reading, per-rank engines, error agreement, the chunked reduce,
rank 0 reading the reduced vector with table_stats -- is the code an 8-GPU node runs over RCCL.
Reads shard per record (kmerdb/parse.py:128-137) and vectors sum (kmerdb/__init__.py:1890); SURVEY 8(e)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, path, k, rwn, canon, block, out_dir, opts=None, timeout=600):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, path, str(k),
                               "1" if rwn else "0", "1" if canon else "0", str(block), str(out_dir), "gloo", json.dumps(opts or {})], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=timeout) == 0
    return [json.load(open(os.path.join(out_dir, f"rank{r}.json"))) for r in range(world)]


@pytest.mark.parametrize("fname,k,rwn,canon,block", [("inputs/reads150.fq", 12, True, True, 20000),       # ~12 blocks of the 240 KB file
                                                     ("inputs/ragged_n.fq", 9, False, False, 5000),      # N expansion, ragged records
                                                     ("inputs/reads150.fq.gz", 13, True, True, 30000),   # gzip stream, wide path
                                                     ("ref_data/sample.fa", 8, False, True, 40000)])     # FASTA
def test_two_ranks_one_gpu_equal_the_oracle(gpu_engine_cls, oracle, tmp_path, fname, k, rwn, canon, block):
    path = os.path.join(GOLDEN, fname)
    res = _run_ranks(2, path, k, rwn, canon, block, tmp_path)
    assert all("error" not in r for r in res), res
    assert res[1]["none"] is True
    recs = [s for _, s in oracle.read_records(path)]
    bases, offsets = oracle.pack_records(recs)
    want, want_total = oracle.c_count(bases, offsets, k, canon, oracle.N_DROP if rwn else oracle.N_EXPAND)
    got = np.load(tmp_path / "counts.npy")
    assert np.array_equal(got, want)
    meta = res[0]["meta"]
    assert meta["total_kmers"] == want_total and meta["total_reads"] == len(recs)
    assert meta["unique_kmers"] == int(np.count_nonzero(want)) and meta["nullomers"] == res[0]["nullomers"]
    # and the single-process parsefile agrees on every metadata field
    from kmerdb_amd import parse
    _, meta1, _ = parse.parsefile(path, k, replace_with_none=rwn, canonicalize=canon)
    assert meta == meta1


def test_three_ranks_k15_deferred_flush_and_failure(gpu_engine_cls, oracle, tmp_path):
    """k = 15 (two-level path, histogram pass deferred until the sync inside reduce_counts / table_stats), three ranks;
    then a file with a short record: every rank must come out with an error instead of hanging in the reduce."""
    path = os.path.join(GOLDEN, "inputs", "reads150.fq")
    (tmp_path / "a").mkdir()
    res = _run_ranks(3, path, 15, True, True, 30000, tmp_path / "a")
    assert all("error" not in r for r in res), res
    recs = [s for _, s in oracle.read_records(path)]
    ids = np.concatenate([oracle.c_shred(r, 15, True, oracle.N_DROP)[0] for r in recs])
    uniq, cnt = np.unique(ids, return_counts=True)
    z = np.load(tmp_path / "a" / "counts_sparse.npz")
    assert np.array_equal(z["ids"], uniq) and np.array_equal(z["cnt"], cnt.astype(np.uint64))
    assert res[0]["meta"]["total_kmers"] == ids.size
    (tmp_path / "b").mkdir()
    res = _run_ranks(2, os.path.join(GOLDEN, "inputs", "short_read.fq"), 8, True, True, 1 << 20, tmp_path / "b")
    assert all("error" in r for r in res), res
    assert any(r["error"].startswith("RankFailed") for r in res) and any(r["error"].startswith("ValueError") for r in res)


def _bench(args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KDB_BENCH_ALL_ON_DEVICE0="1")
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # stdout is the one JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("shape", ["auto", "rs_gather", "a2a_gather"])
def test_bench_launches_its_own_two_ranks_k12(gpu_engine_cls, shape):
    """`python bench.py --gpus 2` without a launcher: two fresh rank processes (both on device 0 here, collectives on gloo),
    per-rank engines, the untimed probe of the reduce shapes, one reduce of the vector inside the timed region, rank 0
    checks Sum(counts) of the reduced vector against every window of every rank's steps (bench.py asserts it)."""
    d = _bench(["--gpus", "2", "--backend", "gloo", "--k", "12", "--reads", "300000", "--steps", "3", "--warmup", "1",
                "--no-cpu-baseline", "--no-extra-regions", "--reduce-shape", shape], 600)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert [r["world_size_seen"] for r in d["per_rank"]] == [2, 2]
    assert d["reduce_calls"] == 1 and d["reduce_ms"] > 0 and d["reduce_shape"] in ("ring", "rs_gather", "a2a_gather")
    assert shape == "auto" or d["reduce_shape"] == shape
    assert set(d["reduce_probe"]["ms"]) == {"ring", "rs_gather", "a2a_gather"} and d["reduce_probe"]["used"] == d["reduce_shape"]
    assert abs(d["value"] - 2 * 3 * 300000 * 139 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3      # whole-job k-mers / max-over-ranks time
    assert len(d["vector_sha256"]) == 64
    # what the first run on a real node has to be readable from without a re-run: counting and the reduce apart, per rank and for
    # the job; the reduce at the probe's size (every shape, GB/s, the error text of a shape that threw) and at the real vector's
    assert all(r["count_ms"] > 0 and r["reduce_ms"] > 0 and r["elapsed_ms"] >= r["count_ms"] for r in d["per_rank"])
    assert d["count_only_ms_per_step"] > 0 and d["reduce_vector_bytes"] == 8 * 4 ** 12 and d["reduce_gbs"] > 0
    assert 0 < d["reduce_share_of_timed_region"] < 1
    assert set(d["reduce_probe"]["gbs"]) == {"ring", "rs_gather", "a2a_gather"} and all(v and v > 0 for v in d["reduce_probe"]["gbs"].values())
    assert d["reduce_probe"]["errors_on_rank0"] is None and d["reduce_probe"]["bytes"] == 8 * 4 ** 12


def test_bench_two_ranks_at_config_4_shape_k17(gpu_engine_cls):
    """BASELINE config 4's control flow on one device: k = 17, two ranks with a 128 GiB vector each, the two-level path with
    its deferred histogram pass, 128 chunked reduce calls (few reads: the reduce crosses gloo here, xGMI on a real node)."""
    d = _bench(["--gpus", "2", "--backend", "gloo", "--k", "17", "--reads", "200000", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-extra-regions"], 900)
    assert d["n_gpus"] == 2 and [r["world_size_seen"] for r in d["per_rank"]] == [2, 2]
    assert d["reduce_calls"] == 128 and d["reduce_shape"] == "ring"
    assert d["config"]["k"] == 17 and d["reduce_vector_bytes"] == 8 * 4 ** 17 and d["reduce_gbs"] > 0
    # the collectives' first contact comes before the arena sizes itself, and the arena leaves the reduce's
    # scratch and RCCL's head-room free: reserve = 1 GiB to receive into + a shard of it + 2 GiB, and budget + reserve <= what was free
    assert d["reduce_probe"]["when"].startswith("before the first batch")
    for r in d["per_rank"]:
        assert r["reserve_bytes"] == (1 << 30) + (1 << 29) + (2 << 30)
        assert r["free_hbm_when_the_arena_was_sized"] > 0 and r["arena_budget_bytes"] + r["reserve_bytes"] <= r["free_hbm_when_the_arena_was_sized"]
        assert r["free_hbm_before_reduce"] > 0
