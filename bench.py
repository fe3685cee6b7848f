#!/usr/bin/env python3
"""bench.py -- k-mers counted per second at k=12 on synthetic 150-bp reads (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one pass of the hot path (record geometry + 2-bit encode + 4^k histogram) over one batch of
10 M synthetic 150-bp reads (BASELINE config 2) that is already resident in HBM.  Each rank (one per
GPU) counts its own batch into its own 4^k uint64 vector (weak scaling, no data-path collective);
for N > 1 the job ends with ONE chunked RCCL reduce of the vector to rank 0 over xGMI, inside the timed
region.  Rank 0 prints one JSON line.

`python bench.py --gpus N` without a launcher starts its own N rank processes (fresh children, created
before anything in this process touches torch or HIP); under torch.distributed.run it uses the ranks it
is given.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="timed steps (500 x ~2.4 ms: a timed region of more than one second)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--k", type=int, default=12)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct atomics, 2 LDS-histogram paths")
    ap.add_argument("--forward", action="store_true", help="do not canonicalize")
    ap.add_argument("--expand", action="store_true", help="N-expansion mode (the reference CLI's default) in the headline region")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning)")
    ap.add_argument("--reduce-shape", default="auto", help="end-of-job reduce: ring | rs_gather | a2a_gather | auto (an untimed probe picks the fastest)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="resident batches (different seeds) the steps rotate through; 0 = 1 for k <= 12 (BASELINE config 2 is ONE 10 M-read batch), "
                         "8 for k = 13..16, 16 for k = 17: the deferred histogram pass of a flush must see distinct reads")
    ap.add_argument("--no-configs", action="store_true", help="skip the other single-GPU BASELINE configs (3, 5, config 4's shard, k = 17 steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-regions", action="store_true", help="skip the H2D-fed / FASTQ end-to-end / other-mode regions")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = size the sample for ~12 s of CPU work")
    return ap.parse_args()


def self_launch(args):
    """No launcher gave us ranks: start N children of this script, one per GPU, and relay their exit status.
    Nothing here imports torch or touches HIP -- the children are ordinary fresh processes."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # relay the ranks' exit status; a rank that dies takes the others with it (they would sit in a collective until its timeout)
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = rc or abs(r)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


def synthetic_batch(torch, dev, n_reads, L, seed):
    """Uniform ACGT reads generated on the device; -> (d_bases uint8[n*L], d_offs int64[n+1])."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nbytes = n_reads * L
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d_bases = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, nbytes, step):
        e = min(nbytes, s + step)
        d_bases[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
    d_offs = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * L
    return d_bases, d_offs


def ragged_batch(torch, dev, n_reads, lo, hi, p_n, seed):
    """Reads of lengths uniform in lo..hi with a fraction p_n of N's, generated on the device; -> (d_bases, d_offs, nbytes)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (n_reads,), generator=g, device=dev, dtype=torch.int64)
    d_offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=d_offs[1:])
    nbytes = int(d_offs[-1].item())
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d_bases = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    step = 1 << 27
    for s in range(0, nbytes, step):
        e = min(nbytes, s + step)
        piece = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
        if p_n > 0:
            piece[torch.rand(e - s, generator=g, device=dev) < p_n] = 78
        d_bases[s:e] = piece
    torch.cuda.synchronize()              # (the engine reads the batch on its own stream)
    return d_bases, d_offs, nbytes


def pattern_ceilings(kmerdb_amd, device, k, per_kernel):
    """GB/s the device's memory system moves for each kernel's ACCESS PATTERN with no compute at all (csrc/kdb_probe.hip.h: residues streamed
    in, random 64-byte lines or 128-byte pieces out, whole pages in; two scratch regions of 4 GiB), measured now, on this box, and each kernel's
    own rate as a fraction of its pattern's ceiling.  `frac` in the roofline block stays a fraction of the 8 TB/s peak; this says how much of
    what is missing the memory system itself withholds from such a pattern."""
    import ctypes
    L = kmerdb_amd._abi.lib()
    n = L.kdb_hbm_pattern_count()
    out = (ctypes.c_double * n)()
    kmerdb_amd._abi.check(L.kdb_hbm_pattern_probe(int(device), out, n))
    gbs = {L.kdb_hbm_pattern_name(i).decode(): round(out[i], 1) for i in range(n)}
    two_level = k >= 14
    pattern_of = {"scatter_bases_kernel": "level1_128" if two_level else ("scatter_64" if k == 13 else "scatter_128"),
                  "scatter_ids_kernel": "level2_128", "page_hist_kernel": "pages_1k_read"}
    kernels = {}
    for name, pat in pattern_of.items():
        if name in per_kernel and per_kernel[name].get("gbs"):
            kernels[name] = {"pattern": pat, "ceiling_gbs": gbs[pat], "ceiling_frac_of_peak": round(gbs[pat] / HBM_PEAK_GBS, 4),
                             "kernel_gbs": per_kernel[name]["gbs"], "kernel_frac_of_ceiling": round(per_kernel[name]["gbs"] / gbs[pat], 4)}
    return {"gbs": gbs, "kernels": kernels,
            "what": "read + written GB/s of kernels that ONLY make the memory accesses (512 workgroups of 512 threads, no LDS, no ids): stream_read / "
                    "stream_write 1 KiB per wave instruction; pages_1k_read whole random 1 KiB pages, four in flight per wave (the histogram pass); "
                    "lines_64_write / pieces_128_write random 64-byte lines (what rounds 2-4 wrote) / 128-byte pieces (round 5), write-through; "
                    "scatter_* 1 KiB streamed in per 2 KiB of those out; level1_* 1 KiB in per 3 KiB out; level2_* two random 1.5 KiB pages in per "
                    "2 KiB out.  The k = 13 kernel (1024 rings of 64 elements) still writes 64-byte lines; level 1's high bytes leave 128 at a time"}


def kernel_bytes(k, tc, n_reads):
    """Bytes each kernel is asked to move per step (1 B/base in, whole 64-byte lines out into pages, whole
    pages back in, the count vector read + written where a bin is touched), from the engine's own counters `tc` (per step)."""
    two_level = tc["pages_ids"] > 0
    pb_bases = 1536 if (two_level and k <= 16) else 1024          # level-1 pages of k <= 16 carry 24-bit remainders as u16 + u8 arrays
    lb_bases = 96 if (two_level and k <= 16) else 64
    kb = {}
    if tc["pages_bases"] or tc["pages_ids"]:
        kb["scatter_bases_kernel"] = {"read": tc["bytes_in"], "write": tc["lines_bases"] * lb_bases + 4 * tc["pages_bases"]}
        if two_level:
            kb["scatter_ids_kernel"] = {"read": tc["pages_bases"] * (pb_bases + 8), "write": tc["lines_ids"] * 64 + 4 * tc["pages_ids"]}
            kb["page_hist_kernel"] = {"read": tc["pages_ids"] * (1024 + 8) + tc["table_bytes"] / 2, "write": tc["table_bytes"] / 2}
        else:
            kb["page_hist_kernel"] = {"read": tc["pages_bases"] * (1024 + 8) + tc["table_bytes"] / 2, "write": tc["table_bytes"] / 2}
    elif tc.get("table_bytes"):
        kb["count_kernel"] = {"read": tc["bytes_in"] + tc["table_bytes"] / 2, "write": tc["table_bytes"] / 2}
    kb["lens+mark_reads_kernel"] = {"read": 8.0 * (n_reads + 1), "write": 0.0}
    return kb


def per_kernel_table(prof, tc, k, steps, n_reads):
    """-> (per_kernel {name: ms, bytes, GB/s, fraction of the HBM peak}, step_ms {name: ms per step}, kern {name: avg, launches}, kbytes)."""
    kern = {name: {"avg_ms": ms / n, "launches": int(n)} for name, (ms, n) in prof.items() if n}
    step_ms = {n: v["avg_ms"] * v["launches"] / steps for n, v in kern.items()}
    kbytes = kernel_bytes(k, tc, n_reads)
    per_kernel = {}
    for name, ms in step_ms.items():
        bts = kbytes.get(name)
        ent = {"ms_per_step": round(ms, 4), "avg_ms": round(kern[name]["avg_ms"], 4), "launches_per_step": round(kern[name]["launches"] / steps, 4)}
        if bts and ms > 0:
            tot = bts["read"] + bts["write"]
            ent.update({"read_bytes_per_step": round(bts["read"]), "write_bytes_per_step": round(bts["write"]),
                        "gbs": round(tot / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(tot / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        per_kernel[name] = ent
    return per_kernel, step_ms, kern, kbytes


def flush_counters(eng):
    return (opt_or_none(eng, "hist_flushes") or 0, opt_or_none(eng, "flushed_batches") or 0, opt_or_none(eng, "full_flushes") or 0)


def job_tail(kmerdb_amd, eng, k, total, unique):
    """What `kmerdb profile` still has to do once a vector is counted (hot loop C of the reference, kmerdb/__init__.py:1939-1998, and
    parse.py:139-147), on the vector `eng` holds: the statistics (device), the copy-back, nullomer_array (compacted on the device) and the
    .kdb rows (format + row-aware deflate + 65536-byte BGZF members, all host threads the cgroup allows) into tmpfs.  Checked in the run:
    the first rows of the written file inflate (Python's gzip) to what Python formats from the copied-back counts."""
    import gzip
    import tempfile
    import numpy as np
    from kmerdb_amd import fileutil, util
    out = {"k": k, "rows": 4 ** k}
    avail_kb = next((int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")), 0)
    if avail_kb * 1024 < 5 * 8 * 4 ** k:                          # the vector, the nullomer ids and the file in tmpfs are host memory
        return {"skipped": "host memory: %.0f GB available" % (avail_kb / 1e6)}
    t = time.perf_counter()
    _, tot, uni = eng.finish(copy=False)
    out["stats_on_device_ms"] = round((time.perf_counter() - t) * 1e3, 2)
    assert (tot, uni) == (total, unique)
    t = time.perf_counter()
    counts, _, _ = eng.finish()
    out["copy_back_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    out["copy_back_gb_per_s"] = round(counts.nbytes / (time.perf_counter() - t) / 1e9, 1)
    t = time.perf_counter()
    nul = eng.nullomers(n=4 ** k - unique)
    out["nullomers_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    out["nullomers"] = int(nul.size)
    step = max(1, nul.size // 1000)
    assert nul.size == 4 ** k - unique and not counts[nul[::step].astype(np.int64)].any() and (nul.size < 2 or bool(np.all(np.diff(nul[::step].astype(np.int64)) > 0)))
    del nul
    md = {"version": fileutil.VERSION, "metadata_blocks": 1, "k": k, "total_kmers": int(total), "unique_kmers": int(unique),
          "unique_nullomers": int(4 ** k / 2 - unique), "sorted": False, "tags": [], "files": []}
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    threads = fileutil.default_writer_threads()
    with tempfile.TemporaryDirectory(dir=tmp) as d:
        sv = os.statvfs(d)
        if sv.f_bavail * sv.f_frsize < 3 * 4 ** k:               # (the file takes ~5-8 bytes per row)
            out["kdb_write"] = {"skipped": "not enough room in %s" % d}
            return out
        pk = os.path.join(d, "tail.%d.kdb" % k)
        t = time.perf_counter()
        nblocks = fileutil.write_kdb(pk, md, counts, nthreads=threads)
        dtw = time.perf_counter() - t
        with gzip.open(pk, "rb") as f:
            head = f.read(1 << 20)
        body = head.split(fileutil.header_delimiter.encode(), 1)[1]
        rows = body[:body.rfind(b"\n") + 1].decode().split("\n")[:-1]
        fr = counts[:len(rows)].astype(np.float64) / np.float64(total)
        assert rows == ["{0}\t{0}\t{1}\t{2}".format(i, int(counts[i]), fr[i]) for i in range(len(rows))], "the .kdb rows differ from Python's"
        # the whole file back through the native reader (KDBReader._slurp): every one of the 4^k rows must return its count
        back_ms = None
        if avail_kb * 1024 > 12 * 8 * 4 ** k:
            t = time.perf_counter()
            back = fileutil.read_kdb(pk, nthreads=threads)
            back_ms = round((time.perf_counter() - t) * 1e3, 1)
            assert np.array_equal(back.counts, counts), "the .kdb read back differs from the vector written"
            del back
        # what profile() does: the copy-back beside the row writer (kdb_copy_back_and_write_kdb_rows) -- against copy_back_ms + kdb_write.ms above
        del counts
        pk2 = os.path.join(d, "tail2.%d.kdb" % k)
        t = time.perf_counter()
        counts, nblocks2 = fileutil.write_kdb_from_engine(pk2, md, eng, nthreads=threads)
        out["copy_back_and_write_ms"] = round((time.perf_counter() - t) * 1e3, 1)
        with gzip.open(pk2, "rb") as f:                    # (the compressed bytes depend on which thread wrote which chunk: the text does not)
            head2 = f.read(1 << 20)
        assert nblocks2 == nblocks and head2 == head, "the overlapped write differs from the plain one"
        os.remove(pk2)
        out["kdb_write"] = {"ms": round(dtw * 1e3, 1), "rows_per_s": round(4 ** k / dtw), "threads": threads, "cpus_visible": len(os.sched_getaffinity(0)),
                            "read_back_ms": back_ms, "read_back_equals_the_vector": back_ms is not None,
                            "cpus_by_cgroup_quota": util._cgroup_cpu_limit(), "text_gb": round(nblocks * 65536 / 1e9, 2),
                            "text_gb_per_s": round(nblocks * 65536 / 1e9 / dtw, 2), "file_gb": round(os.path.getsize(pk) / 1e9, 3),
                            "rows_checked_against_python": len(rows)}
    out["total_ms"] = round(out["stats_on_device_ms"] + out["copy_back_ms"] + out["nullomers_ms"] + out["kdb_write"]["ms"], 1)
    out["total_overlapped_ms"] = round(out["stats_on_device_ms"] + out["nullomers_ms"] + out["copy_back_and_write_ms"], 1)
    return out


def config_region(kmerdb_amd, torch, local, label, k, canonical, L, batches, n_steps, min_len=0, eng=None, tail=False):
    """One more BASELINE configuration on this GPU: `n_steps` steps that rotate through the resident `batches`
    [(d_bases, d_offs, n_reads), ...] (distinct seeds), timed from the first submit to the end of the sync that adds the last
    pending batch to the vector.  Untimed before it: passes that let scratch and the page arena reach their size, then kdb_reset.
    Gate: Sum(counts) == every window of every timed read.  -> dict for the JSON line."""
    own = eng is None
    if own:
        eng = kmerdb_amd.Engine(k, canonicalize=canonical, device=local)
        if k >= 13:
            eng.set_option("arena_grow", 2)
            try:
                eng.set_option("arena_batches", min(64, max(8, min(n_steps, 63) + 1)))     # (one pass = one flush, with room to spare: a full arena asks for a larger one)
            except ValueError:
                pass
        if min_len:
            eng.set_option("min_len", min_len)

    def run(n):
        for i in range(n):
            b, o, nr = batches[i % len(batches)]
            eng.submit_device(b.data_ptr(), nr * L, o.data_ptr(), nr)
        eng.sync()

    t_setup = time.perf_counter()
    for _ in range(4):
        r0 = opt_or_none(eng, "arena_reallocs") or 0
        run(n_steps)
        if (opt_or_none(eng, "arena_reallocs") or 0) == r0:
            break
    eng.reset()
    setup_s = time.perf_counter() - t_setup
    eng.prof_enable(True)
    eng.prof_reset()
    tr0 = eng.traffic_counters()
    fl0 = flush_counters(eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n_steps)
    dt = time.perf_counter() - t0
    prof = eng.prof()
    eng.prof_enable(False)
    tr1 = eng.traffic_counters()
    fl1 = flush_counters(eng)
    _, total, unique = eng.finish(copy=False)
    reads = sum(batches[i % len(batches)][2] for i in range(n_steps))
    kmers_per_read = L - k + 1
    assert total == reads * kmers_per_read, (label, total, reads * kmers_per_read)
    tc = {n: (tr1[n] - tr0[n]) / n_steps for n in tr1}
    per_kernel, step_ms, _, _ = per_kernel_table(prof, tc, k, n_steps, reads / n_steps)
    flushes = fl1[0] - fl0[0]
    out = {"k": k, "canonical": canonical, "reads": reads, "read_len": L, "steps": n_steps, "distinct_batches": len(batches),
           "ms": round(dt * 1e3, 3), "ms_per_step": round(dt * 1e3 / n_steps, 4), "ms_per_10m_reads": round(dt * 1e3 / (reads / 1e7), 4),
           "gbase_per_s": round(reads * L / dt / 1e9, 2), "gkmers_per_s": round(reads * kmers_per_read / dt / 1e9, 2),
           "sum_gate": "Sum(counts) == reads x (L - k + 1) == %d" % total, "unique_kmers": unique,
           "hist_flushes": flushes if flushes else None,
           "batches_per_flush": round((fl1[1] - fl0[1]) / flushes, 2) if flushes else None,
           "flushes_forced_by_full_arena": (fl1[2] - fl0[2]) if flushes else None,
           "arena_pages": opt_or_none(eng, "arena_pages") if flushes else None,
           "device_ms_per_step": round(sum(step_ms.values()), 4),
           "per_kernel": per_kernel, "setup_s": round(setup_s, 2)}
    if tail:
        out["job_tail"] = job_tail(kmerdb_amd, eng, k, total, unique)
    if own:
        eng.close()
    return out


def baseline_configs(kmerdb_amd, torch, dev, local, L, seed0):
    """BASELINE.json's other single-GPU configurations at full size, one pass each over distinct reads:
    config 3 (k = 15, 100 M reads), config 5 (graph k = 12 = the forward 13-mer histogram, 50 M reads; graph.py:108-216), one rank's
    shard of config 4 (k = 17, 62.5 M reads into the 128 GiB vector, no reduce) and the k = 17 steady state (16 distinct batches)."""
    out = {}
    B = 10_000_000

    def make(nreads, seed):
        bs, left, i = [], nreads, 0
        while left > 0:
            n = min(B, left)
            bs.append(synthetic_batch(torch, dev, n, L, seed + 7919 * i) + (n,))
            left -= n
            i += 1
        torch.cuda.synchronize()
        return bs

    t = time.perf_counter()
    bs = make(100_000_000, seed0 + 3)
    out["config3_k15_100m_reads"] = config_region(kmerdb_amd, torch, local, "config3", 15, True, L, bs, len(bs), tail=True)
    out["config3_k15_100m_reads"]["wall_s"] = round(time.perf_counter() - t, 1)
    t = time.perf_counter()
    out["config5_graph_k12_50m_reads"] = config_region(kmerdb_amd, torch, local, "config5", 13, False, L, bs[:5], 5, min_len=12)
    out["config5_graph_k12_50m_reads"]["what"] = ("the weighted edge list of `kmerdb graph` at k = 12 = the forward 13-mer histogram over "
                                                  "records >= 12 long (kmerdb_amd/graph.py; reference graph.py:108-216)")
    out["config5_graph_k12_50m_reads"]["wall_s"] = round(time.perf_counter() - t, 1)
    # k = 17: one engine (one 128 GiB vector: hipMalloc alone takes ~6 s) serves the shard pass and the steady state
    t = time.perf_counter()
    bs = bs + make(60_000_000, seed0 + 4)                      # sixteen distinct batches in all
    try:
        eng = kmerdb_amd.Engine(17, canonicalize=True, device=local)
    except (MemoryError, RuntimeError) as e:
        out["config4_shard_k17_62m5_reads"] = {"skipped": "no room for the 128 GiB vector: %s" % e}
        return out
    eng.set_option("arena_grow", 2)
    eng.set_option("arena_batches", 64)
    shard = bs[:6] + [(bs[6][0], bs[6][1][:2_500_001], 2_500_000)]
    r = out["config4_shard_k17_62m5_reads"] = config_region(kmerdb_amd, torch, local, "config4-shard", 17, True, L, shard, len(shard), eng=eng)
    r["what"] = ("one rank's share of config 4 (500 M reads over 8 GPUs): 62.5 M reads, one histogram pass over the 128 GiB vector; "
                 "the RCCL reduce is not part of it (bench.py --gpus 8 --k 17 --reads 62500000 --steps 1)")
    r["wall_s"] = round(time.perf_counter() - t, 1)
    t = time.perf_counter()
    eng.reset()
    r = out["k17_steady_state"] = config_region(kmerdb_amd, torch, local, "k17-steady", 17, True, L, bs, 64, eng=eng)
    r["what"] = "64 steps rotating through 16 distinct 10 M-read batches; the arena holds as many batches as fit beside the vector (batches_per_flush)"
    r["wall_s"] = round(time.perf_counter() - t, 1)
    eng.close()
    return out


def resident_region(kmerdb_amd, d_bases, d_offs, n_reads, L, k, canonical, n_mode, local, steps, algo, opts):
    """ms per step of one more configuration of the resident-input region (own engine, own vector)."""
    with kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=n_mode, device=local, algo=algo) as e:
        for name, v in opts:
            e.set_option(name, v)
        for _ in range(2):
            e.submit_device(d_bases.data_ptr(), n_reads * L, d_offs.data_ptr(), n_reads)
        e.sync()
        t = time.perf_counter()
        for _ in range(steps):
            e.submit_device(d_bases.data_ptr(), n_reads * L, d_offs.data_ptr(), n_reads)
        e.sync()
        dt = time.perf_counter() - t
        _, total, _ = e.finish(copy=False)
        assert total == (steps + 2) * n_reads * (L - k + 1), total
    return dt / steps * 1e3


def extra_regions(kmerdb_amd, np, torch, d_bases, d_offs, n_reads, L, k, canonical, local, algo, opts):
    """SURVEY 8(d) timed regions (ii) and (iii), and the other modes of region (i); rank 0, N = 1 only.  Never `value`."""
    import tempfile
    from kmerdb_amd import parse, profile, synth
    out = {}
    nbytes = n_reads * L
    gbase = lambda ms: round(nbytes / ms / 1e6, 2)            # noqa: E731  Gbase/s from ms per batch
    # (i) other modes, inputs resident in HBM
    modes = {}
    for name, canon, n_mode in (("forward", False, kmerdb_amd.KDB_N_DROP), ("canonical_n_expand", True, kmerdb_amd.KDB_N_EXPAND)):
        ms = resident_region(kmerdb_amd, d_bases, d_offs, n_reads, L, k, canon, n_mode, local, 20, algo, opts)
        modes[name] = {"ms_per_step": round(ms, 4), "gbase_per_s": gbase(ms)}
    out["resident_other_modes"] = modes
    # (i') the real shape of a FASTQ: ragged lengths (uniform in 35..150) and 0.5 % N, canonical -- in the reference CLI's default N mode
    #      (expansion, kmerdb/__init__.py:1889 / kmer.py:545-565) and with --no-ambiguous (drop, kmer.py:541-544).  Record starts come
    #      from the offsets (lens_kernel's first_rec; nothing is written into the residues); every chunk near an N takes the front end's
    #      slow path.  (Each batch launches the equal-length and the ragged variant of its kernel; the one that does not apply returns at
    #      once -- the engine's per-kernel times cover both launches as one, rocprofv3 lists them apart: tools/pmc_table.py filters.)
    rb, ro, rbytes = ragged_batch(torch, torch.device("cuda", local), n_reads, 35, L, 0.005, synth.SEED0 + 77)
    rag = {"reads": n_reads, "bases": rbytes, "lengths": "uniform 35..%d" % L, "p_N": 0.005}
    for name, n_mode in (("n_expand", kmerdb_amd.KDB_N_EXPAND), ("n_drop", kmerdb_amd.KDB_N_DROP)):
        with kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=n_mode, device=local, algo=algo) as e:
            for oname, v in opts:
                e.set_option(oname, v)
            for _ in range(2):
                e.submit_device(rb.data_ptr(), rbytes, ro.data_ptr(), n_reads)
            e.sync()
            e.prof_enable(True)
            e.prof_reset()
            t = time.perf_counter()
            reps = 20
            for _ in range(reps):
                e.submit_device(rb.data_ptr(), rbytes, ro.data_ptr(), n_reads)
            e.sync()
            dt = (time.perf_counter() - t) / reps
            pk = {kn: round(ms / reps, 4) for kn, (ms, n) in e.prof().items() if n}
            _, total, _ = e.finish(copy=False)                 # (finish() checks Sum(counts) == k-mers emitted)
        rag[name] = {"ms_per_step": round(dt * 1e3, 4), "gbase_per_s": round(rbytes / dt / 1e9, 2), "kmers_per_step": total // (reps + 2),
                     "kernels_ms_per_step": pk}
    # the same lengths with a tenth of the N's (0.05 %: what a good Illumina run has): what N expansion costs when N-windows are rare
    rb2, ro2, rbytes2 = ragged_batch(torch, torch.device("cuda", local), n_reads, 35, L, 0.0005, synth.SEED0 + 78)
    with kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=kmerdb_amd.KDB_N_EXPAND, device=local, algo=algo) as e:
        for oname, v in opts:
            e.set_option(oname, v)
        for _ in range(2):
            e.submit_device(rb2.data_ptr(), rbytes2, ro2.data_ptr(), n_reads)
        e.sync()
        t = time.perf_counter()
        for _ in range(20):
            e.submit_device(rb2.data_ptr(), rbytes2, ro2.data_ptr(), n_reads)
        e.sync()
        dt = (time.perf_counter() - t) / 20
        e.finish(copy=False)
    rag["n_expand_p_N_0.0005"] = {"ms_per_step": round(dt * 1e3, 4), "gbase_per_s": round(rbytes2 / dt / 1e9, 2), "bases": rbytes2}
    del rb2, ro2
    uniform_ms_per_gbase = out["resident_other_modes"]["canonical_n_expand"]["ms_per_step"] / (nbytes / 1e9)
    rag["ms_per_gbase_over_uniform"] = {m: round(rag[m]["ms_per_step"] / (rag[m].get("bases", rbytes) / 1e9) / uniform_ms_per_gbase, 3)
                                        for m in ("n_expand", "n_drop", "n_expand_p_N_0.0005")}
    rag["what"] = ("config 2's read count with ragged lengths and N's, inputs resident in HBM; ms_per_gbase_over_uniform compares the time per base "
                   "with the uniform all-ACGT batch in N-expansion mode (fixed per-read costs weigh more on shorter reads); checked against the "
                   "oracle on a sample in cpu_baseline.ragged_sample")
    out["resident_ragged_n"] = rag
    out["_ragged"] = (rb, ro, rbytes)                  # (for the oracle check in the cpu_baseline leg; removed from the line)
    # (ii) H2D-fed: the same batch in pinned host memory, through kdb_submit_pinned's double-buffered pipeline
    m = min(n_reads, 10_000_000)
    pb = kmerdb_amd.pinned_empty(m * L)
    pb[:] = (d_bases[:m * L].cpu().numpy() & 0x7F)
    ho = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    with kmerdb_amd.Engine(k, canonicalize=canonical, device=local, algo=algo) as e:
        for name, v in opts:
            e.set_option(name, v)
        e.submit_pinned(pb, ho)
        e.sync()
        t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            e.submit_pinned(pb, ho)
        e.sync()
        dt = (time.perf_counter() - t) / reps
        _, total, _ = e.finish(copy=False)
        assert total == (reps + 1) * m * (L - k + 1)
    out["h2d_pinned"] = {"ms": round(dt * 1e3, 3), "gbase_per_s": round(m * L / dt / 1e9, 2), "reads": m,
                         "what": "batch in pinned host memory -> hipMemcpyAsync double buffering -> count (parsing excluded)"}
    # (iii) end to end from FASTQ files: read + split + md5/sha256 + H2D + count + vector copy-back (parse.parsefile),
    #       the same reads uncompressed, as one gzip stream (what the reference usually gets: parse.py:63-72) and as BGZF;
    #       then a 4-file samplesheet through profile() (vector summed on the device, one copy-back)
    import gzip
    from kmerdb_amd import fileutil, util
    mf = min(n_reads, 2_000_000)
    hb = (d_bases[:mf * L].cpu().numpy() & 0x7F).astype(np.uint8)
    hof = np.arange(mf + 1, dtype=np.uint64) * np.uint64(L)
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=tmp) as d:
        text = synth.fastq_text(hb, hof)
        paths = []
        for i in range(4):
            p = os.path.join(d, f"synthetic{i}.fq")
            with open(p, "wb") as f:
                f.write(text)
            paths.append(p)
        mz = min(mf, 500_000)                                      # (one gzip stream: a quarter of the reads -- zlib level 1 writes ~60 MB/s)
        tz = text if mz == mf else synth.fastq_text(hb[:mz * L], hof[:mz + 1])
        pgz, pbg = os.path.join(d, "synthetic.fq.gz"), os.path.join(d, "synthetic_bgzf.fq.gz")
        with gzip.open(pgz, "wb", compresslevel=1) as f:
            f.write(tz)
        # BGZF: all of the reads; the members are deflated by a few threads (zlib releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, util.effective_cpus()))) as pool, open(pbg, "wb") as f:
            mv = memoryview(text)
            for member in pool.map(lambda i: fileutil._bgzf_member(bytes(mv[i:i + 65280]), 1), range(0, len(text), 65280)):
                f.write(member)
            f.write(fileutil._bgzf_member(b""))
        del text, tz
        formats = {}
        for name, p, nr in (("fastq", paths[0], mf), ("fastq_gz", pgz, mz), ("fastq_bgzf", pbg, mf)):
            parse.parsefile(p, k, canonicalize=canonical, device=local)            # warm (pinned ring, page cache)
            tm = {}
            t = time.perf_counter()
            _, meta, _ = parse.parsefile(p, k, canonicalize=canonical, device=local, timings=tm)
            dt = time.perf_counter() - t
            assert meta["total_kmers"] == nr * (L - k + 1)
            stages = {n: round(v * 1e3, 1) for n, v in tm.items() if v is not None}
            walls = {n: v for n, v in stages.items() if not n.endswith("_thread_s")}
            formats[name] = {"ms": round(dt * 1e3, 1), "gbase_per_s": round(nr * L / dt / 1e9, 3), "reads": nr, "file_bytes": os.path.getsize(p),
                             "stages_ms": stages, "longest_stage": max(walls, key=walls.get) if walls else None,
                             # md5 of the raw file is one sequential stream (util.py:35-50 mandates it): no file goes faster than its digest
                             "md5_bound_gbase_per_s": round(nr * L / (tm["md5_thread_s"] or 1e-9) / 1e9, 3) if tm.get("md5_thread_s") else None}
        dt1 = formats["fastq"]["ms"] / 1e3
        sheet = os.path.join(d, "sheet.txt")
        open(sheet, "w").write("\n".join(paths) + "\n")
        t = time.perf_counter()
        counts4, md, _ = profile.profile([sheet], k, os.path.join(d, "out"), no_ambiguous=True, do_not_canonicalize=not canonical,
                                         device=local, write=False)
        dt4 = time.perf_counter() - t
        assert md["total_kmers"] == 4 * mf * (L - k + 1)
        # hot loop C of the reference (kmerdb/__init__.py:1980-1998): the .kdb rows -- 4^k lines "i \t id \t count \t frequency", cut into
        # 65536-byte BGZF members, deflate level 6 -- through the native writer; then the same samplesheet once more with the file written
        threads = fileutil.default_writer_threads()
        pk = os.path.join(d, "rows.%d.kdb" % k)
        t = time.perf_counter()
        nblocks = fileutil.write_kdb(pk, dict(md), counts4, nthreads=threads)
        dtw = time.perf_counter() - t
        kdb_bytes = os.path.getsize(pk)
        # ... and back through the native reader (KDBReader._slurp, fileutil.py:308-466): the whole vector must come back
        t = time.perf_counter()
        back = fileutil.read_kdb(pk, nthreads=threads)
        dtr = time.perf_counter() - t
        assert np.array_equal(back.counts, counts4) and np.array_equal(back.kmer_ids, np.arange(4 ** k, dtype=np.uint64)), "the .kdb read back differs from the vector written"
        del back
        t = time.perf_counter()
        fileutil.write_kdb(pk, dict(md), counts4, nthreads=threads, encoder="zlib") if k <= 13 else None
        dtwz = time.perf_counter() - t
        kdb_bytes_zlib = os.path.getsize(pk)
        t = time.perf_counter()
        nblocks1 = fileutil.write_kdb(pk, dict(md), counts4, nthreads=1) if k <= 12 else None
        dtw1 = time.perf_counter() - t
        t = time.perf_counter()
        _, md2, outp = profile.profile([sheet], k, os.path.join(d, "out"), no_ambiguous=True, do_not_canonicalize=not canonical,
                                       device=local, write=True)
        dt4w = time.perf_counter() - t
        assert md2["total_kmers"] == md["total_kmers"] and outp and os.path.getsize(outp) > 0
        del counts4
        # the whole `kmerdb profile -k 15` job on one file: counting is milliseconds, the 2^30 rows are the job
        pj = None
        avail_kb = next((int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")), 0)
        if k != 15 and avail_kb * 1024 > 6 * 8 * 4 ** 15:
            tmj = {}
            t = time.perf_counter()
            _, md15, out15 = profile.profile([paths[0]], 15, os.path.join(d, "job"), no_ambiguous=True, do_not_canonicalize=not canonical,
                                             device=local, write=True, timings=tmj)
            dtj = time.perf_counter() - t
            assert md15["total_kmers"] == mf * (L - 15 + 1)
            pj = {"k": 15, "reads": mf, "ms": round(dtj * 1e3, 1), "stages_ms": {n: round(v * 1e3, 1) for n, v in tmj.items()},
                  "kdb_file_gb": round(os.path.getsize(out15) / 1e9, 3), "rows": 4 ** 15, "writer_threads": fileutil.default_writer_threads(),
                  "what": "profile() of one FASTQ file at k = 15 with the .kdb written (tmpfs): read + split + md5/sha256 + H2D + count (count_s), statistics on the "
                          "device, then one copy-back of the 8 GiB vector in pieces beside the 2^30 rows being formatted, deflated and written (copy_back_and_write_kdb_s); "
                          "configs.config3_k15_100m_reads.job_tail has the same tail on config 3's own vector"}
            os.remove(out15)
    out["kdb_write"] = {"k": k, "rows": 4 ** k, "ms": round(dtw * 1e3, 1), "rows_per_s": round(4 ** k / dtw), "threads": threads,
                        "text_mb": round(nblocks * 65536 / 1e6, 1), "text_mb_per_s": round(nblocks * 65536 / 1e6 / dtw, 1),
                        "file_mb": round(kdb_bytes / 1e6, 1), "one_thread_ms": round(dtw1 * 1e3, 1) if nblocks1 else None,
                        "read_back_ms": round(dtr * 1e3, 1), "read_back_rows_per_s": round(4 ** k / dtr),
                        "zlib_level6_ms": round(dtwz * 1e3, 1) if k <= 13 else None, "zlib_level6_file_mb": round(kdb_bytes_zlib / 1e6, 1) if k <= 13 else None,
                        "what": "fileutil.write_kdb of the 4-file vector: header + kdb_write_kdb_rows (one pipeline: format, row-aware deflate, 65536-byte BGZF members, "
                                "parallel pwrite) into tmpfs; read_back_*: fileutil.read_kdb of that file through kdb_read_kdb_rows (members inflated and parsed in parallel), compared with the vector "
                                "in the run; zlib_level6_*: the same pipeline with zlib as the encoder (what Bio.bgzf does for the reference)"}
    if pj is not None:
        out["profile_k15"] = pj
    out["fastq_e2e"] = {"ms": round(dt1 * 1e3, 1), "gbase_per_s": round(mf * L / dt1 / 1e9, 3), "reads": mf,
                        "files4_ms": round(dt4 * 1e3, 1), "files4_gbase_per_s": round(4 * mf * L / dt4 / 1e9, 3),
                        "files4_with_kdb_written_ms": round(dt4w * 1e3, 1), "kdb_write_share_of_profile": round(max(dt4w - dt4, 0.0) / dt4w, 3),
                        "formats": formats,
                        "gz_over_plain": round((formats["fastq_gz"]["ms"] / formats["fastq_gz"]["reads"]) / (formats["fastq"]["ms"] / formats["fastq"]["reads"]), 2),
                        "what": "FASTQ in tmpfs -> parse.parsefile (read, split, md5+sha256, H2D, count, copy-back), uncompressed / one gzip stream / BGZF; "
                                "stages_ms: wall time of the consecutive stages of one file (read_split_submit = inflate + record splitting + "
                                "kdb_submit_pinned calls; the *_thread_s entries are the md5 / sha256 threads that run beside them); "
                                "files4 = a 4-file samplesheet through profile() without writing the .kdb, files4_with_kdb_written = the same with write=True (the whole `kmerdb profile` job)"}
    return out


def opt_or_none(eng, name):
    """An engine option that an older build of the library may not have (A/B runs with KDB_LIB)."""
    try:
        return eng.get_option(name)
    except ValueError:
        return None


def committed_counters(k, n_reads, L, canonical, algo):
    """HBM bytes and LDS counters of one step from the committed rocprofv3 PMC passes (profiles/), only when they were
    taken on this exact workload."""
    traffic, lds = None, None
    for name in ("traffic_k%d.json" % k,):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            tj = json.load(open(p))
            if (tj.get("k"), tj.get("reads"), tj.get("read_len"), tj.get("canonical")) == (k, n_reads, L, canonical) \
                    and tj.get("algo") == ("direct" if algo == 1 else "lds"):
                traffic = tj
    p = os.path.join(ROOT, "profiles", "lds_k%d.json" % k)
    if os.path.exists(p):
        lj = json.load(open(p))
        if (lj.get("k"), lj.get("reads"), lj.get("read_len"), lj.get("canonical")) == (k, n_reads, L, canonical):
            lds = lj
    return traffic, lds


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # stdout is reserved for the one JSON line, and libraries write there (RCCL prints a version banner when its first communicator
    # comes up, gloo's C++ side announces its peers): fd 1 goes to stderr for the whole run, the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import kmerdb_amd
    from kmerdb_amd import distributed, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("KDB_BENCH_ALL_ON_DEVICE0") == "1":     # rehearsal of the N>1 control flow on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("KDB_BENCH_FORCE_DIST") == "1":     # (FORCE_DIST: a one-rank process group, to exercise the RCCL calls on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        dist.barrier()

    k, n_reads, L = args.k, args.reads, args.read_len
    canonical = not args.forward
    n_mode = kmerdb_amd.KDB_N_EXPAND if args.expand else kmerdb_amd.KDB_N_DROP
    kmers_per_read = L - k + 1
    nbytes = n_reads * L
    opts = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt]

    # ---- synthetic batch, generated on the device (uniform ACGT, no N), resident in HBM ----------
    D = args.distinct_batches if args.distinct_batches > 0 else (1 if k <= 12 else (8 if k <= 16 else 16))
    batches = [synthetic_batch(torch, dev, n_reads, L, synth.SEED0 + 2 + 1000 * rank + 7919 * i) for i in range(D)]
    d_bases, d_offs = batches[0]
    table = torch.zeros(4 ** k, dtype=torch.int64, device=dev)     # the engine adopts this vector (RCCL reduces it)
    torch.cuda.synchronize()

    eng = kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=n_mode, device=local, table_ptr=table.data_ptr(), algo=args.algo)
    arena_grow = None
    if k >= 13 and not any(name == "arena_grow" for name, _ in opts):
        # steady state of a long-lived engine: the page arena doubles whenever it has filled up (the default lets it grow
        # only once the saved sweeps outweigh the allocation: hundreds of batches at k = 17, in effect never at k <= 15)
        try:
            eng.set_option("arena_grow", 2)
            arena_grow = 2
            eng.set_option("arena_batches", min(64, args.steps + 1))        # its first size: what the job will want (one allocation, not a doubling series)
        except ValueError:
            pass                                                   # (an older build of the library: A/B runs with KDB_LIB)
    for name, v in opts:
        eng.set_option(name, v)
        if name == "arena_grow":
            arena_grow = v

    # N > 1: RCCL's first contact comes FIRST.  The communicator exists since init_process_group, but RCCL
    # creates its channel / peer-to-peer buffers at the first collective of each kind -- and below, the page arena of k >= 14 grows to
    # "85 % of what is free" beside a 128 GiB vector.  So: every reduce shape runs once on a scratch tensor now (timed: the probe picks the
    # fastest), and the engine is told to leave the reduce's scratch and RCCL's head-room alone ("reserve_bytes").
    reduce_shape, reduce_probe = args.reduce_shape, None
    if dist is not None:
        cdev = dev if args.backend == "nccl" else None
        probe_bytes = int(min(4 ** k * 8, 1 << 30))
        chosen, probe_ms = distributed.probe_reduce_shapes(cdev, None, nbytes=probe_bytes)
        torch.cuda.synchronize()
        if reduce_shape == "auto":
            reduce_shape = chosen
        reduce_probe = {"bytes": probe_bytes, "ms": probe_ms, "fastest": chosen, "used": reduce_shape,
                        "gbs": {n: (round(probe_bytes / (v * 1e-3) / 1e9, 1) if v else None) for n, v in probe_ms.items()},
                        "errors_on_rank0": dict(distributed.last_probe_errors) or None, "when": "before the first batch (before the arena sizes itself)"}
        try:
            eng.set_option("reserve_bytes", distributed.reduce_reserve_bytes(world))
        except ValueError:
            pass                                                   # (an older build of the library: A/B runs with KDB_LIB)

    step_no = [0]

    def one_step():
        b, o = batches[step_no[0] % D]
        step_no[0] += 1
        eng.submit_device(b.data_ptr(), nbytes, o.data_ptr(), n_reads)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    # k >= 13 keeps scattered batches pending in a page arena until the sync (deferred histogram pass); the arena grows
    # with hipMalloc as the job goes on.  An untimed cycle lets it reach its size, so that the timed region measures
    # counting, not allocation (the default k = 12 run is unaffected).
    # (the arena doubles when it has filled up: cycles of `steps` batches until one passes without a reallocation -- hipMalloc
    #  of a 100-GiB arena takes seconds, profiles/r03/malloc_time.txt)
    pool_warmup = 0
    if k >= 13 and args.steps > args.warmup:
        for _ in range(6):
            r0 = opt_or_none(eng, "arena_reallocs") or 0
            eng.sync()
            for _ in range(min(args.steps, 64)):
                one_step()
            eng.sync()
            pool_warmup += min(args.steps, 64)
            if (opt_or_none(eng, "arena_reallocs") or 0) == r0:
                break
    barrier()
    # per-rank gate on the warm-up steps: Sum(counts) == every window of every read
    _, total, _ = eng.finish(copy=False)
    assert total == (args.warmup + pool_warmup) * n_reads * kmers_per_read, (total, (args.warmup + pool_warmup) * n_reads * kmers_per_read)
    barrier()
    eng.prof_enable(True)
    eng.prof_reset()
    traffic0 = eng.traffic_counters()
    arena_reallocs0 = opt_or_none(eng, "arena_reallocs") or 0
    flush0 = (opt_or_none(eng, "hist_flushes") or 0, opt_or_none(eng, "flushed_batches") or 0, opt_or_none(eng, "full_flushes") or 0)

    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    eng.sync()
    t_count = time.perf_counter() - t0
    reduce_ms, reduce_calls = 0.0, 0
    free_before_reduce = opt_or_none(eng, "free_hbm") if dist is not None else None
    if dist is not None:
        tr = time.perf_counter()
        rt = table
        if args.backend != "nccl":     # gloo rehearsal on one GPU: the sharded shapes run on a host copy (small vectors only), else the plain reduce
            if reduce_shape != "ring" and 4 ** k * 8 <= (1 << 30):
                rt = table.cpu()
            else:
                reduce_shape = "ring"
        try:
            reduce_calls = distributed.reduce_vector(rt, dst=0, shape=reduce_shape)   # one reduce of the 4^k vector, in <= 1 GiB chunks
        except Exception as e:  # noqa: BLE001 - no silent fall-back to another shape: the line would not say what was measured
            sys.stderr.write("[bench] rank %d: reduce shape %r failed on the real vector: %s: %s\n" % (rank, reduce_shape, type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)
        if rt is not table and rank == 0:
            table.copy_(rt)
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - tr) * 1e3
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([t_count * 1e3, reduce_ms, elapsed_local * 1e3, float(dist.get_world_size()), float(free_before_reduce or 0),
                             float(opt_or_none(eng, "reserve_bytes") or 0), float(opt_or_none(eng, "arena_budget_bytes") or 0),
                             float(opt_or_none(eng, "free_at_sizing") or 0)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "count_ms": round(float(x[0]), 3), "reduce_ms": round(float(x[1]), 3),
                     "elapsed_ms": round(float(x[2]), 3), "world_size_seen": int(x[3]), "free_hbm_before_reduce": int(x[4]),
                     "reserve_bytes": int(x[5]), "arena_budget_bytes": int(x[6]), "free_hbm_when_the_arena_was_sized": int(x[7])} for r, x in enumerate(allr)]

    # ---- correctness gate: Sum(counts) == every window of every step, on every rank ---------------
    prof = eng.prof()
    eng.prof_enable(False)
    traffic1 = eng.traffic_counters()
    arena = (opt_or_none(eng, "arena_pages"), (opt_or_none(eng, "arena_reallocs") or 0) - arena_reallocs0)
    flush1 = (opt_or_none(eng, "hist_flushes") or 0, opt_or_none(eng, "flushed_batches") or 0, opt_or_none(eng, "full_flushes") or 0)
    n_flushes = flush1[0] - flush0[0]
    total_steps = args.steps + args.warmup + pool_warmup
    expect = total_steps * n_reads * kmers_per_read
    if dist is not None:
        if rank == 0:      # rank 0's vector now holds the sum over ranks (read with table_stats: finish() would rightly refuse it)
            _, got, _ = eng.table_stats(copy=False)
            assert got == expect * world, (got, expect * world)
    else:
        _, total, _ = eng.finish(copy=False)
        assert total == expect, (total, expect)

    if rank != 0:
        eng.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline (HIP events on the engine's compute stream + the engine's own byte counters, both of THIS run) ----
    tc = {n: (traffic1[n] - traffic0[n]) / args.steps for n in traffic1}        # per step, over the timed region
    per_kernel, step_ms, kern, kbytes = per_kernel_table(prof, tc, k, args.steps, n_reads)
    per_step_ms = sum(step_ms.values())
    dominant = max(step_ms, key=step_ms.get)
    step_bytes = sum(v["read"] + v["write"] for v in kbytes.values())
    compulsory = nbytes + 8 * (n_reads + 1) + 2 * 8 * min(4 ** k, n_reads * kmers_per_read)     # input once + every touched counter read and written once
    traffic, lds = committed_counters(k, n_reads, L, canonical, args.algo)
    if traffic and traffic.get("distinct_batches", 1) != D:
        traffic = None          # (a PMC pass of another batch rotation: the histogram pass touches other bins)
    if traffic and k >= 13 and n_flushes and abs(traffic.get("batches_per_flush", 0) - (flush1[1] - flush0[1]) / n_flushes) > 0.5:
        traffic = None          # (the deferred histogram pass runs once per flush: per-launch PMC bytes only compare at equal batches per flush)
    dom = per_kernel[dominant]
    dom_launch_bytes = (dom.get("read_bytes_per_step", 0) + dom.get("write_bytes_per_step", 0)) / max(dom["launches_per_step"], 1e-9)
    pmc_dom = None
    if traffic:
        pk = traffic.get("per_kernel_per_launch", {}).get(dominant.split("<")[0])
        if pk:
            pmc_dom = pk["read_bytes"] + pk["write_bytes"]
    alg_bytes_step = n_reads * (L + 16 * kmers_per_read)           # 1 B/base + 16 B/k-mer
    # what limits the dominant kernel: the scatter kernels issue VALU instructions most of the time (SQ_INSTS_VALU x 4 cycles / SIMDs /
    # clock = 0.65-0.73 of their duration, profiles/) and keep the LDS 55 % busy; the histogram pass is the HBM-bound one.  `frac`
    # is the fraction of the HBM peak either way (the unit the contract asks for); `bound` names the limiter.
    bound = "hbm" if dominant.startswith("page_hist") or dominant.startswith("stats") else "valu-issue"
    roofline = {"bound": bound, "frac_is": "fraction of the HBM peak (8 TB/s) the dominant kernel moves", "kernel": dominant,
                "achieved": dom.get("gbs"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom.get("hbm_frac"),
                "traffic": pmc_dom,
                "bytes_per_launch": round(dom_launch_bytes), "kernel_avg_ms": dom["avg_ms"],
                "accounting": "bytes the dominant kernel moves per launch by the engine's own counters of this run (residue bytes in; whole "
                              "128-byte pieces -- k = 13: 64-byte lines -- and page tags out) / its average duration by HIP events on its stream; `traffic` = rocprofv3 PMC "
                              "FETCH_SIZE+WRITE_SIZE bytes per launch of the same command (profiles/), null if no pass matches this workload",
                "limiter": "since the pages are written in 128-byte pieces (round 5) the k <= 12 scatter kernel and level 1 are bound on the CU (VALU issue 72 %, LDS 54 % "
                           "busy, two barriers per tile); level 2 and the histogram pass sit at what the memory system gives their access patterns; with 64-byte "
                           "lines all three scatter kernels sat at the ceiling for random 64-byte writes (pattern_ceilings; DESIGN.md section 4)",
                "per_kernel": per_kernel,
                "step": {"device_ms": round(per_step_ms, 4), "bytes": round(step_bytes), "gbs": round(step_bytes / (per_step_ms * 1e-3) / 1e9, 1),
                         "hbm_frac": round(step_bytes / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "compulsory_bytes": compulsory,
                         "compulsory_frac": round(compulsory / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "pmc_bytes": traffic["hbm_bytes_per_step"] if traffic else None,
                         "engine_over_pmc": round(step_bytes / traffic["hbm_bytes_per_step"], 3) if traffic else None},
                "engine_counters_per_step": {n: round(v, 1) for n, v in tc.items()},
                "arena": {"pages": arena[0], "reallocs_in_timed_region": arena[1], "arena_grow": arena_grow, "hist_flushes": n_flushes,
                          "batches_per_flush": round((flush1[1] - flush0[1]) / n_flushes, 2) if n_flushes else None,
                          "flushes_forced_by_full_arena": flush1[2] - flush0[2]} if k >= 13 else None,
                "distinct_batches": D,
                "kernels_avg_ms": {n: round(v["avg_ms"], 4) for n, v in kern.items()},
                "kernels_ms_per_step": {n: round(v, 4) for n, v in step_ms.items()},
                # SURVEY 8(d)'s formula, kept for continuity: it prices a 16-byte RMW per k-mer that this design does not perform
                "virtual_8d": {"algorithmic_bytes_per_step": alg_bytes_step, "gbs": round(alg_bytes_step / (per_step_ms * 1e-3) / 1e9, 1),
                               "frac_of_peak": round(alg_bytes_step / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "a virtual bandwidth (> 1 is possible); not a roofline"}}
    # ---- what the memory system of THIS box delivers for the kernels' access patterns with no compute (kdb_hbm_pattern_probe) ----
    if world == 1 and not args.no_extra_regions:
        try:
            roofline["pattern_ceilings"] = pattern_ceilings(kmerdb_amd, local, k, per_kernel)
        except (MemoryError, RuntimeError, ValueError) as e:      # (8 GiB of scratch beside a 128 GiB vector: a diagnostic must not end the run)
            roofline["pattern_ceilings"] = {"error": str(e)}
    lds_block = None
    if lds:
        per = lds.get("per_kernel_per_launch", {})
        dk = dominant.split("<")[0]
        if dk in per:
            v = per[dk]
            lds_block = {"kernel": dk, "lds_busy_frac": v.get("lds_busy_frac"), "lds_bank_conflict_share": v.get("lds_bank_conflict_share"),
                         "valu_busy_frac": v.get("valu_busy_frac"), "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"),
                         "source": lds.get("source") or ("profiles/lds_k%d.json (tools/profile_gpu.sh; summary: profiles/r04/rocprof_k%d_summary.md)" % (k, k)), "note": "from a committed rocprofv3 PMC pass of this workload (profiles/), not measured in this run"}

    regions, ragged = None, None
    if world == 1 and not args.no_extra_regions and k <= 13:
        regions = {"resident": {"ms": round(elapsed / args.steps * 1e3, 4), "gbase_per_s": round(args.steps * nbytes / elapsed / 1e9, 3)}}
        regions.update(extra_regions(kmerdb_amd, np, torch, d_bases, d_offs, n_reads, L, k, canonical, local, args.algo, opts))
        ragged = regions.pop("_ragged", None)

    # ---- CPU baseline: the oracle (a port of the reference's per-window loop) on a bounded sample ---
    cpu = None
    if world == 1 and not args.no_cpu_baseline and k <= 15:      # (the oracle's dense host vector is 4^k * 8 bytes)
        from oracle import kmer_oracle
        kmer_oracle.build()
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        from kmerdb_amd import util
        quota = util._cgroup_cpu_limit()
        cores = max(1, min(avail, int(quota + 0.999) if quota else 16))     # the CPUs this process really has: its cgroup's quota (16 on the GPU pool's boxes: one GPU's part of a host shared 8 ways)
        o_mode = kmer_oracle.N_EXPAND if args.expand else kmer_oracle.N_DROP
        # probe, then size the sample for ~12 s on `cores` threads and ~6 s on one thread
        mp = min(20_000, n_reads)
        hb = (d_bases[:mp * L].cpu().numpy() & 0x7F).astype(np.uint8)
        ho = (np.arange(mp + 1, dtype=np.uint64) * np.uint64(L))
        tc = time.perf_counter()
        kmer_oracle.c_count(hb, ho, k, canonical, o_mode)
        rate1 = mp * kmers_per_read / (time.perf_counter() - tc)
        m1 = int(min(n_reads, max(mp, 6.0 * rate1 / kmers_per_read)))
        m = int(min(n_reads, args.cpu_sample_reads if args.cpu_sample_reads > 0 else max(m1, 12.0 * rate1 * cores * 0.5 / kmers_per_read)))
        hb = (d_bases[:m * L].cpu().numpy() & 0x7F).astype(np.uint8)
        ho = (np.arange(m + 1, dtype=np.uint64) * np.uint64(L))
        tc = time.perf_counter()
        want, want_total = kmer_oracle.c_count(hb, ho, k, canonical, o_mode, nthreads=cores)
        t_all = time.perf_counter() - tc
        m1 = min(m1, m)
        tc = time.perf_counter()
        kmer_oracle.c_count(hb[:m1 * L], ho[:m1 + 1], k, canonical, o_mode)
        t_one = time.perf_counter() - tc
        # SURVEY 8(d) also asks for "all host cores": one thread per CPU the process may be scheduled on (256 here) on the same sample --
        # under a CPU quota those threads share the quota's CPUs, so this is the same machine share, oversubscribed
        t_vis = None
        if avail > cores:
            tc = time.perf_counter()
            kmer_oracle.c_count(hb, ho, k, canonical, o_mode, nthreads=min(avail, 256))
            t_vis = time.perf_counter() - tc
        # parity of the sample, through the same device-resident path
        chk = kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=n_mode, device=local, algo=args.algo)
        for name, v in opts:
            chk.set_option(name, v)
        chk.submit_device(d_bases.data_ptr(), m * L, d_offs.data_ptr(), m)
        got, got_total, _ = chk.finish()
        chk.close()
        assert got_total == want_total and np.array_equal(got, want), "GPU counts differ from the oracle on the sample"
        del got, want
        if True:
            cpu = {"value": round(want_total / t_all, 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
                   "sample": f"first {m} reads of the same batch ({want_total} k-mers, {t_all:.1f} s on {cores} threads = the CPU quota of this process's cgroup "
                             f"on a host with {avail} CPUs visible; {m1} reads, {t_one:.1f} s on 1 thread); GPU counts on the sample equal the oracle's bit-for-bit",
                   "single_thread_value": round(m1 * kmers_per_read / t_one, 1) if t_one > 0 else None, "host_cpus_visible": avail,
                   "cgroup_cpu_quota": quota,
                   "all_cores_value": round(want_total / t_vis, 1) if t_vis else round(want_total / t_all, 1),
                   "all_cores_threads": min(avail, 256) if t_vis else cores,
                   "reference_python_1core": "0.13-0.21 M k-mers/s (BASELINE.md section 2, survey container)"}
        if ragged is not None:
            # the ragged, N-bearing batch of timed_regions.resident_ragged_n: the oracle on its first reads (N-expansion mode, the
            # reference CLI's default), timed, and the engine's counts on the same reads compared bit for bit
            rb, ro, _ = ragged
            mr = int(min(n_reads, 200_000))
            ho = ro[:mr + 1].cpu().numpy().astype(np.uint64)
            hb = (rb[:int(ho[-1])].cpu().numpy() & 0x7F).astype(np.uint8)
            tc = time.perf_counter()
            want, want_total = kmer_oracle.c_count(hb, ho, k, canonical, kmer_oracle.N_EXPAND, nthreads=cores)
            t_r = time.perf_counter() - tc
            with kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=kmerdb_amd.KDB_N_EXPAND, device=local, algo=args.algo) as chk:
                for name, v in opts:
                    chk.set_option(name, v)
                chk.submit_device(rb.data_ptr(), int(ho[-1]), ro.data_ptr(), mr)
                got, got_total, _ = chk.finish()
            assert got_total == want_total and np.array_equal(got, want), "GPU counts differ from the oracle on the ragged N-bearing sample"
            cpu["ragged_sample"] = {"value": round(want_total / t_r, 1), "unit": "k-mers/s", "cores": cores, "reads": mr, "bases": int(ho[-1]),
                                    "kmers_incl_n_expansions": int(want_total),
                                    "what": "first reads of the ragged batch (lengths 35..%d, 0.5 %% N), N-expansion mode; GPU counts equal the oracle's bit for bit" % L}
            del got, want
    del ragged

    eng.close()

    kmers_total = world * args.steps * n_reads * kmers_per_read
    out = {
        "metric": "k-mers counted/sec at k=12, synthetic 150 bp FASTQ" if k == 12 else f"k-mers counted/sec at k={k}",
        "value": round(kmers_total / elapsed, 1),
        "unit": "k-mers/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"k={k} profile, {n_reads} synthetic {L} bp reads per GPU per step, dense 4^{k} uint64 histogram "
                               f"({'canonical' if canonical else 'forward'}{', N expansion' if args.expand else ''}), inputs resident in HBM",
                   "k": k, "reads_per_gpu_per_step": n_reads, "read_len": L, "canonical": canonical, "n_expand": bool(args.expand),
                   "algo": {0: "auto", 1: "direct-atomics", 2: "lds-histogram"}[args.algo],
                   "sharding": f"reads x{world}, one chunked RCCL reduce at the end" if world > 1 else "single GPU"},
        "gbase_per_s": round(world * args.steps * nbytes / elapsed / 1e9, 3),
        "timed_region_s": round(elapsed, 3),
        "count_only_ms_per_step": round(t_count / args.steps * 1e3, 4),
        "reduce_ms": round(reduce_ms, 3), "reduce_calls": reduce_calls, "reduce_shape": reduce_shape if dist is not None else None,
        "reduce_vector_bytes": 4 ** k * 8 if dist is not None else None,
        "reduce_gbs": round(4 ** k * 8 / (reduce_ms * 1e-3) / 1e9, 1) if dist is not None and reduce_ms > 0 else None,
        "reduce_share_of_timed_region": round(reduce_ms * 1e-3 / elapsed, 4) if dist is not None else None,
        "reduce_probe": reduce_probe,
        "per_rank": per_rank,
        "roofline": roofline,
        "pmc_lds": lds_block,
        "timed_regions": regions,
        "cpu_baseline": cpu,
    }
    if k <= 13:     # sha256 of the little-endian uint64 vector of the whole job (after the reduce for N > 1)
        import hashlib
        out["vector_sha256"] = hashlib.sha256(table.cpu().numpy().tobytes()).hexdigest()
    if world == 1 and not args.no_configs and k == 12 and args.algo == 0 and not args.opt and canonical and not args.expand and n_reads == 10_000_000 and L == 150:
        # the default run also carries BASELINE.json's other single-GPU configurations (the headline's buffers are released first)
        del table, batches, d_bases, d_offs
        torch.cuda.empty_cache()
        t_cfg = time.perf_counter()
        out["configs"] = baseline_configs(kmerdb_amd, torch, dev, local, L, synth.SEED0)
        out["configs"]["wall_s"] = round(time.perf_counter() - t_cfg, 1)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
