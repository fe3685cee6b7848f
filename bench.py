#!/usr/bin/env python3
"""bench.py -- k-mers counted per second at k=12 on synthetic 150-bp reads (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one pass of the hot path (record marks + 2-bit encode + 4^k histogram) over one batch of
10 M synthetic 150-bp reads (BASELINE config 2) that is already resident in HBM.  Each rank (one per
GPU) counts its own batch into its own 4^12 uint64 vector (weak scaling, no data-path collective);
for N > 1 the job ends with ONE RCCL reduce of the vector to rank 0 over xGMI, inside the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=12)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct atomics, 2 LDS-histogram")
    ap.add_argument("--forward", action="store_true", help="do not canonicalize")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = size the sample for ~12 s of CPU work")
    args = ap.parse_args()

    import numpy as np
    import torch
    import kmerdb_amd
    from kmerdb_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("KDB_BENCH_ALL_ON_DEVICE0") == "1":     # rehearsal of the N>1 control flow on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    k, n_reads, L = args.k, args.reads, args.read_len
    canonical = not args.forward
    kmers_per_read = L - k + 1
    nbytes = n_reads * L

    # ---- synthetic batch, generated on the device (uniform ACGT, no N), resident in HBM ----------
    g = torch.Generator(device=dev)
    g.manual_seed(synth.SEED0 + 2 + 1000 * rank)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    d_bases = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, nbytes, step):
        e = min(nbytes, s + step)
        d_bases[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
    d_offs = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * L
    table = torch.zeros(4 ** k, dtype=torch.int64, device=dev)     # the engine adopts this vector (RCCL reduces it)
    torch.cuda.synchronize()

    eng = kmerdb_amd.Engine(k, canonicalize=canonical, n_mode=kmerdb_amd.KDB_N_DROP, device=local,
                            table_ptr=table.data_ptr(), algo=args.algo)

    for kv in args.opt:
        name, v = kv.split("=")
        eng.set_option(name, int(v))

    def one_step():
        eng.submit_device(d_bases.data_ptr(), nbytes, d_offs.data_ptr(), n_reads)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    # k >= 14 keeps partitioned batches pending until the sync (deferred histogram pass); their buffers come from a
    # pool that grows with hipMalloc the first time.  One untimed cycle of `steps` submits sizes the pool, so that the
    # timed region below measures counting, not first-time allocation (the default k = 12 run is unaffected).
    pool_warmup = args.steps if k >= 14 and args.steps > args.warmup else 0
    if pool_warmup:
        eng.sync()
        for _ in range(pool_warmup):
            one_step()
    if dist is not None:      # untimed: bring up the RCCL communicator and its xGMI rings before the timed reduce
        scratch = torch.zeros(4 ** k, dtype=torch.int64, device=dev)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        del scratch
    barrier()
    # per-rank gate on the warm-up steps: Sum(counts) == every window of every read
    _, total, _ = eng.finish(copy=False)
    assert total == (args.warmup + pool_warmup) * n_reads * kmers_per_read, (total, (args.warmup + pool_warmup) * n_reads * kmers_per_read)
    barrier()
    eng.prof_enable(True)
    eng.prof_reset()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    eng.sync()
    t_count = time.perf_counter() - t0
    reduce_ms = 0.0
    if dist is not None:
        tr = time.perf_counter()
        dist.reduce(table, dst=0, op=dist.ReduceOp.SUM)            # one RCCL reduce of the 4^k vector
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - tr) * 1e3
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness gate: Sum(counts) == every window of every step, on every rank ---------------
    prof = eng.prof()
    eng.prof_enable(False)
    total_steps = args.steps + args.warmup + pool_warmup
    expect = total_steps * n_reads * kmers_per_read
    if world > 1:
        if rank == 0:      # rank 0's vector now holds the sum over ranks
            got = int(table.sum().item())
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

    # ---- roofline of the step (HIP events on the engine's compute stream) -------------------------
    kern = {name: {"avg_ms": ms / n, "launches": int(n)} for name, (ms, n) in prof.items() if n}
    per_step_ms = sum(v["avg_ms"] * v["launches"] for v in kern.values()) / args.steps
    dominant = max(kern, key=lambda n: kern[n]["avg_ms"] * kern[n]["launches"])
    alg_bytes_step = n_reads * (L + 16 * kmers_per_read)           # 1 B/base + 16 B/k-mer
    achieved = alg_bytes_step / (per_step_ms * 1e-3) / 1e9
    # HBM bytes per step from the committed rocprofv3 PMC passes (FETCH_SIZE/WRITE_SIZE, gfx950 corrections applied;
    # tools/profile_gpu.sh) -- only quoted when the profile was taken on this exact workload
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_k12.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if (tj.get("k"), tj.get("reads"), tj.get("read_len"), tj.get("canonical")) == (k, n_reads, L, canonical) \
                and tj.get("algo") == ("direct" if args.algo == 1 else "lds"):
            traffic = tj["hbm_bytes_per_step"]
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": dominant, "kernel_avg_ms": round(kern[dominant]["avg_ms"], 4),
                "step_device_ms": round(per_step_ms, 4),
                "algorithmic_bytes_per_step": alg_bytes_step,
                "kernels_avg_ms": {n: round(v["avg_ms"], 4) for n, v in kern.items()},
                "kernels_ms_per_step": {n: round(v["avg_ms"] * v["launches"] / args.steps, 4) for n, v in kern.items()},
                "launches_per_step": {n: v["launches"] // args.steps for n, v in kern.items()}}

    # ---- CPU baseline: the oracle (a port of the reference's per-window loop) on a bounded sample ---
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import kmer_oracle
        kmer_oracle.build()
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 16))          # one GPU's share of the host (the box is shared 8 ways)
        # probe, then size the sample for ~12 s on `cores` threads and ~6 s on one thread
        mp = min(20_000, n_reads)
        hb = (d_bases[:mp * L].cpu().numpy() & 0x7F).astype(np.uint8)
        ho = (np.arange(mp + 1, dtype=np.uint64) * np.uint64(L))
        tc = time.perf_counter()
        kmer_oracle.c_count(hb, ho, k, canonical, kmer_oracle.N_DROP)
        rate1 = mp * kmers_per_read / (time.perf_counter() - tc)
        m1 = int(min(n_reads, max(mp, 6.0 * rate1 / kmers_per_read)))
        m = int(min(n_reads, args.cpu_sample_reads if args.cpu_sample_reads > 0 else max(m1, 12.0 * rate1 * cores * 0.5 / kmers_per_read)))
        hb = (d_bases[:m * L].cpu().numpy() & 0x7F).astype(np.uint8)
        ho = (np.arange(m + 1, dtype=np.uint64) * np.uint64(L))
        tc = time.perf_counter()
        want, want_total = kmer_oracle.c_count(hb, ho, k, canonical, kmer_oracle.N_DROP, nthreads=cores)
        t_all = time.perf_counter() - tc
        m1 = min(m1, m)
        tc = time.perf_counter()
        kmer_oracle.c_count(hb[:m1 * L], ho[:m1 + 1], k, canonical, kmer_oracle.N_DROP)
        t_one = time.perf_counter() - tc
        # parity of the sample, through the same device-resident path
        chk = kmerdb_amd.Engine(k, canonicalize=canonical, device=local, algo=args.algo)
        chk.submit_device(d_bases.data_ptr(), m * L, d_offs.data_ptr(), m)
        got, got_total, _ = chk.finish()
        chk.close()
        assert got_total == want_total and np.array_equal(got, want), "GPU counts differ from the oracle on the sample"
        cpu = {"value": round(want_total / t_all, 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
               "sample": f"first {m} reads of the same batch ({want_total} k-mers, {t_all:.1f} s on {cores} threads; "
                         f"{m1} reads, {t_one:.1f} s on 1 thread); GPU counts on the sample equal the oracle's bit-for-bit",
               "single_thread_value": round(m1 * kmers_per_read / t_one, 1), "host_cpus_visible": avail,
               "reference_python_1core": "0.13-0.21 M k-mers/s (BASELINE.md section 2, survey container)"}
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
                               f"({'canonical' if canonical else 'forward'}), inputs resident in HBM",
                   "k": k, "reads_per_gpu_per_step": n_reads, "read_len": L, "canonical": canonical,
                   "algo": {0: "auto", 1: "direct-atomics", 2: "lds-histogram"}[args.algo],
                   "sharding": f"reads x{world}, one RCCL reduce at the end" if world > 1 else "single GPU"},
        "gbase_per_s": round(world * args.steps * nbytes / elapsed / 1e9, 3),
        "count_only_ms_per_step": round(t_count / args.steps * 1e3, 4),
        "reduce_ms": round(reduce_ms, 3),
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if k <= 13:     # sha256 of the little-endian uint64 vector of the whole job (after the reduce for N > 1)
        import hashlib
        out["vector_sha256"] = hashlib.sha256(table.cpu().numpy().tobytes()).hexdigest()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
