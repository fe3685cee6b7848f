#!/usr/bin/env python3
"""Randomised differential test on the GPU box: random k / read lengths / N density / strand mode / N mode / algo /
engine options / submit chunking, every case compared with the CPU oracle (full vector for k <= 13, sparse for k >= 14).
Usage: python tests/fuzz_gpu.py [seconds] [seed] [k,k,...]      prints one line per case and a summary; exit 1 on a mismatch.
Also collected by pytest -m gpu through tests/test_gpu_fuzz.py (run_cases with a case count instead of a time budget)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

LET = np.frombuffer(b"ACGTN", dtype=np.uint8)
K_MIX = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 12, 12, 13, 13, 14, 14, 15, 15, 16, 17]


def draw_case(rng, only_k=None):
    """One random case: (description dict, bases, offsets)."""
    k = int(rng.choice(only_k if only_k else K_MIX))
    canon = bool(rng.integers(0, 2))
    expand = bool(rng.integers(0, 2))                 # N expansion at every k (two-level scatter kernels included)
    algo = int(rng.choice([0, 1, 2, 2]))
    uniform = bool(rng.integers(0, 2))
    nreads = int(rng.choice([1, 2, 7, 100, 1000, 5000]))
    if uniform:
        L = int(rng.integers(k, k + 300))
        lens = np.full(nreads, L)
    else:
        lens = rng.integers(k, k + int(rng.choice([5, 300, 40000 // max(1, nreads // 10)])), size=nreads)
    p_n = float(rng.choice([0.0, 0.0, 0.001, 0.01])) if not expand else float(rng.choice([0.0, 0.0005, 0.002 if k <= 13 else 0.0005]))
    if expand and 9 <= k <= 12 and nreads <= 1000 and rng.integers(0, 3) == 0:
        p_n = float(rng.choice([0.01, 0.04]))        # tiles dense with N's (more than the image's list of N positions holds at 0.04), windows with three and more N's
    total = int(lens.sum())
    bases = LET[rng.choice(5, size=total, p=[(1 - p_n) / 4] * 4 + [p_n])].copy()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    opts = {}
    if rng.integers(0, 3) == 0:
        opts["stage_bytes"] = int(rng.choice([4096, 65536, 1 << 20]))
        opts["stage_reads"] = int(rng.choice([3, 64, 4096]))
    if k >= 13 and rng.integers(0, 2):
        opts["defer_flush"] = int(rng.integers(0, 2))
    if k >= 8 and rng.integers(0, 4) == 0:
        opts["sc_grid"] = int(rng.choice([1, 7, 64]))
    if k >= 8 and rng.integers(0, 3) == 0:
        opts["sc_lo_bits"] = int(rng.choice([1, 3, 6, 9, 12, 14, 15]))     # where the bucket field sits in the id
    if k >= 8 and rng.integers(0, 4) == 0:
        opts["sc_contig_pages"] = 1
    if k >= 15 and rng.integers(0, 2):
        opts["accum_bytes"] = int(rng.choice([0, 1 << 20]))
    if k <= 8 and rng.integers(0, 4) == 0:
        opts["smallk_old"] = 1                       # k <= 7: count_lds_kernel, k = 8: the paged scatter (the paths before the one-CU LDS histogram)
    if k <= 7 and rng.integers(0, 4) == 0:
        opts["sc_grid"] = int(rng.choice([1, 7, 64]))
    if k == 13 and rng.integers(0, 3) == 0:
        opts["one_level_max_k"] = 12                 # k = 13 through the two-level path instead of the 1024-ring kernel
    # the forms that write 64-byte lines (two workgroups of 512 threads per CU; the default since round 5 writes 128-byte pieces)
    if 8 <= k <= 12 and rng.integers(0, 4) == 0:
        opts["sc_wide_lines"] = 0
    if k >= 13 and rng.integers(0, 4) == 0:
        opts["l1_wide_lines"] = 0
    if k >= 13 and rng.integers(0, 4) == 0:
        opts["l2_wide_lines"] = 0
    if 13 <= k <= 15 and rng.integers(0, 4) == 0:
        opts["l1_one_round"] = 0                     # level 1 with 256 rings of 128 elements, two placement rounds per tile
    nsub = int(rng.choice([1, 1, 2, 5]))
    cuts = sorted(set([0, nreads] + [int(x) for x in rng.integers(0, nreads + 1, size=nsub - 1)]))
    # per piece: 0 host submit, 1 handed over in HBM (kdb_submit_device: no staging, no accumulation), 2 host submit followed by a sync
    how = [int(x) for x in rng.choice([0, 0, 1, 1, 2], size=len(cuts) - 1)] if rng.integers(0, 2) else [0] * (len(cuts) - 1)
    # kdb_reset before piece `reset_at` (what was submitted before it -- pending batches of the two-level path included -- is forgotten)
    reset_at = int(rng.integers(1, len(cuts) - 1)) if len(cuts) > 2 and rng.integers(0, 4) == 0 else 0
    if k >= 14 and rng.integers(0, 3) == 0:
        opts["arena_batches"] = int(rng.choice([1, 2]))          # a small arena: flushes forced by a full arena, growth
        opts["arena_grow"] = int(rng.choice([0, 1, 2]))
    if 9 <= k <= 13 and not expand and rng.integers(0, 4) == 0:
        opts["overlap"] = 1                          # the scatter kernel of a piece beside the histogram pass of the piece before (two page sets, side list)
        # (not with CU masks: on ROCm 7.2 a process that has created a CU-masked stream crashes or hangs in the runtime when a later
        #  hipMalloc runs out of memory -- tools/experiments/repro_r05_oom_after_cumask.py -- and this suite fills the device on purpose)
    desc = dict(k=k, canon=canon, expand=expand, algo=algo, uniform=uniform, nreads=nreads, bases=total, p_n=p_n, opts=opts, cuts=cuts, how=how, reset_at=reset_at)
    return desc, bases, offsets


def check_case(desc, bases, offsets):
    """Count the case on the GPU (through the C ABI) and compare with the oracle -> bool."""
    import torch
    import kmerdb_amd
    from oracle import kmer_oracle as oracle
    k, canon, expand, nreads, cuts = desc["k"], desc["canon"], desc["expand"], desc["nreads"], desc["cuts"]
    omode = oracle.N_EXPAND if expand else oracle.N_DROP
    with kmerdb_amd.Engine(k, canonicalize=canon, n_mode=1 if expand else 0, algo=desc["algo"]) as eng:
        for name, v in desc["opts"].items():
            eng.set_option(name, v)
        keep = []
        how, reset_at = desc.get("how") or [0] * (len(cuts) - 1), desc.get("reset_at", 0)
        for pi, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            if reset_at and pi == reset_at:
                eng.reset()
            if b > a:
                o = offsets[a:b + 1] - offsets[a]
                piece = bases[int(offsets[a]):int(offsets[b])]
                if how[pi] == 1 and piece.size:
                    d_b = torch.from_numpy(np.ascontiguousarray(piece)).cuda()
                    d_o = torch.from_numpy(o.astype(np.uint64).view(np.int64).copy()).cuda()
                    eng.submit_device(d_b.data_ptr(), piece.size, d_o.data_ptr(), b - a)
                    keep.append((d_b, d_o))              # (asynchronous: the buffers live until the sync)
                else:
                    eng.submit(piece, o.astype(np.uint64))
                    if how[pi] == 2:
                        eng.sync()
        if reset_at:                                      # what counts: the reads from the reset on
            r0 = cuts[reset_at]
            bases, offsets, nreads = bases[int(offsets[r0]):], offsets[r0:] - offsets[r0], nreads - r0
        if k <= 13:
            got, tot, uniq = eng.finish()
            want, want_total = oracle.c_count(bases, offsets, k, canon, omode)
            return bool(tot == want_total and uniq == int(np.count_nonzero(want)) and np.array_equal(got, want))
        _, tot, uniq = eng.finish(copy=False)
        ids = np.concatenate([oracle.c_shred(bytes(bases[int(offsets[r]):int(offsets[r + 1])]), k, canon, omode)[0]
                              for r in range(nreads)]) if nreads else np.zeros(0, np.uint64)
        u, c = np.unique(ids, return_counts=True)
        t = eng.table_tensor()
        g = t[torch.as_tensor(u.astype(np.int64), device=t.device)].cpu().numpy().astype(np.uint64)
        return bool(tot == ids.size and uniq == u.size and np.array_equal(g, c.astype(np.uint64)))


def run_cases(seed, budget_s=None, max_cases=None, only_k=None, verbose=True):
    """-> (cases run, description of the first failing case or None)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t_end = time.time() + budget_s if budget_s else None
    ncase = 0
    while (t_end is None or time.time() < t_end) and (max_cases is None or ncase < max_cases):
        desc, bases, offsets = draw_case(rng, only_k)
        try:
            ok = check_case(desc, bases, offsets)
        except Exception as e:  # noqa: BLE001 - an engine error is a failing case too: name the case before it goes up
            print("FAIL " + json.dumps(desc) + "  raised %s: %s" % (type(e).__name__, e), flush=True)
            raise
        ncase += 1
        if verbose:
            print(("ok   " if ok else "FAIL ") + json.dumps(desc), flush=True)
        if not ok:
            return ncase, desc
    return ncase, None


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    only_k = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    n, bad = run_cases(seed, budget_s=budget, only_k=only_k)
    if bad is not None:
        sys.exit(1)
    print(f"{n} cases, all equal to the oracle (seed {seed})")
