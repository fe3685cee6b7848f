#!/usr/bin/env python3
"""Timed regions (ii) and (iii) of SURVEY 8(d): host-fed pipeline (pageable and pinned buffers, parsing
excluded) and end-to-end from a FASTQ file.  Prints one JSON object. Not the headline metric (bench.py is)."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import kmerdb_amd  # noqa: E402
from kmerdb_amd import parse, synth  # noqa: E402

k = 12
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
bases, offsets = synth.reads(n, 150, seed=synth.SEED0 + 2)
out = {"k": k, "reads": n, "bases": int(bases.size)}

with kmerdb_amd.Engine(k) as eng:
    eng.submit(bases[:15_000_000], offsets[:100_001]); eng.sync(); eng.reset()      # warm-up (allocations)
    for name, threads in (("pageable_1thread", 1), ("pageable_8threads", 8)):
        eng.set_option("copy_threads", threads)
        t = time.perf_counter(); eng.submit(bases, offsets); eng.sync(); dt = time.perf_counter() - t
        out[name + "_gbase_s"] = round(bases.size / dt / 1e9, 2)
        _, total, _ = eng.finish(copy=False); eng.reset()
        assert total == n * (151 - k)
    pb = kmerdb_amd.pinned_empty(bases.size); pb[:] = bases
    t = time.perf_counter(); eng.submit_pinned(pb, offsets); eng.sync(); dt = time.perf_counter() - t
    out["pinned_gbase_s"] = round(bases.size / dt / 1e9, 2)

m = min(n, 1_000_000)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "synthetic.fq")
    with open(path, "wb") as f:
        f.write(synth.fastq_text(bases[:m * 150], offsets[:m + 1]))
    t = time.perf_counter()
    counts, meta, _ = parse.parsefile(path, k)
    dt = time.perf_counter() - t
    out["end_to_end_fastq_reads"] = m
    out["end_to_end_fastq_gbase_s"] = round(m * 150 / dt / 1e9, 3)
    out["end_to_end_seconds"] = round(dt, 2)
    assert meta["total_kmers"] == m * (151 - k)
print(json.dumps(out))
