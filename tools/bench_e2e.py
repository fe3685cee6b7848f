#!/usr/bin/env python3
"""End-to-end from files (SURVEY 8(d) region iii): parse.parsefile on one FASTQ, profile() on samplesheets of 1, 2, 4, 8
files (workers = files), plain and BGZF-compressed.  Prints one JSON object.  Not the headline metric (bench.py is)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmerdb_amd
from kmerdb_amd import fileutil, parse, profile, synth

k = 12
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
bases, offsets = synth.reads(n, 150, seed=synth.SEED0 + 2)
text = synth.fastq_text(bases, offsets)
out = {"k": k, "reads_per_file": n, "file_bytes": len(text)}
tmp = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
with tempfile.TemporaryDirectory(dir=tmp) as d:
    paths = []
    for i in range(8):
        p = os.path.join(d, f"s{i}.fq")
        open(p, "wb").write(text)
        paths.append(p)
    bz = os.path.join(d, "s.bgzf.fq.gz")
    with open(bz, "wb") as f:
        for i in range(0, len(text), 65280):
            f.write(fileutil._bgzf_member(text[i:i + 65280], 1))
    out["bgzf_bytes"] = os.path.getsize(bz)
    for rep in range(2):                                   # second round = steady state (rings pinned, scratch sized)
        t = time.perf_counter(); _, meta, _ = parse.parsefile(paths[0], k); dt = time.perf_counter() - t
        assert meta["total_kmers"] == n * (151 - k)
        out[f"parsefile_1file_gbase_s_round{rep}"] = round(n * 150 / dt / 1e9, 3)
        for nf in (1, 2, 4, 8):
            sheet = os.path.join(d, "sheet.txt")
            open(sheet, "w").write("\n".join(paths[:nf]) + "\n")
            t = time.perf_counter()
            _, md, _ = profile.profile([sheet], k, os.path.join(d, "o"), no_ambiguous=True, write=False, workers=nf)
            dt = time.perf_counter() - t
            assert md["total_kmers"] == nf * n * (151 - k)
            out[f"profile_{nf}files_gbase_s_round{rep}"] = round(nf * n * 150 / dt / 1e9, 3)
        t = time.perf_counter(); _, meta, _ = parse.parsefile(bz, k); dt = time.perf_counter() - t
        assert meta["total_kmers"] == n * (151 - k)
        out[f"parsefile_bgzf_gbase_s_round{rep}"] = round(n * 150 / dt / 1e9, 3)
print(json.dumps(out))
