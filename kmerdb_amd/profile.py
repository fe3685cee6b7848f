"""kmerdb_amd.profile -- the `kmerdb profile` driver (reference kmerdb/__init__.py:1792-1858 profile,
:1862-2013 _profile): loop the input files through parse.parsefile (GPU), sum the vectors, build the YAML
metadata, write <output_name>.<k>.kdb.  Only the flags that matter on the path are kept."""
import os
import sys
from collections import OrderedDict

import numpy as np

from . import fileutil, parse, util
from .engine import Engine, KDB_N_DROP, KDB_N_EXPAND


def expand_inputs(inputs):
    """kmerdb/__init__.py:1822-1844: exactly one positional; a .txt/.tsv is a samplesheet (one path per line)."""
    if len(inputs) != 1:
        raise ValueError("kmerdb profile expects exactly one input (a sequence file or a .txt/.tsv samplesheet)")
    p = inputs[0]
    if p.endswith(".txt") or p.endswith(".tsv"):
        with open(p) as f:
            return [line.rstrip() for line in f if line.strip()]
    return [p]


def _workers_for(k, nfiles, device, workers):
    """How many files to count at the same time: each needs its own 4^k vector and scatter scratch on the device."""
    if workers is not None:
        return max(1, min(int(workers), nfiles))
    # a file keeps about four host threads busy (reader + splitter, md5, sha256, inflate): four files on a 16-core share, up to eight on a larger host
    cores = util.effective_cpus()
    want = min(max(4, min(8, cores // 4)), nfiles)
    if want <= 1:
        return 1
    try:
        import torch
        free_b, _ = torch.cuda.mem_get_info(device)
    except Exception:
        return 1
    per_engine = 8 * 4 ** k + (6 << 30)
    return max(1, min(want, int(free_b * 0.5 // per_engine)))


def _metadata(k, N, total, unique, do_not_canonicalize, file_metadata):
    """The header dictionary of _profile (kmerdb/__init__.py:1901-1936)."""
    unique_nullomers = N - unique if do_not_canonicalize is True else int((N / 2) - unique)
    return OrderedDict({
        "version": fileutil.VERSION,
        "metadata_blocks": 1,
        "k": k,
        "total_kmers": int(total),
        "unique_kmers": int(unique),
        "unique_nullomers": unique_nullomers,
        "sorted": False,
        "tags": [],
        "files": file_metadata,
    })


def profile(inputs, k, output_name, no_ambiguous=False, do_not_canonicalize=False, quiet=True, device=0, write=True, workers=None, timings=None):
    """-> (counts uint64[4**k], metadata OrderedDict, output_filepath|None).  Mirrors _profile (:1862-2013).

    Nothing on the host walks the 4^k bins: Sum(counts) and count_nonzero(counts) (:1901-1902) come from the device's sweep of
    the vector (np.sum + np.count_nonzero of the 8 GiB of k = 15 took the host 0.5 s), the one copy-back lands in pages touched
    by several threads, and the rows are written by the native pipeline.  A job of ONE file keeps its vector where it was counted
    (no accumulator, no fold).  `timings`: a dict that receives the wall-clock seconds of the job's stages.

    `counts = counts + counts_` (:1888-1891) stays on the device: every file's vector is folded into a second 4^k vector
    in HBM and only the sum is copied to the host, once.  Hashing a raw file (md5 + sha256, util.py:35-50) is slower than
    counting it, so `workers` files (default: up to 4, memory permitting) are read, hashed and counted at the same
    time, each by its own engine, all folding into one accumulator (up to 8 on hosts with more than 16 cores).  If a second vector does not fit (k = 17) the
    vectors are summed on the host as the reference does."""
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    if type(k) is not int:
        raise TypeError("k must be an int")
    t_start = time.perf_counter()
    stage = {}
    files = expand_inputs(list(inputs))
    N = 4 ** k
    n_mode = KDB_N_DROP if no_ambiguous else KDB_N_EXPAND
    W = _workers_for(k, len(files), device, workers)
    engines = []
    file_metadata = [None] * len(files)
    counts = None
    metadata, out = None, None
    totals = None                      # (Sum, count_nonzero) of the job's vector, from the device
    try:
        for _ in range(W):
            engines.append(Engine(k, canonicalize=not do_not_canonicalize, n_mode=n_mode, device=device))
        stage["engine_setup_s"] = time.perf_counter() - t_start
        if W > 1 and k >= 13:
            # each engine keeps an arena of scattered pages (k >= 13); left alone, each would size it for the whole device
            try:
                import torch
                free_b, _ = torch.cuda.mem_get_info(device)
                share = int((free_b - 8 * N) * 0.6 / W)                  # (8 N: the accumulator, allocated at the first fold)
            except Exception:
                share = 0
            for e in engines:
                e.set_option("pending_budget", max(share, 1 << 30))
        acc, lock = engines[0], threading.Lock()
        idle = list(engines)
        idle_lock = threading.Lock()

        def one(i):
            with idle_lock:
                eng = idle.pop()
            try:
                return parse.parsefile_folded(files[i], k, eng, replace_with_none=bool(no_ambiguous), into=acc, lock=lock)
            finally:
                with idle_lock:
                    idle.append(eng)

        t_count = time.perf_counter()
        try:
            folded = len(files) > 1
            if not folded:
                file_metadata[0] = parse.parsefile_folded(files[0], k, acc, replace_with_none=bool(no_ambiguous), fold=False)
            elif W == 1:
                # one engine: the next files' checksums are still started ahead of the counting
                sums, ahead = {}, 2
                try:
                    for i, f in enumerate(files):
                        for g in files[i:i + 1 + ahead]:
                            if g not in sums and type(g) is str and os.path.exists(g):
                                sums[g] = util.ChecksumJob(g)
                        file_metadata[i] = parse.parsefile_folded(f, k, acc, replace_with_none=bool(no_ambiguous), sums=sums.pop(f, None))
                finally:
                    for j in sums.values():
                        try:
                            j.result()
                        except Exception:
                            pass
            else:
                with ThreadPoolExecutor(W) as pool:
                    for i, md in enumerate(pool.map(one, range(len(files)))):
                        file_metadata[i] = md
            stage["count_s"] = time.perf_counter() - t_count
            # Sum and count_nonzero from the device's sweep of the job's vector (:1901-1902); then the one copy-back -- beside the row
            # writer when the .kdb is wanted: a chunk of rows is formatted as soon as its counts have arrived (kdb_copy_back_and_write_kdb_rows)
            t_copy = time.perf_counter()
            _, total, unique = acc.finish_folded(copy=False) if folded else acc.finish(copy=False)
            totals = (total, unique)
            if write:
                metadata = _metadata(k, N, total, unique, do_not_canonicalize, file_metadata)
                out = "{0}.{1}.kdb".format(output_name, k)                            # :1950
                counts, _ = fileutil.write_kdb_from_engine(out, dict(metadata), acc, folded=folded)
                stage["copy_back_and_write_kdb_s"] = time.perf_counter() - t_copy
            else:
                counts, _, _ = acc.finish_folded() if folded else acc.finish()
                stage["copy_back_s"] = time.perf_counter() - t_copy
        except MemoryError:
            if any(m is not None for m in file_metadata):
                raise
            counts = np.zeros(N, dtype="uint64")                                      # :1879-1881
            for i, sequence_file in enumerate(files):
                counts_, file_metadata[i], _ = parse.parsefile(sequence_file, k, replace_with_none=bool(no_ambiguous),
                                                               canonicalize=not do_not_canonicalize, engine=acc)
                counts = counts + counts_
    finally:
        t_close = time.perf_counter()
        for e in engines:
            e.close()
        stage["engine_close_s"] = time.perf_counter() - t_close
    if totals is not None:
        all_observed_kmers, unique_kmers = int(totals[0]), int(totals[1])             # :1901-1903, by the device's sweep
    else:
        all_observed_kmers = int(np.sum(counts))
        unique_kmers = int(np.count_nonzero(counts))
    unique_nullomers = N - unique_kmers if do_not_canonicalize is True else int((N / 2) - unique_kmers)
    if metadata is None:
        metadata = _metadata(k, N, all_observed_kmers, unique_kmers, do_not_canonicalize, file_metadata)      # :1926-1936
    if write and out is None:                        # (the host-summed fall-back of k = 17: the vector is on the host already)
        out = "{0}.{1}.kdb".format(output_name, k)                                    # :1950
        t_write = time.perf_counter()
        fileutil.write_kdb(out, dict(metadata), counts)
        stage["write_kdb_s"] = time.perf_counter() - t_write
    stage["total_s"] = time.perf_counter() - t_start
    if timings is not None:
        timings.update(stage)
    if not quiet:
        sys.stderr.write("Total k-mers processed: {0}\nUnique nullomer count:   {1}\nUnique {2}-mer count:     {3}\n".format(
            all_observed_kmers, unique_nullomers, k, unique_kmers))
    return counts, metadata, out


def main(argv=None):
    """`python -m kmerdb_amd profile -k K -o NAME input` -- the reference's profile flags (__init__.py:2084-2107)."""
    import argparse
    ap = argparse.ArgumentParser(prog="kmerdb_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    pp = sub.add_parser("profile", help="k-mer count profile of FASTA/FASTQ file(s) -> <name>.<k>.kdb")
    pp.add_argument("-k", type=int)
    pp.add_argument("--minK", type=int)
    pp.add_argument("--maxK", type=int)
    pp.add_argument("-o", "--output-name", required=True)
    pp.add_argument("--no-ambiguous", action="store_true", help="drop k-mers containing N instead of expanding them")
    pp.add_argument("--do-not-canonicalize", action="store_true")
    pp.add_argument("--quiet", action="store_true")
    pp.add_argument("--device", type=int, default=0)
    pp.add_argument("input", nargs="+")
    gp = sub.add_parser("graph", help="k-mer adjacency list of FASTA/FASTQ file(s) -> <name>.kdbg (__init__.py:2110-2130)")
    gp.add_argument("-k", type=int, required=True)
    gp.add_argument("--do-not-canonicalize", action="store_true")
    gp.add_argument("--replace-with-none", action="store_true")
    gp.add_argument("--quiet", action="store_true")
    gp.add_argument("--device", type=int, default=0)
    gp.add_argument("input", nargs="+")
    gp.add_argument("kdbg")
    a = ap.parse_args(argv)
    if a.cmd == "graph":
        from . import graph
        _, n = graph.make_graph(a.input, a.k, a.kdbg, quiet=a.quiet, do_not_canonicalize=a.do_not_canonicalize,
                                replace_with_none=a.replace_with_none, device=a.device)
        print(a.kdbg, n)
        return 0
    if a.k is not None:
        ks = [a.k]
    elif a.minK is not None and a.maxK is not None:                                   # :1846-1858
        ks = list(range(a.minK, a.maxK + 1))
    else:
        ap.error("either -k or --minK and --maxK are required")
    for k in ks:
        _, md, out = profile(a.input, k, a.output_name, no_ambiguous=a.no_ambiguous,
                             do_not_canonicalize=a.do_not_canonicalize, quiet=a.quiet, device=a.device)
        print(out)
    return 0
