#!/usr/bin/env python3
"""Do the scatter kernels of two engines overlap usefully on one device?  One engine with 512 persistent workgroups against two
engines (own streams, one thread each) with 256 each, same total work.  -> stdout, one line per case."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import kmerdb_amd  # noqa: E402
from bench import synthetic_batch  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n_reads, L = 10_000_000, 150
dev = torch.device("cuda:0")
batches = [synthetic_batch(torch, dev, n_reads, L, 77 + i) for i in range(2)]
torch.cuda.synchronize()


def run(engines, per_engine):
    def work(i):
        e = engines[i]
        b, o = batches[i % 2]
        for _ in range(per_engine):
            e.submit_device(b.data_ptr(), n_reads * L, o.data_ptr(), n_reads)
        e.sync()
    for e in engines:                                            # warm-up: arena, scratch
        b, o = batches[0]
        for _ in range(10):
            e.submit_device(b.data_ptr(), n_reads * L, o.data_ptr(), n_reads)
        e.sync()
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(engines))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / (per_engine * len(engines))


for grids in ([512], [256, 256], [512, 512], [384, 128]):
    engines = [kmerdb_amd.Engine(k, canonicalize=True, device=0) for _ in grids]
    try:
        for e, g in zip(engines, grids):
            e.set_option("sc_grid", g)
            e.set_option("arena_grow", 0)
        ms = run(engines, steps // len(engines))
        print(f"k={k} grids={grids}: {ms:.3f} ms per batch (aggregate)", flush=True)
    finally:
        for e in engines:
            e.close()
