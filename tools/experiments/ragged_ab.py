"""ms per step of the ragged, N-bearing batches of bench.py (10 M reads of 35..150 bp) at several N densities, N-drop and N-expansion mode.
Usage: KDB_LIB=$PWD/kmerdb_amd/libkdbhip.so python tools/experiments/ragged_ab.py [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
import torch
import kmerdb_amd
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda', 0)
n = 10_000_000
out = []
for p_n in (0.005, 0.0005, 0.0):
    rb, ro, nbytes = b.ragged_batch(torch, dev, n, 35, 150, p_n, 20240612 + 77)
    for name, mode in (("drop", 0), ("expand", 1)):
        with kmerdb_amd.Engine(k, canonicalize=True, n_mode=mode, device=0) as e:
            if k >= 13:
                e.set_option("arena_batches", 16)
            for _ in range(2):
                e.submit_device(rb.data_ptr(), nbytes, ro.data_ptr(), n)
            e.sync()
            e.prof_enable(True); e.prof_reset()
            t = time.perf_counter()
            for _ in range(10):
                e.submit_device(rb.data_ptr(), nbytes, ro.data_ptr(), n)
            e.sync()
            dt = (time.perf_counter() - t) / 10
            pk = {kn: round(ms / 10, 3) for kn, (ms, c) in e.prof().items() if c}
            e.finish(copy=False)
        out.append("k=%d p_N=%g %s: %.3f ms  %s" % (k, p_n, name, dt * 1e3, pk))
    del rb, ro
print("\n".join(out))
