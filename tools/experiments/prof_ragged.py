"""Diagnostic (GPU box): phase clocks of scatter_bases_kernel (-DKDB_SC_PROF build) on the ragged, N-bearing batch of bench.py,
N-drop against N-expansion mode.  Usage: KDB_LIB=$PWD/kmerdb_amd/libkdbhip_prof.so python tools/experiments/prof_ragged.py [k]"""
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
p_n = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
rb, ro, nbytes = b.ragged_batch(torch, dev, n, 35, 150, p_n, 20240612 + 77)
for name, mode in (("drop", 0), ("expand", 1)):
    with kmerdb_amd.Engine(k, canonicalize=True, n_mode=mode, device=0) as e:
        e.submit_device(rb.data_ptr(), nbytes, ro.data_ptr(), n)
        e.sync()
        sys.stderr.write("==== %s (3 steps)\n" % name); sys.stderr.flush()
        e.prof_enable(True); e.prof_reset()
        t = time.perf_counter()
        for _ in range(3):
            e.submit_device(rb.data_ptr(), nbytes, ro.data_ptr(), n)
        e.sync()
        dt = (time.perf_counter() - t) / 3
        sys.stderr.write("%s: %.3f ms per step; %s\n" % (name, dt * 1e3, {kn: round(ms / 3, 3) for kn, (ms, c) in e.prof().items() if c}))
