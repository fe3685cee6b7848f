"""Round 5: what does the RAGGED form of scatter_bases_kernel cost by itself?  The headline batch (10 M reads of 150 bases) once as it is (records of one length: record starts
computed) and once with its last record one base shorter (lens_kernel then calls the batch ragged: record starts from the offsets), same residues otherwise.
Usage (GPU box): python tools/experiments/exp_r05_ragged_kernel_on_uniform.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
import torch
import kmerdb_amd
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dev = torch.device('cuda', 0)
n, L, k = 10_000_000, 150, (int(sys.argv[1]) if len(sys.argv) > 1 else 12)
offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1234)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
step = 1 << 27
for s in range(0, n * L, step):
    e = min(n * L, s + step)
    bases[s:e] = lut[torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8).long()]
offs_r = offs.clone(); offs_r[-1] -= 1
torch.cuda.synchronize()
for name, o, nb in (("uniform", offs, n * L), ("last record one base shorter", offs_r, n * L - 1)):
    with kmerdb_amd.Engine(k, canonicalize=True, n_mode=0, device=0) as e:
        for _ in range(3):
            e.submit_device(bases.data_ptr(), nb, o.data_ptr(), n)
        e.sync()
        e.prof_enable(True); e.prof_reset()
        t = time.perf_counter()
        for _ in range(50):
            e.submit_device(bases.data_ptr(), nb, o.data_ptr(), n)
        e.sync()
        dt = (time.perf_counter() - t) / 50
        print("%-32s %.3f ms per step; %s" % (name, dt * 1e3, {kn: round(ms / 50, 4) for kn, (ms, c) in e.prof().items() if c}), flush=True)
