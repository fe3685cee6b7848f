#!/usr/bin/env python3
"""Diagnostic (GPU box): time scatter_bases_kernel with parts switched off (build: tools/sc_phases.sh makes libkdbhip_prof.so).
Results are meaningless in these modes; only the kernel time is read."""
import os, sys
sys.path.insert(0, ".")
os.environ["KDB_LIB"] = os.path.abspath("kmerdb_amd/libkdbhip_prof.so")
import torch, kmerdb_amd
n, L, k = 10_000_000, 150, int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = torch.Generator(device="cuda"); g.manual_seed(1)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
d_b = lut[torch.randint(0, 4, (n * L,), generator=g, device="cuda", dtype=torch.uint8).long()]
d_o = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
for ab in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, 2, 3)):
    eng = kmerdb_amd.Engine(k, algo=2)
    eng.set_option("sc_ablate", ab)
    eng.submit_device(d_b.data_ptr(), n * L, d_o.data_ptr(), n); eng._lib.kdb_sync(eng._h)
    eng.prof_enable(True); eng.prof_reset()
    for _ in range(5): eng.submit_device(d_b.data_ptr(), n * L, d_o.data_ptr(), n)
    eng._lib.kdb_sync(eng._h)
    pr = eng.prof()
    print("ablate", ab, {kk: round(v[0] / max(v[1], 1), 3) for kk, v in pr.items() if v[1]}, flush=True)
    eng.set_option("sc_ablate", 0)
    eng.close()
