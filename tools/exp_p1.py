import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import kmerdb_amd
n, L, k = 10_000_000, 150, 12
g = torch.Generator(device="cuda"); g.manual_seed(1)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
d_b = lut[torch.randint(0, 4, (n * L,), generator=g, device="cuda", dtype=torch.uint8).long()]
d_o = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
eng = kmerdb_amd.Engine(k)
eng.submit_device(d_b.data_ptr(), n * L, d_o.data_ptr(), n); eng._lib.kdb_sync(eng._h)
eng.prof_enable(True); eng.prof_reset()
for _ in range(5): eng.submit_device(d_b.data_ptr(), n * L, d_o.data_ptr(), n)
eng._lib.kdb_sync(eng._h)
print(os.environ.get("KDB_LIB", "default").split("/")[-1], {k2: round(v[0] / max(v[1], 1), 3) for k2, v in eng.prof().items() if v[1]})
