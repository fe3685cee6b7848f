"""Round 5: what the k = 15 `profile` job costs around the kernels, stage by stage, on the GPU box's host.
python tools/experiments/exp_r05_job_k15.py [k] [reads] -> JSON lines on stdout."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kmerdb_amd                                    # noqa: E402
from kmerdb_amd import fileutil, synth               # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
L = 150
out = {"k": k, "reads": n_reads, "cpus": len(os.sched_getaffinity(0))}


def tick(name, t0):
    out[name] = round(time.perf_counter() - t0, 4)
    print(json.dumps({name: out[name]}), flush=True)


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


out["mem_available_gb"] = round(mem_available_gb(), 1)
sv = os.statvfs("/dev/shm")
out["shm_free_gb"] = round(sv.f_bavail * sv.f_frsize / 1e9, 1)
print(json.dumps(out), flush=True)
need_gb = 8 * 4 ** k / 1e9 * 3.5
if out["mem_available_gb"] < need_gb + 8 or out["shm_free_gb"] < 8 * 4 ** k / 1e9 * 1.2:
    print(json.dumps({"skipped": "not enough host memory for k=%d (%.0f GB wanted)" % (k, need_gb)}))
    sys.exit(0)

import torch                                          # noqa: E402
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(synth.SEED0)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
d_bases = lut[torch.randint(0, 4, (n_reads * L,), generator=g, device=dev)]
d_offs = (torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * L)
torch.cuda.synchronize()

t = time.perf_counter()
eng = kmerdb_amd.Engine(k, canonicalize=True, device=0)
tick("engine_create_s", t)
t = time.perf_counter()
eng.submit_device(d_bases.data_ptr(), n_reads * L, d_offs.data_ptr(), n_reads)
eng.sync()
tick("count_first_batch_s", t)
t = time.perf_counter()
_, total, unique = eng.finish(copy=False)
tick("finish_nocopy_s", t)
t = time.perf_counter()
counts, total, unique = eng.finish(copy=True)
tick("finish_copy_fresh_array_s", t)
t = time.perf_counter()
import ctypes                                         # noqa: E402
tot = ctypes.c_uint64(0)
uni = ctypes.c_uint64(0)
kmerdb_amd._abi.check(kmerdb_amd._abi.lib().kdb_finish(eng._h, counts.ctypes.data, ctypes.byref(tot), ctypes.byref(uni)))
tick("finish_copy_touched_array_s", t)
out["bytes"] = int(counts.nbytes)
t = time.perf_counter()
s = int(np.sum(counts))
tick("np_sum_s", t)
t = time.perf_counter()
u = int(np.count_nonzero(counts))
tick("np_count_nonzero_s", t)
assert s == total and u == unique
t = time.perf_counter()
nul = np.flatnonzero(counts == 0).astype("uint64")
tick("np_flatnonzero_s", t)
out["nullomers"] = int(nul.size)
del nul
t = time.perf_counter()
eng.close()
tick("engine_close_s", t)

md = {"version": fileutil.VERSION, "metadata_blocks": 1, "k": k, "total_kmers": total, "unique_kmers": unique, "unique_nullomers": 0,
      "sorted": False, "tags": [], "files": []}
tmp = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
for enc, threads in (("rows", 64), ("rows", 16), ("rows", 32), ("rows", 128), ("zlib", 64)):
    if enc == "zlib" and k > 13:
        continue
    p = os.path.join(tmp, "exp_r05.%d.kdb" % k)
    t = time.perf_counter()
    nb = fileutil.write_kdb(p, md, counts, nthreads=threads, encoder=enc)
    dt = time.perf_counter() - t
    rec = {"write_kdb": enc, "threads": threads, "s": round(dt, 3), "rows_per_s": round(4 ** k / dt), "text_gb": round(nb * 65536 / 1e9, 2),
           "text_gb_per_s": round(nb * 65536 / 1e9 / dt, 2), "file_gb": round(os.path.getsize(p) / 1e9, 3)}
    print(json.dumps(rec), flush=True)
    os.remove(p)
print(json.dumps(out))
