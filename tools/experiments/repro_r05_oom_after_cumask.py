"""Round 5 (observed: yes -- segmentation fault in one run, a hang in another; without masks: no): does filling the device with torch (an out-of-memory hipMalloc inside torch's allocator) crash after an engine used CU-masked
streams (hipExtStreamCreateWithCUMask)?   python tools/experiments/repro_r05_oom_after_cumask.py {none|overlap|mask}"""
import os
import sys
os.environ["KDB_ALLOW_CU_MASKS"] = "1"
import numpy as np
import torch
sys.path.insert(0, ".")
import kmerdb_amd
from kmerdb_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "mask"
b, o = synth.reads(30000, 150, seed=3)
with kmerdb_amd.Engine(12) as eng:
    if mode in ("overlap", "mask"):
        eng.set_option("overlap", 1)
    if mode == "mask":
        eng.set_option("overlap_hist_cus", 64)
    for _ in range(3):
        eng.submit(b, o)
    eng.finish(copy=False)
print("engine done, mode", mode, flush=True)
torch.cuda.synchronize()
free, _ = torch.cuda.mem_get_info()
hog = [torch.empty(free - (1 << 30), dtype=torch.uint8, device="cuda")]
n = 0
for size in (256 << 20, 16 << 20, 1 << 20):
    while True:
        try:
            hog.append(torch.empty(size, dtype=torch.uint8, device="cuda"))
            n += 1
        except torch.OutOfMemoryError:
            break
print("filled the device:", n, "further blocks; no crash", flush=True)
