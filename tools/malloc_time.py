#!/usr/bin/env python3
"""How long hipMalloc / hipFree take by size (the page arena of the two-level path is grown with them)."""
import ctypes
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmerdb_amd import _abi  # noqa: E402

hip = ctypes.CDLL(_abi._preload_hip_runtime() or "libamdhip64.so")      # (the copy PyTorch bundles, as the engine uses)
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipDeviceSynchronize()
for gib in [float(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128]:
    for rep in range(2):
        p = ctypes.c_void_p()
        n = int(gib * (1 << 30))
        t0 = time.perf_counter()
        rc = hip.hipMalloc(ctypes.byref(p), n)
        t1 = time.perf_counter()
        if rc:
            print(f"{gib} GiB: hipMalloc failed rc={rc}")
            break
        hip.hipMemset(p, 0, min(n, 1 << 20))
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipFree(p)
        hip.hipDeviceSynchronize()
        t3 = time.perf_counter()
        print(f"{gib:6.1f} GiB rep {rep}: hipMalloc {1e3 * (t1 - t0):9.1f} ms   first touch {1e3 * (t2 - t1):7.1f} ms   hipFree {1e3 * (t3 - t2):9.1f} ms", flush=True)
