"""ctypes binding of libkdbhip.so (C ABI declared in include/kdbhip.h).

This is the stub a kmerdb maintainer would add next to kmerdb/parse.py to call
the MI355X engine.  There is no CPU fallback: if the
shared library is missing or HIP has no device, every counting call raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KDB_LIB") or os.path.join(_HERE, "libkdbhip.so")     # KDB_LIB: A/B a second build in one run
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

KDB_OK, KDB_ERR_ARG, KDB_ERR_HIP, KDB_ERR_SHORT_READ, KDB_ERR_BAD_RESIDUE, KDB_ERR_NOMEM, KDB_ERR_STATE = range(7)
KDB_N_DROP, KDB_N_EXPAND = 0, 1
KDB_SUBMIT_PINNED, KDB_SUBMIT_CONTINUES = 1, 2
KDB_N_KERNELS = 7
ABI_VERSION = 6

# every symbol include/kdbhip.h declares: (name, restype, argtypes)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
SYMBOLS = (
    ("kdb_abi_version", ctypes.c_int, []),
    ("kdb_last_error", ctypes.c_char_p, []),
    ("kdb_device_count", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("kdb_create", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
    ("kdb_destroy", ctypes.c_int, [_vp]),
    ("kdb_reset", ctypes.c_int, [_vp]),
    ("kdb_submit", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t]),
    ("kdb_submit_pinned", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t]),
    ("kdb_submit_ex", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int]),
    ("kdb_host_alloc", ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    ("kdb_host_free", ctypes.c_int, [_vp]),
    ("kdb_submit_device", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t]),
    ("kdb_submit_device_const", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t]),
    ("kdb_sync", ctypes.c_int, [_vp]),
    ("kdb_finish", ctypes.c_int, [_vp, _vp, _u64p, _u64p]),
    ("kdb_table_stats", ctypes.c_int, [_vp, _vp, _u64p, _u64p]),
    ("kdb_nullomers", ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_uint64, _u64p]),
    ("kdb_reduce", ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    ("kdb_fold_file", ctypes.c_int, [_vp, _u64p, _u64p]),
    ("kdb_fold_file_into", ctypes.c_int, [_vp, _vp, _u64p, _u64p]),
    ("kdb_finish_folded", ctypes.c_int, [_vp, _vp, _u64p, _u64p]),
    ("kdb_create_ids", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    ("kdb_table", ctypes.c_int, [_vp, ctypes.POINTER(_vp), _u64p]),
    ("kdb_error_counts", ctypes.c_int, [_vp, _u64p, _u64p]),
    ("kdb_shred", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_window_ids", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    ("kdb_parse_fastq", ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_parse_fastq_mt", ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp,
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]),
    ("kdb_parse_fasta", ctypes.c_int, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_parse_fasta_chunk", ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp,
                                             ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                                             ctypes.POINTER(ctypes.c_int)]),
    ("kdb_bgzf_inflate", ctypes.c_int, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_bgzf_scan", ctypes.c_int, [ctypes.c_char_p, _vp, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_gz_open", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    ("kdb_gz_read", ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    ("kdb_gz_close", ctypes.c_int, [_vp]),
    ("kdb_write_kdb_rows", ctypes.c_int, [ctypes.c_char_p, _vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _u64p]),
    ("kdb_write_kdb_rows_ex", ctypes.c_int, [ctypes.c_char_p, _vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p]),
    ("kdb_copy_back_and_write_kdb_rows", ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p]),
    ("kdb_read_kdb_rows", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, _vp, _vp, _vp, ctypes.c_int, _u64p]),
    ("kdb_format_frequency", ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]),
    ("kdb_hbm_pattern_count", ctypes.c_int, []),
    ("kdb_hbm_pattern_name", ctypes.c_char_p, [ctypes.c_int]),
    ("kdb_hbm_pattern_probe", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    ("kdb_prof_enable", ctypes.c_int, [_vp, ctypes.c_int]),
    ("kdb_prof_reset", ctypes.c_int, [_vp]),
    ("kdb_prof_get", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), _u64p]),
    ("kdb_prof_kernel_name", ctypes.c_char_p, [ctypes.c_int]),
    ("kdb_set_option", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    ("kdb_get_option", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
)


class KdbHipError(RuntimeError):
    """libkdbhip.so is missing, or a HIP runtime call failed."""


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 the engine into kmerdb_amd/libkdbhip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(INCLUDE, "kdbhip.h"))
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
           "-I", INCLUDE, "-o", LIB_PATH] + srcs + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7); if libkdbhip.so pulled in /opt/rocm's copy first and torch were imported later,
    the process would hold two HIP runtimes and the second one sees no GPU.  Loading torch's copy
    first (without importing torch) makes libkdbhip.so's DT_NEEDED resolve to it by SONAME."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
    return cand


def lib():
    """Load libkdbhip.so and bind every declared symbol; raises KdbHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KdbHipError(
            f"{LIB_PATH} not found: the MI355X engine is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or kmerdb_amd._abi.build()). kmerdb_amd has no CPU fallback.")
    _preload_hip_runtime()
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise KdbHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise KdbHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    v = L.kdb_abi_version()
    if v != ABI_VERSION:
        raise KdbHipError(f"libkdbhip ABI version {v}, expected {ABI_VERSION}")
    _lib = L
    return L


def last_error():
    msg = lib().kdb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a C status to the Python exception the reference's callers expect."""
    if rc == KDB_OK:
        return
    msg = last_error()
    if rc in (KDB_ERR_SHORT_READ, KDB_ERR_BAD_RESIDUE, KDB_ERR_ARG):
        raise ValueError(msg)
    if rc == KDB_ERR_NOMEM:
        raise MemoryError(msg)
    raise KdbHipError(f"[status {rc}] {msg}")
