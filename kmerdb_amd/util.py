"""Host helpers mirrored from the reference's kmerdb/util.py (checksum, format sniffing)."""
import hashlib
import os


def checksum(filepath):
    """md5 and sha256 of the raw (possibly gzipped) file bytes -- kmerdb/util.py:35-50."""
    if type(filepath) is not str:
        raise TypeError("kmerdb_amd.util.checksum() expects a str as its argument")
    elif not os.path.exists(filepath):
        raise IOError("kmerdb_amd.util.checksum could not find '{}' on the filesystem".format(filepath))
    hash_md5 = hashlib.md5()
    hash_sha256 = hashlib.sha256()
    with open(filepath, "rb") as ifile:
        for chunk in iter(lambda: ifile.read(1 << 20), b""):
            hash_md5.update(chunk)
            hash_sha256.update(chunk)
    return (hash_md5.hexdigest(), hash_sha256.hexdigest())


class ChecksumJob:
    """checksum(filepath) on two background threads (hashlib releases the GIL), so that hashing the raw file
    overlaps parsing and counting.  result() -> (md5, sha256), identical to checksum()."""

    def __init__(self, filepath):
        import threading
        if type(filepath) is not str:
            raise TypeError("kmerdb_amd.util.ChecksumJob expects a str as its argument")
        elif not os.path.exists(filepath):
            raise IOError("kmerdb_amd.util.ChecksumJob could not find '{}' on the filesystem".format(filepath))
        self._out = {}
        self._err = []
        self.seconds = {}                 # thread time of each digest (bench.py: which stage bounds a file end to end)

        def run(name):
            import time
            try:
                t0 = time.perf_counter()
                h = hashlib.new(name)
                with open(filepath, "rb") as f:
                    for chunk in iter(lambda: f.read(4 << 20), b""):
                        h.update(chunk)
                self._out[name] = h.hexdigest()
                self.seconds[name] = time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001 - re-raised in result()
                self._err.append(e)

        self._threads = [threading.Thread(target=run, args=(n,), daemon=True) for n in ("md5", "sha256")]
        for t in self._threads:
            t.start()

    def result(self):
        for t in self._threads:
            t.join()
        if self._err:
            raise self._err[0]
        return self._out["md5"], self._out["sha256"]


def _cgroup_cpu_limit(proc_cgroup="/proc/self/cgroup", sys_root="/sys/fs/cgroup"):
    """CPUs' worth of time the cgroup(s) of this process may use (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1; the
    smallest along the path up to the root), or None if there is no quota.  (The two paths are arguments for the tests.)"""
    best = None

    def take(quota, period):
        nonlocal best
        if quota > 0 and period > 0:
            v = quota / period
            best = v if best is None else min(best, v)

    try:
        lines = open(proc_cgroup).read().split("\n")
    except OSError:
        return None
    for line in lines:
        parts = line.split(":", 2)
        if len(parts) != 3:
            continue
        _, ctrl, path = parts
        if ctrl == "":                                       # v2: 0::/path
            base, v2 = sys_root, True
        elif "cpu" in ctrl.split(","):                       # v1: N:cpu,cpuacct:/path
            base, v2 = os.path.join(sys_root, ctrl), False
            if not os.path.isdir(base):
                base = os.path.join(sys_root, "cpu")
        else:
            continue
        path = path.strip("/")
        while True:
            d = os.path.join(base, path) if path else base
            try:
                if v2:
                    q, per = open(os.path.join(d, "cpu.max")).read().split()[:2]
                    if q != "max":
                        take(int(q), int(per))
                else:
                    take(int(open(os.path.join(d, "cpu.cfs_quota_us")).read()), int(open(os.path.join(d, "cpu.cfs_period_us")).read()))
            except (OSError, ValueError):
                pass
            if not path:
                break
            path = os.path.dirname(path)
    return best


def effective_cpus():
    """How many threads of CPU-bound work this process can run at once: the CPUs it may be scheduled on, cut down to its cgroup's
    CPU quota if it has one (KDB_CPUS overrides).  On a GPU box that shows 256 CPUs to a container with a 16-CPU quota, 64 threads
    of deflate are slower than 16: the quota throttles all of them, also the one everybody else waits for."""
    env = os.environ.get("KDB_CPUS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = _cgroup_cpu_limit()
    if q is not None:
        n = min(n, max(1, int(q + 0.999)))
    return max(1, n)


def is_gz_file(filepath):
    """Content sniff, like kmerdb/util.py:80-88 (which tries gzip.open + readline)."""
    with open(filepath, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def is_fasta(fname):
    """kmerdb/util.py:120-126: format is chosen by filename suffix."""
    if type(fname) is not str:
        raise TypeError("kmerdb_amd.util.is_fasta() expects a str argument")
    return fname.endswith((".fna", ".fna.gz", ".fa.gz", ".fa", ".fasta", ".fasta.gz"))


def is_fastq(fname):
    """kmerdb/util.py:128-134."""
    if type(fname) is not str:
        raise TypeError("kmerdb_amd.util.is_fastq() expects a str argument")
    return fname.endswith((".fastq", ".fastq.gz", ".fq.gz", ".fq"))
