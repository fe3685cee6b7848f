"""CPU: the C-ABI library loads and exports every symbol include/kdbhip.h declares; host logic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import kmerdb_amd
    kmerdb_amd._abi.build()
    L = ctypes.CDLL(kmerdb_amd._abi.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "kdbhip.h")).read()
    declared = set(re.findall(r"\b(kdb_[a-z_]+)\s*\(", header))
    assert len(declared) >= 18
    bound = {name for name, _, _ in kmerdb_amd._abi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name), name
    assert kmerdb_amd._abi.lib().kdb_abi_version() == kmerdb_amd._abi.ABI_VERSION == 6


def test_no_device_fails_loudly():
    """Without a GPU the product path must raise, not fall back to any CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import kmerdb_amd
    with pytest.raises((kmerdb_amd._abi.KdbHipError, ValueError)):
        kmerdb_amd.Engine(8)
    with pytest.raises((kmerdb_amd._abi.KdbHipError, ValueError)):
        kmerdb_amd.parse.parsefile(os.path.join(ROOT, "tests", "golden", "inputs", "tiny.fq"), 5)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kmerdb_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "kmer_oracle" not in text, f


def test_parsefile_argument_errors():
    """parse.py:109-116 error ladder, checked before anything touches the device."""
    from kmerdb_amd import parse
    tiny = os.path.join(ROOT, "tests", "golden", "inputs", "tiny.fq")
    with pytest.raises(TypeError):
        parse.parsefile(None, 4)
    with pytest.raises(OSError):
        parse.parsefile("/does/not/exist.fa", 4)
    with pytest.raises(TypeError):
        parse.parsefile(tiny, "4")
    with pytest.raises(TypeError):
        parse.parsefile(tiny, 4, replace_with_none=1)


def test_kmer_host_utilities(golden_dir):
    import json
    from kmerdb_amd import kmer
    g = json.load(open(os.path.join(golden_dir, "kmer_to_id.json")))
    dinucs = [a + b for a in "ACGT" for b in "ACGT"]
    assert [kmer.kmer_to_id(s) for s in dinucs] == g["dinuc_canonical"]
    assert [kmer.kmer_to_id(s, canonicalize=False) for s in dinucs] == g["dinuc_forward"]
    assert kmer.kmer_to_id("ATCNATC") is None                                  # reference test_kmer.py:38-42
    for bad in (None, 1, 1.0, [1], {"hello": "world"}):                         # reference test_kmer.py:44-57
        with pytest.raises(TypeError):
            kmer.kmer_to_id(bad)
    for s, canon, fwd in g["random"]:
        assert kmer.kmer_to_id(s) == canon and kmer.kmer_to_id(s, canonicalize=False) == fwd
        assert kmer.id_to_kmer(fwd, len(s)) == s
    with pytest.raises(ValueError):
        kmer.kmer_to_id("ACGR")


def test_reader_matches_oracle_reader(oracle, golden_dir):
    from kmerdb_amd import reader
    for f in ("inputs/tiny.fq", "inputs/reads150.fq", "inputs/reads150.fq.gz", "inputs/ragged_n.fq", "inputs/contigs.fa",
              "ref_data/sample.fa", "ref_data/Cacetobutylicum_ATCC824.fasta.gz"):
        path = os.path.join(golden_dir, f)
        want = list(oracle.read_records(path))
        got_ids, got = [], []
        for bases, offsets, ids in reader.iter_blocks(path, want_ids=True, block_bytes=4096):
            o = offsets.astype(np.int64)
            got += [bytes(bases[o[i]:o[i + 1]]).decode() for i in range(len(o) - 1)]
            got_ids += ids
        assert got == [s for _, s in want], f
        assert got_ids == [i for i, _ in want], f
    with pytest.raises(ValueError):
        list(reader.iter_blocks(os.path.join(golden_dir, "parsefile.json")))


def test_reader_crlf_and_no_trailing_newline(tmp_path):
    from kmerdb_amd import reader
    p = tmp_path / "x.fq"
    p.write_bytes(b"@a x\r\nACGT\r\n+\r\nIIII\r\n@b\r\nGGCC\r\n+\r\nIIII")
    blocks = list(reader.iter_blocks(str(p), want_ids=True))
    assert b"".join(bytes(b) for b, _, _ in blocks) == b"ACGTGGCC"
    assert [int(x) for _, o, _ in blocks for x in np.diff(o.astype(np.int64))] == [4, 4]
    assert [i for _, _, ids in blocks for i in ids] == ["a", "b"]
    q = tmp_path / "y.fa"
    q.write_bytes(b">s1 d\r\nAC GT\r\nAC\r\n>s2\nTTTT")
    (bases, offsets, ids), = list(reader.iter_blocks(str(q), want_ids=True))
    assert bytes(bases) == b"ACGTACTTTT" and offsets.tolist() == [0, 6, 10] and ids == ["s1", "s2"]


def test_thread_counts_follow_the_cgroup_cpu_quota(tmp_path, monkeypatch):
    """util.effective_cpus sizes every host thread pool (the .kdb writer, the BGZF inflate, the record splitter): the CPUs the process may
    run on, cut down to its cgroup's quota -- the GPU pool's boxes show 256 CPUs and grant 16, and 64 deflate threads on 16 CPUs were
    2.2 x slower than 16.  cgroup v2 (cpu.max, the smallest quota along the path), v1 (cfs_quota_us / cfs_period_us), no quota, overrides."""
    from kmerdb_amd import util, fileutil
    root = tmp_path / "cg"
    (root / "a" / "b").mkdir(parents=True)
    proc = tmp_path / "cgroup"
    proc.write_text("0::/a/b\n")
    (root / "cpu.max").write_text("max 100000\n")
    (root / "a" / "cpu.max").write_text("1600000 100000\n")
    (root / "a" / "b" / "cpu.max").write_text("max 100000\n")
    assert util._cgroup_cpu_limit(str(proc), str(root)) == 16.0
    (root / "a" / "b" / "cpu.max").write_text("250000 100000\n")
    assert util._cgroup_cpu_limit(str(proc), str(root)) == 2.5
    (root / "a" / "cpu.max").write_text("max 100000\n")
    (root / "a" / "b" / "cpu.max").write_text("max 100000\n")
    assert util._cgroup_cpu_limit(str(proc), str(root)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu,cpuacct" / "job").mkdir(parents=True)
    (v1 / "cpu,cpuacct" / "job" / "cpu.cfs_quota_us").write_text("400000\n")
    (v1 / "cpu,cpuacct" / "job" / "cpu.cfs_period_us").write_text("100000\n")
    (v1 / "cpu,cpuacct" / "cpu.cfs_quota_us").write_text("-1\n")
    (v1 / "cpu,cpuacct" / "cpu.cfs_period_us").write_text("100000\n")
    proc.write_text("4:memory:/x\n3:cpu,cpuacct:/job\n")
    assert util._cgroup_cpu_limit(str(proc), str(v1)) == 4.0
    assert util._cgroup_cpu_limit(str(tmp_path / "missing"), str(v1)) is None
    monkeypatch.setattr(util, "_cgroup_cpu_limit", lambda: 3.2)
    monkeypatch.delenv("KDB_CPUS", raising=False)
    monkeypatch.delenv("KDB_WRITER_THREADS", raising=False)
    assert util.effective_cpus() == min(4, len(os.sched_getaffinity(0)))
    assert fileutil.default_writer_threads() == util.effective_cpus()
    monkeypatch.setenv("KDB_CPUS", "7")
    assert util.effective_cpus() == 7
    monkeypatch.setenv("KDB_WRITER_THREADS", "2")
    assert fileutil.default_writer_threads() == 2
