"""-m gpu: a plain C program drives the engine through include/kdbhip.h (no Python in the loop)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c_binary(tmp_path_factory):
    import kmerdb_amd
    kmerdb_amd._abi.build()
    out = str(tmp_path_factory.mktemp("cabi") / "abi_smoke")
    libdir = os.path.dirname(kmerdb_amd._abi.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", libdir, "-lkdbhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_c_program_counts_match_oracle(c_binary, oracle):
    recs = ["ACGTACGTTTGACCANNACGTAGCTAGCTAGGATCCA", "GATTACAGATTACAGATTACA", "TTTTTTTTTTTTTTTT", "ACGTACGTAC"]
    for k, canon in ((5, 1), (9, 0), (10, 1)):
        p = subprocess.run([c_binary, str(k), str(canon)] + recs, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        lines = p.stdout.strip().split("\n")
        bases, offsets = oracle.pack_records(recs)
        want, want_total = oracle.c_count(bases, offsets, k, bool(canon), oracle.N_DROP)
        assert lines[0] == f"total {want_total} unique {int(np.count_nonzero(want))}"
        got = {int(a): int(b) for a, b in (ln.split() for ln in lines[1:])}
        assert got == {int(i): int(want[i]) for i in np.flatnonzero(want)}


def test_c_program_reduces_several_engines(c_binary, oracle):
    """kdb_reduce from plain C: records dealt out over 3 engines (device j mod the device count), vectors summed into engine 0."""
    recs = ["ACGTACGTTTGACCANNACGTAGCTAGCTAGGATCCA", "GATTACAGATTACAGATTACA", "TTTTTTTTTTTTTTTT", "ACGTACGTAC", "CCCCCCCCCCGGGGGGGGGGAT",
            "ATATATATATATATATATATAT", "GGGCCCAAATTTGGGCCCAAATTT"]
    env = dict(os.environ, KDB_SMOKE_ENGINES="3")
    for k, canon in ((5, 1), (10, 0)):
        p = subprocess.run([c_binary, str(k), str(canon)] + recs, capture_output=True, text=True, timeout=120, env=env)
        assert p.returncode == 0, p.stderr
        lines = p.stdout.strip().split("\n")
        bases, offsets = oracle.pack_records(recs)
        want, want_total = oracle.c_count(bases, offsets, k, bool(canon), oracle.N_DROP)
        assert lines[0] == f"total {want_total} unique {int(np.count_nonzero(want))}"
        got = {int(a): int(b) for a, b in (ln.split() for ln in lines[1:])}
        assert got == {int(i): int(want[i]) for i in np.flatnonzero(want)}


def test_c_program_reports_errors(c_binary):
    p = subprocess.run([c_binary, "8", "1", "ACGTACGTACGT", "ACG"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 10 + 3 and "shorter than k" in p.stderr          # KDB_ERR_SHORT_READ
    p = subprocess.run([c_binary, "4", "1", "ACGTRACGT"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 10 + 4 and "outside ACGTN" in p.stderr           # KDB_ERR_BAD_RESIDUE
    p = subprocess.run([c_binary, "18", "1", "ACGTACGTACGTACGTACGT"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "outside 1..17" in p.stderr                 # KDB_ERR_ARG from kdb_create
