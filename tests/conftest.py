import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure). Built on demand with gcc."""
    from oracle import kmer_oracle
    kmer_oracle.build()
    return kmer_oracle


@pytest.fixture(scope="session")
def gpu_engine_cls():
    """kmerdb_amd.Engine, after checking that the HIP library is really there (no fallback)."""
    import kmerdb_amd
    kmerdb_amd._abi.lib()
    assert kmerdb_amd.device_count() >= 1, "no HIP device visible"
    return kmerdb_amd.Engine


@pytest.fixture(scope="module", autouse=True)
def _release_pooled_engines():
    """parse.parsefile keeps an engine per parameter set between calls (its HBM with it); a test module must not leave them to the
    next one -- the config-4 tests want two 128 GiB vectors beside whatever else the process holds."""
    yield
    try:
        from kmerdb_amd import parse
        parse.release_engines()
    except Exception:
        pass
