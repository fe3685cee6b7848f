"""The randomised differential run (tests/fuzz_gpu.py) under the driver: seeded cases over k = 1..17, both strand and N
modes, every algo, engine options (sc_grid, sc_lo_bits, sc_contig_pages, defer_flush, staging sizes, chunk
accumulation) and chunked submits, each compared with the oracle (kmer.py:234-317, :489-577; parse.py:133-136)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gpu  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,ncases,only_k", [
    (20240612, 64, None),                         # the whole k mix
    (777, 18, [12, 12, 13, 15, 17]),              # the scatter paths, one and two levels
    (4242, 12, [14, 15, 16, 17]),                 # two levels (EXPAND included), sparse compare (k = 17: a 128 GiB vector per case)
    (9001, 60, [1, 2, 3, 5, 7, 8, 8, 8, 13, 13, 13]),   # the one-CU LDS histogram (k <= 8) and the 1024-ring one-level kernel (k = 13)
])
def test_seeded_fuzz_cases_equal_the_oracle(gpu_engine_cls, oracle, seed, ncases, only_k):
    n, bad = fuzz_gpu.run_cases(seed, max_cases=ncases, only_k=only_k, verbose=False)
    assert bad is None, f"case {n} of seed {seed} differs from the oracle: {bad}"
    assert n == ncases
