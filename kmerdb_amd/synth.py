"""Synthetic workloads of BASELINE.md section 4: 150-bp reads, i.i.d. uniform ACGT, no N."""
import numpy as np

SEED0 = 20240612
_LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)


def reads(n_reads, read_len=150, seed=SEED0, p_n=0.0):
    """-> (bases uint8[n_reads*read_len], offsets uint64[n_reads+1]); reproducible (PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    total = n_reads * read_len
    bases = np.empty(total, dtype=np.uint8)
    step = 1 << 26
    for s in range(0, total, step):
        e = min(total, s + step)
        bases[s:e] = _LETTERS[rng.integers(0, 4, size=e - s, dtype=np.uint8)]
        if p_n > 0:
            bases[s:e][rng.random(e - s) < p_n] = ord("N")
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return bases, offsets


def fastq_text(bases, offsets, qual=b"I"):
    """FASTQ framing of BASELINE.md: @r{i}\\n{seq}\\n+\\n{'I'*len}\\n."""
    out = []
    o = offsets.astype(np.int64)
    for i in range(len(o) - 1):
        s = bytes(bases[o[i]:o[i + 1]])
        out.append(b"@r%d\n%s\n+\n%s\n" % (i, s, qual * len(s)))
    return b"".join(out)
