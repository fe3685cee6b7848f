"""Minimal stand-in for the narrow Biopython surface that the reference's
kmerdb/kmer.py and kmerdb/parse.py touch (Seq, SeqRecord, SeqIO.parse).

Biopython (pinned ==1.83 by the reference's Pipfile.lock) is not installed in
the build container and there is no network, so tests/golden/make_golden.py
puts this package on sys.path to run the reference's own hot-path files
UNMODIFIED and record their outputs as golden vectors.  This is our code, not
the reference's; it is used only by make_golden.py, never by tests or product.
The only arithmetic Biopython contributes on the path is
Seq.reverse_complement() (kmer.py:310): complement A<->T, C<->G, then reverse.
"""
