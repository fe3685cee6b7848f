#!/usr/bin/env python3
"""Golden vectors for records whose IUPAC codes lie next to N's: what the REFERENCE's own
kmer.shred / parse.parsefile (loaded unmodified from /root/reference, as in make_golden.py) return or raise.

    python tests/golden/make_golden_iupac.py        (build container only: needs /root/reference)

kmer_to_id returns None for a window that holds an 'N' before it can meet another IUPAC code (kmer.py:287-289), so a
record is accepted when every window that holds such a code also holds an N: dropped with replace_with_none=True
(kmer.py:541-544), passed through _substitute_na_doublets / _triplets with replace_with_none=False (kmer.py:545-565,
:630-851) -- whose own quirks (a code that occurs once in a window is left in place: kmer.py:612 replaces "N", not the
code; two different codes in one window: NameError from the reversed comprehension clauses, kmer.py:652 ff.) decide
whether counts or an exception come out.  Output: iupac_next_to_n.json (data only).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402

RECORDS = [
    "ACGTNRNACGT",        # the judge's example: R between two N's
    "ACGTNNRNNACGT",      # k = 4 still reaches no N-free window over the R
    "ACGTRNACGT",         # R has an N on one side only: a window without N holds it
    "ACGTNRRNACGT",       # the same code twice in every window (k >= 3 windows hold both R's or an N)
    "NRNACGT",            # code at the start of the record
    "ACGTNYN",            # ... and at the end
    "ACGTNBNACGT",        # a triplet code
    "ACGTNBBNACGT",
    "ACGTNRYNACGT",       # two different doublet codes in one window
    "ACGTNRBNACGT",       # a doublet and a triplet code
    "NNRRNNACGTNNBBNN",
    "ACGTNRNRNACGT",
    "RNACGTACGT",         # first window holds R and N, the second one only ... N? no: k=3 'RNA','NAC' both hold N
    "ACGTACGTNR",
    "ACGTSACGT",          # no N at all: every mode raises
    "ANRRNA",             # k = 4: every window holds the code twice -- the one shape the expansion gets through
]


def main():
    ref = load_reference()
    kmer, parse = ref["kmer"], ref["parse"]
    from Bio.SeqRecord import SeqRecord
    from Bio.Seq import Seq
    ind = os.path.join(HERE, "inputs")
    out = []
    for s in RECORDS:
        for k in (3, 4, 5):
            if len(s) < k:
                continue
            for rwn in (True, False):
                for canon in (True, False):
                    case = {"seq": s, "k": k, "replace_with_none": rwn, "canonicalize": canon}
                    try:
                        ids, _, pos = kmer.shred(SeqRecord(Seq(s), id="s"), k, replace_with_none=rwn, canonicalize=canon)
                        counts = np.zeros(4 ** k, dtype=np.uint64)
                        for i in ids:
                            counts[i] += 1
                        case.update({"raises": None, "ids": [int(x) for x in ids], "pos": [int(p) for p in pos], "counts": [int(c) for c in counts]})
                    except BaseException as e:  # noqa: BLE001 - whatever class the reference raises is the datum
                        case.update({"raises": type(e).__name__})
                    out.append(case)
    # through parse.parsefile: a FASTA of accepted records, and one with a record the reference refuses
    files = {"iupac_n_ok.fa": [">a\nACGTNRNACGT\n", ">b\nACGTNNYNNACGTACGT\n", ">c\nACGTACGTACGTNBN\n"],
             "iupac_n_bad.fa": [">a\nACGTNRNACGT\n", ">b\nACGTRNACGT\n"]}
    pf = []
    for name, recs in files.items():
        with open(os.path.join(ind, name), "w") as f:
            f.write("".join(recs))
        for k in (3, 4):
            for rwn in (True, False):
                for canon in (True, False):
                    case = {"file": "inputs/" + name, "k": k, "replace_with_none": rwn, "canonicalize": canon}
                    cwd = os.getcwd()
                    os.chdir(HERE)
                    try:
                        counts, meta, _ = parse.parsefile("inputs/" + name, k, replace_with_none=rwn, canonicalize=canon)
                        case.update({"raises": None, "counts": [int(c) for c in counts],
                                     "metadata": {kk: (int(v) if isinstance(v, (int, np.integer)) else v) for kk, v in meta.items()}})
                    except BaseException as e:  # noqa: BLE001
                        case.update({"raises": type(e).__name__})
                    finally:
                        os.chdir(cwd)
                    pf.append(case)
    json.dump({"shred": out, "parsefile": pf}, open(os.path.join(HERE, "iupac_next_to_n.json"), "w"))
    ok = sum(1 for c in out if c["raises"] is None)
    print(f"{len(out)} shred cases ({ok} return, {len(out) - ok} raise), {len(pf)} parsefile cases "
          f"({sum(1 for c in pf if c['raises'] is None)} return)")
    for c in out:
        if c["canonicalize"] and c["k"] in (3, 4):
            print(c["seq"], c["k"], "drop" if c["replace_with_none"] else "expand", c["raises"] or f"{len(c['ids'])} ids")


if __name__ == "__main__":
    main()
