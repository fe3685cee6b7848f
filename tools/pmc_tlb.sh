#!/bin/bash
# UTCL1 (per-CU TLB) request / hit / miss counters per kernel for the bench command; separate passes (one counter each)
OUT=gpurun_out/pmc_tlb; mkdir -p $OUT; export TMPDIR=/tmp
for C in TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_$C.json 2> $OUT/$C.err || echo "pmc $C failed"
done
python3 - <<'PY'
import csv,glob,collections
for C in ("TCP_UTCL1_REQUEST_sum","TCP_UTCL1_TRANSLATION_MISS_sum","TCP_UTCL1_TRANSLATION_HIT_sum"):
    f=glob.glob(f"gpurun_out/pmc_tlb/{C}/*/*counter_collection.csv")
    if not f: print(C,"no csv"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Kernel_Name"].startswith(("kdb::","void kdb::")): acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    print("##",C)
    for k,v in acc.items(): print(f"{k:62s} n={len(v)} mean={sum(v)/len(v):.0f}")
PY
