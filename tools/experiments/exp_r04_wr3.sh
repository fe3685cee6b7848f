#!/bin/bash
# round 4, VERDICT item 7, third pass: page lines stored write-through (sc1) or non-temporal (nt) instead of plain -- time and bytes written
set -o pipefail
O=gpurun_out/r04/wr3; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
AB_LIBS="libkdbhip.so libkdbhip_SC1.so libkdbhip_NT.so" AB_STEPS=200 bash tools/ab_libs.sh --no-configs 2>&1 | tee $O/ab_k12.txt
AB_LIBS="libkdbhip.so libkdbhip_SC1.so libkdbhip_NT.so" AB_STEPS=64 bash tools/ab_libs.sh --no-configs --k 15 2>&1 | tee $O/ab_k15.txt
for L in libkdbhip.so libkdbhip_SC1.so libkdbhip_NT.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$L -- python3 bench.py --no-configs --no-cpu-baseline --no-extra-regions --steps 5 --warmup 1 > $O/w_$L.json 2> $O/w_$L.err || echo "$L failed"
  KDB_LIB=$PWD/kmerdb_amd/$L rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_NORMAL_WRITEBACK_sum --output-format csv -d $O/q_$L -- python3 bench.py --no-configs --no-cpu-baseline --no-extra-regions --steps 5 --warmup 1 > $O/q_$L.json 2> $O/q_$L.err || echo "$L failed"
done
for d in $O/w_* $O/q_*; do [ -d $d ] && python3 tools/pmc_table.py $d; done > $O/tables.md 2>&1
grep -h "scatter_bases\|^###\|^| kernel\|page_hist" $O/tables.md | cut -c1-200
