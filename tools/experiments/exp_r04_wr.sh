#!/bin/bash
# round 4, VERDICT item 7: where do the 17-21 % of write traffic beyond the engine's own count go?  PMC write counters of
# scatter_bases_kernel with parts of it switched off (diagnostic build -DKDB_SC_PROF; counts are meaningless in these modes).
set -o pipefail
O=gpurun_out/r04/wr; mkdir -p $O; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKDB_SC_PROF -I include -o kmerdb_amd/libkdbhip_prof.so kmerdb_amd/csrc/kdb_engine.hip -lz -lpthread || exit 1
for K in 12 15; do
  for AB in 0 4 8 12 1; do
    for C in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum"; do
      T=$(echo $C | tr ' ' '_' | cut -c1-24)
      rocprofv3 --pmc $C --output-format csv -d $O/k${K}_ab${AB}_$T -- python3 tools/sc_ablate.py $K $AB > $O/k${K}_ab${AB}_$T.log 2>&1 || echo "k=$K ab=$AB $C failed"
    done
    echo "k=$K ablate=$AB done"
  done
done
for d in $O/k*_ab*/; do python3 tools/pmc_table.py $d; done > $O/tables.md 2>&1
grep -h "scatter_bases\|^###" $O/tables.md | cut -c1-200
