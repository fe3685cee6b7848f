#!/bin/bash
# round 4, VERDICT item 7: where do the 17-21 % of write traffic beyond the engine's own count go?  PMC write counters of
# scatter_bases_kernel with parts of it switched off (diagnostic build -DKDB_SC_PROF; counts are meaningless in these modes).
# ablate bits: 1 no line stores of the flush, 4 no page tags, 8 no drain at the end of the kernel
set -o pipefail
O=gpurun_out/r04/wr; mkdir -p $O; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKDB_SC_PROF -I include -o kmerdb_amd/libkdbhip_prof.so kmerdb_amd/csrc/kdb_engine.hip -lz -lpthread || exit 1
run() {  # k ablate counters...
  K=$1; AB=$2; shift 2
  T=$(echo "$*" | tr ' ' '_' | cut -c1-28)
  rocprofv3 --pmc $* --output-format csv -d $O/k${K}_ab${AB}_$T -- python3 tools/sc_ablate.py $K $AB > $O/k${K}_ab${AB}_$T.log 2>&1 || echo "k=$K ab=$AB $* failed"
  echo "k=$K ablate=$AB [$*] done"
}
for AB in 0 4 8 1; do
  run 12 $AB WRITE_SIZE
  run 12 $AB TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
done
run 12 0 TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum
for AB in 0 4; do run 15 $AB WRITE_SIZE; done
for d in $O/k*_ab*/; do python3 tools/pmc_table.py $d; done > $O/tables.md 2>&1
grep -h "scatter_bases\|^###\|^| kernel" $O/tables.md | cut -c1-220
