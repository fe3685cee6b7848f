#!/bin/bash
# round 4, VERDICT item 7, second pass: which L2 events make up the write requests that leave the L2 beyond the lines the kernel stores
set -o pipefail
O=gpurun_out/r04/wr2; mkdir -p $O; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKDB_SC_PROF -I include -o kmerdb_amd/libkdbhip_prof.so kmerdb_amd/csrc/kdb_engine.hip -lz -lpthread || exit 1
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt; wc -l $O/tcc_counters.txt
run() {
  K=$1; AB=$2; shift 2
  T=$(echo "$*" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $* --output-format csv -d $O/k${K}_ab${AB}_$T -- python3 tools/sc_ablate.py $K $AB > $O/k${K}_ab${AB}_$T.log 2>&1 || echo "k=$K ab=$AB $* failed"
}
run 12 0 TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_INV_EVICT_sum
run 12 0 TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_WRITE_sum
run 12 0 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum
run 12 0 TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
for d in $O/k*_ab*; do [ -d $d ] && python3 tools/pmc_table.py $d; done > $O/tables.md 2>&1
grep -h "scatter_bases\|^###\|^| kernel\|page_hist" $O/tables.md | cut -c1-260
