#!/bin/bash
# round 3, experiment B (GPU box): k = 17 histogram pass vs sc_lo_bits (+ UTCL1 translation counters), and the two speeds
# of level 1 at k = 15 (page numbering interleaved vs contiguous, alternating runs).   -> gpurun_out/r03b/
set -e
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra-regions"
run() {  # tag, args...
  local tag=$1; shift
  python bench.py $COMMON "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; return 1; }
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[2], "ms/step", d["ms_per_step"], {k: round(v, 3) for k, v in r["kernels_ms_per_step"].items()},
      "avg", {k: round(v, 3) for k, v in r["kernels_avg_ms"].items() if "hist" in k}, flush=True)
PY
}
pmc() {  # tag, counters, args...
  local tag=$1 ctrs=$2; shift; shift
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$tag -- python3 bench.py $COMMON "$@" > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $OUT/pmc_$tag.err; }
  python3 tools/pmc_table.py $OUT/pmc_$tag > $OUT/pmc_$tag.md && cat $OUT/pmc_$tag.md
  rm -rf $OUT/pmc_$tag
}
if [ "${1:-all}" != tlb ]; then
for lo in 6 9 12 14; do run k17_lo$lo --k 17 --steps 96 --warmup 2 --opt sc_lo_bits=$lo; done
for i in 1 2 3; do for c in 0 1; do run k15_contig${c}_run$i --k 15 --steps 64 --warmup 3 --opt sc_contig_pages=$c; done; done
fi
TLB="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
for lo in 6 12; do pmc tlb_k17_lo$lo "$TLB" --k 17 --steps 30 --warmup 1 --opt sc_lo_bits=$lo; done
for c in 0 1; do pmc tlb_k15_contig$c "$TLB" --k 15 --steps 8 --warmup 1 --opt sc_contig_pages=$c; done
