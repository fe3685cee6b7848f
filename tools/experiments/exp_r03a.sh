#!/bin/bash
# round 3, experiment A (GPU box): where the bucket field sits in the id (sc_lo_bits) and how level-1 pages are numbered
# (sc_contig_pages), per-kernel ms at k = 12 / 15 / 17.   -> gpurun_out/r03a/
set -e
OUT=gpurun_out/r03a
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-extra-regions"
run() {  # tag, args...
  local tag=$1; shift
  python bench.py $COMMON "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; return 1; }
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step", d["ms_per_step"], {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_step"].items()}, flush=True)
PY
}
PART=${1:-all}
if [ $PART != k17 ]; then
for lo in 6 9 12 14; do run k12_lo$lo --k 12 --steps 100 --warmup 3 --opt sc_lo_bits=$lo; done
for lo in 6 9 12 14; do run k15_lo$lo --k 15 --steps 96 --warmup 3 --opt sc_lo_bits=$lo; done
for lo in 6 12; do for c in 0 1; do run k15_lo${lo}_contig$c --k 15 --steps 96 --warmup 3 --opt sc_lo_bits=$lo --opt sc_contig_pages=$c; done; done
for c in 0 1; do run k12_lo12_contig$c --k 12 --steps 100 --warmup 3 --opt sc_lo_bits=12 --opt sc_contig_pages=$c; done
fi
if [ $PART != small ]; then
for lo in 6 9 12 14; do run k17_lo$lo --k 17 --steps 96 --warmup 2 --opt sc_lo_bits=$lo; done
fi
