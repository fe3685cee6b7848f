#!/bin/bash
# round 3, call C: the new defaults (lo 9 / 12, contiguous pages, arena 85 %) at k = 12 / 15 / 17 with the in-run roofline block
set -e
OUT=gpurun_out/r03c
mkdir -p $OUT
run() {  # tag, args...
  local tag=$1; shift
  python bench.py "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; return 1; }
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "roofline", r["kernel"], r["achieved"], r["frac"], "step", r["step"])
for k, v in r["per_kernel"].items():
    print("   ", k, v)
PY
}
run k12_default --steps 300 --warmup 5
run k15 --k 15 --steps 128 --warmup 3 --no-cpu-baseline --no-extra-regions
run k17 --k 17 --steps 128 --warmup 2 --no-cpu-baseline --no-extra-regions
