#!/bin/bash
# round 3, call F: full default bench line (k = 12) + rocprofv3 kernel stats / PMC passes for profiles/r03
set -e
OUT=gpurun_out/r03f
mkdir -p $OUT
python bench.py --steps 300 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo FAILED; tail -5 $OUT/bench_default.err; }
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("k12", d["ms_per_step"], d["value"], "roofline", r["kernel"], r["achieved"], r["frac"], r["step"])
print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
print("regions", json.dumps(d["timed_regions"])[:3000])
PY
tools/profile_gpu.sh r03_k12 2>&1 | tail -30
