#!/bin/bash
# round 3, final check: smoke, default bench line (with e2e formats), one-rank RCCL path (all three reduce shapes on a 1-rank group)
set -e
OUT=gpurun_out/r03i
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --steps 300 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo FAILED default; tail -5 $OUT/bench_default.err; }
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("k12", d["ms_per_step"], d["value"], d["gbase_per_s"], "roofline", r["kernel"], r["achieved"], r["frac"], "step", r["step"]["hbm_frac"], r["step"]["engine_over_pmc"])
e = d["timed_regions"]["fastq_e2e"]
print("e2e", {k: (v["ms"], v["gbase_per_s"], v["longest_stage"]) for k, v in e["formats"].items()}, "gz_over_plain", e["gz_over_plain"], "files4", e["files4_gbase_per_s"], "h2d", d["timed_regions"]["h2d_pinned"]["gbase_per_s"])
PY
for shape in ring rs_gather a2a_gather; do
KDB_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-regions --reduce-shape $shape > $OUT/nccl1_$shape.json 2> $OUT/nccl1_$shape.err || { echo FAILED nccl1 $shape; tail -5 $OUT/nccl1_$shape.err; continue; }
python -c "
import json; d=json.load(open('$OUT/nccl1_$shape.json')); print('nccl 1 rank', '$shape', d['ms_per_step'], d['reduce_ms'], d['reduce_shape'], d['reduce_probe']['ms'])"
done
