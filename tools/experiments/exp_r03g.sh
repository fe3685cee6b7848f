#!/bin/bash
# round 3, call G: the numbers DESIGN.md section 5 quotes -- default line, every k, skewed inputs
set -e
OUT=gpurun_out/r03g
mkdir -p $OUT
python bench.py --steps 300 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo FAILED default; tail -5 $OUT/bench_default.err; }
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("k12", d["ms_per_step"], d["value"], d["gbase_per_s"], "roofline", r["kernel"], r["achieved"], r["frac"])
print("regions", json.dumps(d["timed_regions"])[:3500])
PY
for k in 8 9 10 11 13 14 15 16 17; do
  steps=96; [ $k -ge 15 ] && steps=192; [ $k -eq 17 ] && steps=256
  python bench.py --k $k --steps $steps --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/k$k.json 2> $OUT/k$k.err || { echo FAILED k$k; tail -3 $OUT/k$k.err; continue; }
  python -c "
import json; d=json.load(open('$OUT/k$k.json')); r=d['roofline']; print('k$k', d['ms_per_step'], d['gbase_per_s'], {n: round(v,3) for n,v in r['kernels_ms_per_step'].items()}, {n: v.get('hbm_frac') for n,v in r['per_kernel'].items()}, r.get('arena'), flush=True)"
done
python tools/bench_skew.py > $OUT/skew.json 2> $OUT/skew.err || { echo FAILED skew; tail -3 $OUT/skew.err; }
cat $OUT/skew.json | head -c 3000
