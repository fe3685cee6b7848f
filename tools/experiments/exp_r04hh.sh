#!/bin/bash
# final build: default bench line, k = 8 and k = 13 again, rocprofv3 + PMC passes of k = 8 and k = 13
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 800 python -u bench.py ) > $O/bench_final3.json 2> $O/bench_final3.err; echo "bench rc=$?"; tail -n 4 $O/bench_final3.err
for K in 8 13; do
  timeout -k 10 300 python -u bench.py --k $K --steps 128 --no-cpu-baseline --no-extra-regions --no-configs > $O/bk3_$K.json 2> $O/bk3_$K.err; echo "k=$K rc=$?"
done
bash tools/experiments/exp_r04_prof.sh 8 13 > $O/prof_final3.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bk3_*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['gbase_per_s'], j['roofline']['kernels_ms_per_step'])
j=json.loads(open('gpurun_out/r04/bench_final3.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['timed_regions']['resident_ragged_n']['ms_per_gbase_over_uniform'])
PY
