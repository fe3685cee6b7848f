#!/bin/bash
# final build of the round: default bench line, every k, rocprofv3 + PMC passes of k = 12, 13, 8, 15, 17
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 800 python -u bench.py ) > $O/bench_final2.json 2> $O/bench_final2.err; echo "bench rc=$?"; tail -n 4 $O/bench_final2.err
for K in 8 9 10 11 13 14 15 16 17; do
  timeout -k 10 300 python -u bench.py --k $K --steps 128 --no-cpu-baseline --no-extra-regions --no-configs > $O/bk2_$K.json 2> $O/bk2_$K.err; echo "k=$K rc=$?"
done
bash tools/experiments/exp_r04_prof.sh 12 13 8 15 17 > $O/prof_final2.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bk2_*.json'), key=lambda x:int(x.split('_')[-1].split('.')[0])):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        a=j['roofline'].get('arena') or {}
        print(f.split('/')[-1], j['ms_per_step'], j['gbase_per_s'], j['roofline']['kernels_ms_per_step'], a.get('batches_per_flush'), {k:v.get('hbm_frac') for k,v in j['roofline']['per_kernel'].items() if 'hbm_frac' in v})
    except Exception as e: print(f,'ERR',e)
PY
