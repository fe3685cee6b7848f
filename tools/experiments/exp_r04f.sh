#!/bin/bash
# the whole -m gpu suite (unbuffered, so that progress is visible), then the default bench line
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --durations=15 ) > $O/t6_all.txt 2>&1; echo "gpu suite rc=$?"; tail -25 $O/t6_all.txt
timeout -k 10 700 python -u bench.py > $O/bench_d.json 2> $O/bench_d.err; echo "bench rc=$?"; tail -c 300 $O/bench_d.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04/bench_d.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
print(json.dumps(j['timed_regions']['resident_other_modes']))
r=j['timed_regions']['resident_ragged_n']
print({k:v for k,v in r.items() if k!='what'})
for k,v in j['configs'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), {n:x['ms_per_step'] for n,x in v.get('per_kernel',{}).items()}, v.get('batches_per_flush'))
PY
