#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "n_dense or all_n_reads or fuzz or golden_parsefile or random_reads" > $O/t9.txt 2>&1; echo "tests rc=$?"; tail -4 $O/t9.txt
timeout -k 10 600 python -u bench.py --no-configs --no-cpu-baseline > $O/bench_g.json 2> $O/bench_g.err; echo "bench rc=$?"; tail -c 300 $O/bench_g.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04/bench_g.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
print(json.dumps(j['timed_regions']['resident_other_modes']))
r=j['timed_regions']['resident_ragged_n']
print({k:v for k,v in r.items() if k!='what'})
PY
