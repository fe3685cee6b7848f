#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "n_dense or all_n_reads or fuzz or golden_parsefile or accumulates or random_reads" > $O/t4.txt 2>&1; echo "tests rc=$?"; tail -4 $O/t4.txt
timeout -k 10 600 python bench.py --no-configs > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc=$?"; tail -c 300 $O/bench_c.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04/bench_c.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
print(json.dumps(j['timed_regions']['resident_other_modes']))
r=j['timed_regions']['resident_ragged_n']
print({k:v for k,v in r.items() if k!='what'})
PY
