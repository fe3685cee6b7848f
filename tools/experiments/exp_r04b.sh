#!/bin/bash
# round 4: batched returning atomics in the 16-bit histogram pass, cheaper wrap test at k = 8, arena charged by the device's cursor
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k8_lds or k17_bins or arena_is_charged or deferred_histogram or accumulated_submits" > $O/t2_par.txt 2>&1; echo "parity subset rc=$?"; tail -3 $O/t2_par.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/t2_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 $O/t2_fuzz.txt
for K in 8 13 15 17; do
  timeout -k 10 400 python bench.py --k $K --steps 128 --no-cpu-baseline --no-extra-regions > $O/b2_k$K.json 2> $O/b2_k$K.err; echo "bench k=$K rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/b2_k*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['roofline']['kernels_ms_per_step'], j['roofline'].get('arena'))
    except Exception as e:
        print(f, 'ERR', e)
PY
