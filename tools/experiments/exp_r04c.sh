#!/bin/bash
# round 4: N expansion through the rings / LDS histogram, IUPAC codes shielded by N, suspects list -- whole parity + fuzz suites, then the default bench
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_c_abi_gpu.py tests/test_graph.py -x -q -m gpu > $O/t3_all.txt 2>&1; echo "parity+fuzz rc=$?"; tail -5 $O/t3_all.txt
timeout -k 10 600 python bench.py --no-configs > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc=$?"; tail -c 300 $O/bench_c.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04/bench_c.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
print(json.dumps(j['timed_regions']['resident_other_modes']))
r=j['timed_regions']['resident_ragged_n']
print({k:v for k,v in r.items() if k!='what'})
PY
