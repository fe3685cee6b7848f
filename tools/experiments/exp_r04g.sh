#!/bin/bash
# round 4: one N round per tile, record starts from the offsets, deeper flight in the histogram flush -- tests, A/B of the headline against round 3's build, bench
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "n_dense or all_n_reads or fuzz or golden_parsefile or random_reads or iupac or shielded or k17_bins or k8_lds or deferred" > $O/t7.txt 2>&1; echo "tests rc=$?"; tail -4 $O/t7.txt
AB_LIBS="libkdbhip_r3.so libkdbhip_premf.so libkdbhip.so" AB_STEPS=200 bash tools/ab_libs.sh --no-configs 2>&1 | tee $O/ab_k12.txt
AB_LIBS="libkdbhip_premf.so libkdbhip.so" AB_STEPS=64 bash tools/ab_libs.sh --no-configs --k 17 2>&1 | tee $O/ab_k17.txt
AB_LIBS="libkdbhip_premf.so libkdbhip.so" AB_STEPS=64 bash tools/ab_libs.sh --no-configs --k 15 2>&1 | tee $O/ab_k15.txt
AB_LIBS="libkdbhip_premf.so libkdbhip.so" AB_STEPS=100 bash tools/ab_libs.sh --no-configs --k 13 2>&1 | tee $O/ab_k13.txt
timeout -k 10 600 python -u bench.py --no-configs > $O/bench_e.json 2> $O/bench_e.err; echo "bench rc=$?"; tail -c 300 $O/bench_e.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04/bench_e.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
print(json.dumps(j['timed_regions']['resident_other_modes']))
r=j['timed_regions']['resident_ragged_n']
print({k:v for k,v in r.items() if k!='what'})
PY
