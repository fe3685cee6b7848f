#!/bin/bash
# round 4, first contact of the new kernels: parity of the one-CU LDS histogram (k <= 8) and the 1024-ring kernel (k = 13),
# then the default bench line (with the configs block) and A/B lines against the paths they replace
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k8_lds or k17_bins" > $O/t_k8.txt 2>&1; echo "k8 test rc=$?"; tail -3 $O/t_k8.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "9001 or 777" > $O/t_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 $O/t_fuzz.txt
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config5 or config2" > $O/t_cfg.txt 2>&1; echo "configs rc=$?"; tail -3 $O/t_cfg.txt
for K in 8 13; do
  timeout -k 10 300 python bench.py --k $K --steps 100 --no-cpu-baseline --no-extra-regions > $O/b_k$K.json 2> $O/b_k$K.err; echo "bench k=$K rc=$?"
done
timeout -k 10 300 python bench.py --k 8 --steps 100 --no-cpu-baseline --no-extra-regions --opt smallk_old=1 > $O/b_k8_old.json 2> $O/b_k8_old.err; echo "bench k=8 old rc=$?"
timeout -k 10 300 python bench.py --k 13 --steps 100 --no-cpu-baseline --no-extra-regions --opt one_level_max_k=12 > $O/b_k13_old.json 2> $O/b_k13_old.err; echo "bench k=13 old rc=$?"
for K in 5 7; do
  timeout -k 10 300 python bench.py --k $K --steps 50 --no-cpu-baseline --no-extra-regions > $O/b_k$K.json 2> $O/b_k$K.err; echo "bench k=$K rc=$?"
  timeout -k 10 300 python bench.py --k $K --steps 50 --no-cpu-baseline --no-extra-regions --opt smallk_old=1 > $O/b_k${K}_old.json 2> $O/b_k${K}_old.err; echo "bench k=$K old rc=$?"
done
timeout -k 10 900 python bench.py > $O/bench_b.json 2> $O/bench_b.err; echo "default bench rc=$?"; tail -c 400 $O/bench_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/b_k*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['roofline']['kernels_ms_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
