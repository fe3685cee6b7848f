#!/bin/bash
# round 3: full GPU suite on the build with paired high-byte lines, two rounds at k = 17 and the arena growth rule; all k
set -e
OUT=gpurun_out/r03n
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_graph.py tests/test_kdb_format.py tests/test_reader_cpu.py tests/test_oracle_golden.py tests/test_host_sanitize.py -m gpu -q -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for k in 13 14 15 16 17; do
  steps=96; [ $k = 17 ] && steps=128
  python bench.py --k $k --steps $steps --warmup 3 --no-cpu-baseline --no-extra-regions > $OUT/bench_k$k.json 2> $OUT/bench_k$k.err || { echo FAILED k=$k; tail -5 $OUT/bench_k$k.err; continue; }
  python -c "
import json; d=json.load(open('$OUT/bench_k$k.json')); r=d['roofline']; print('k=$k', d['ms_per_step'], d['gbase_per_s'], {k: round(v,3) for k, v in r['kernels_ms_per_step'].items()}, r.get('arena'), {n: round(v['hbm_frac'],3) for n, v in r['per_kernel'].items() if 'hbm_frac' in v}, flush=True)"
done
