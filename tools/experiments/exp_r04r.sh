#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 800 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu --durations=8 -k "config or golden_k13 or n_dense or all_n_reads or iupac" ) > $O/t_r.txt 2>&1; echo "tests rc=$?"; tail -n 16 $O/t_r.txt
for L in libkdbhip_dpp.so libkdbhip.so libkdbhip_dpp.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 12 2>&1 | grep "expand" | sed "s/^/$L /" | cut -c1-200
done
