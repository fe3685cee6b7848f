#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
for L in libkdbhip.so libkdbhip_9a.so; do
  echo "== $L"; KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 400 python -u tools/experiments/repro_smallk.py 2>&1 | tail -n 30 | cut -c1-400
done
