#!/bin/bash
# N-windows dealt out over the whole workgroup at the top of the tile: tests + A/B on the ragged batches
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "n_dense or all_n_reads or golden_parsefile or fuzz or iupac or shielded or random_reads" > $O/t_p.txt 2>&1; echo "tests rc=$?"; tail -n 5 $O/t_p.txt
for L in libkdbhip_base.so libkdbhip_dpp.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 12 2>&1 | grep "expand" | sed "s/^/$L /"
done
for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 15 2>&1 | grep "expand" | sed "s/^/$L /"
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 13 2>&1 | grep "expand" | sed "s/^/$L /"
done
