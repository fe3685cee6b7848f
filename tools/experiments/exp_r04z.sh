#!/bin/bash
# record offsets fetched two tiles ahead: A/B on the ragged batches (drop and expand), tests of ragged batches
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "rebatched or random_reads_vs_oracle or few_bases or tile_and_buffer or fuzz" > $O/t_z.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_z.txt
for r in 1 2; do for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 12 2>&1 | grep "p_N=0 \|p_N=0.0005 expand" | sed "s/^/$L /" | cut -c1-230
done; done
for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 15 2>&1 | grep "p_N=0 " | sed "s/^/$L /" | cut -c1-260
done
