#!/bin/bash
# N fills ask the rings for slots in the tile's own request phase (no placement round of their own): A/B on the ragged batches + the N tests
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "n_dense or all_n_reads or golden_parsefile or fuzz or iupac or shielded" > $O/t_n.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_n.txt
for r in 1 2; do for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 12 2>&1 | sed "s/^/$L /"
done; done
for L in libkdbhip_base.so libkdbhip.so; do
  KDB_LIB=$PWD/kmerdb_amd/$L timeout -k 10 300 python -u tools/experiments/ragged_ab.py 15 2>&1 | sed "s/^/$L /"
done
