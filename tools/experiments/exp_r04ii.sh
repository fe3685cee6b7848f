#!/bin/bash
# rings_place: an element that does not exist adds 0 to its ring's word instead of sitting in an exec-mask region (-DKDB_RINGS_ADD0): A/B
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
for K in 12 15; do
  AB_STEPS=100 AB_LIBS="libkdbhip.so libkdbhip_add0.so" timeout -k 10 400 bash tools/ab_libs.sh --k $K --no-configs 2>&1 | tee $O/ab_ii_k$K.txt
done
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_add0.so timeout -k 10 300 python -u tests/fuzz_gpu.py 100 9060 12,13,15 > $O/fuzz_add0.txt 2>&1; echo "fuzz rc=$?"; tail -n 1 $O/fuzz_add0.txt | cut -c1-300
