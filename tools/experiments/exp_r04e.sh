#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
for T in "n_dense and 8" "n_dense and 12" "n_dense and 13" "n_dense and 15" "all_n_reads" "const_device or bit_7 or rebatched or iupac or shielded"; do
  echo "== $T" >> $O/t5.txt
  timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -x -v -m gpu --timeout 200 -k "$T" >> $O/t5.txt 2>&1; echo "[$T] rc=$?"
done
grep -E "PASSED|FAILED|ERROR|Timeout|passed|failed" $O/t5.txt | tail -40
