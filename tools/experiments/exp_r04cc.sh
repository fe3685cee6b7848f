#!/bin/bash
# more fuzz on the final build (new seeds)
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 460 python -u tests/fuzz_gpu.py 400 9050 > $O/fuzz_j.txt 2>&1; echo "fuzz j rc=$?"; tail -n 1 $O/fuzz_j.txt | cut -c1-500
timeout -k 10 400 python -u tests/fuzz_gpu.py 340 9051 13,14,15,16,17 > $O/fuzz_k.txt 2>&1; echo "fuzz k rc=$?"; tail -n 1 $O/fuzz_k.txt | cut -c1-500
timeout -k 10 300 python -u tests/fuzz_gpu.py 240 9052 1,2,3,4,5,6,7,8,9 > $O/fuzz_l.txt 2>&1; echo "fuzz l rc=$?"; tail -n 1 $O/fuzz_l.txt | cut -c1-500
