#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 440 python -u tests/fuzz_gpu.py 400 9070 > $O/fuzz_p.txt 2>&1; echo "fuzz p rc=$?"; tail -n 1 $O/fuzz_p.txt | cut -c1-300
timeout -k 10 400 python -u tests/fuzz_gpu.py 360 9071 14,15,16,17 > $O/fuzz_q.txt 2>&1; echo "fuzz q rc=$?"; tail -n 1 $O/fuzz_q.txt | cut -c1-300
