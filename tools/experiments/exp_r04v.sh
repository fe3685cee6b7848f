#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scratch_that_does_not_fit or shielded" > $O/t_v.txt 2>&1; echo "tests rc=$?"; tail -n 25 $O/t_v.txt
