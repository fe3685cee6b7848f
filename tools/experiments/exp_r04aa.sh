#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "ragged_reads_with_n_at_scale or without_a_countable_window or deferred_histogram" --durations=6 ) > $O/t_aa.txt 2>&1; echo "tests rc=$?"; tail -n 14 $O/t_aa.txt | cut -c1-300
timeout -k 10 460 python -u tests/fuzz_gpu.py 400 9045 > $O/fuzz_e.txt 2>&1; echo "fuzz e rc=$?"; tail -n 1 $O/fuzz_e.txt | cut -c1-400
timeout -k 10 300 python -u tests/fuzz_gpu.py 240 9046 14,15,16,17 > $O/fuzz_f.txt 2>&1; echo "fuzz f rc=$?"; tail -n 1 $O/fuzz_f.txt | cut -c1-400
