#!/bin/bash
# fuzz with per-piece submit kinds, syncs, resets, small arenas
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 400 python -u tests/fuzz_gpu.py 340 9047 > $O/fuzz_g.txt 2>&1; echo "fuzz g rc=$?"; tail -n 1 $O/fuzz_g.txt | cut -c1-500
timeout -k 10 400 python -u tests/fuzz_gpu.py 340 9048 14,15,16,17,13 > $O/fuzz_h.txt 2>&1; echo "fuzz h rc=$?"; tail -n 1 $O/fuzz_h.txt | cut -c1-500
timeout -k 10 260 python -u tests/fuzz_gpu.py 200 9049 9,10,11,12 > $O/fuzz_i.txt 2>&1; echo "fuzz i rc=$?"; tail -n 1 $O/fuzz_i.txt | cut -c1-500
