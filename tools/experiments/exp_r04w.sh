#!/bin/bash
# builder-run fuzz on the final build: new seeds, every k; then EXPAND-heavy ks
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 420 python -u tests/fuzz_gpu.py 360 9041 > $O/fuzz_a.txt 2>&1; echo "fuzz a rc=$?"; tail -n 2 $O/fuzz_a.txt | cut -c1-300
timeout -k 10 300 python -u tests/fuzz_gpu.py 240 9042 9,10,11,12,13 > $O/fuzz_b.txt 2>&1; echo "fuzz b rc=$?"; tail -n 2 $O/fuzz_b.txt | cut -c1-300
timeout -k 10 200 python -u tests/fuzz_gpu.py 150 9043 8,13,17 > $O/fuzz_c.txt 2>&1; echo "fuzz c rc=$?"; tail -n 2 $O/fuzz_c.txt | cut -c1-300
grep -c "^ok" $O/fuzz_a.txt $O/fuzz_b.txt $O/fuzz_c.txt; grep "^FAIL" $O/fuzz_*.txt | head
