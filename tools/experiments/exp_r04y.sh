#!/bin/bash
# after the starts_apply fix: regression test, the reproduction script, the fuzz seeds again and new ones
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reads_of_a_few_bases" > $O/t_y.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_y.txt
timeout -k 10 300 python -u tools/experiments/repro_smallk.py 2>&1 | tail -n 3
timeout -k 10 460 python -u tests/fuzz_gpu.py 400 9041 > $O/fuzz_a2.txt 2>&1; echo "fuzz a rc=$?"; tail -n 1 $O/fuzz_a2.txt | cut -c1-300
timeout -k 10 300 python -u tests/fuzz_gpu.py 240 9044 1,2,3,4,5,6,7,8 > $O/fuzz_d.txt 2>&1; echo "fuzz d rc=$?"; tail -n 1 $O/fuzz_d.txt | cut -c1-300
grep -c "^ok" $O/fuzz_a2.txt $O/fuzz_d.txt; grep "^FAIL" $O/fuzz_a2.txt $O/fuzz_d.txt | head
