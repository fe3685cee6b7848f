#!/bin/bash
# last call of the round: the whole -m gpu suite on the final tree, then fuzz with new seeds
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=8 ) > $O/t_all_final2.txt 2>&1; echo "tests rc=$?"; tail -n 16 $O/t_all_final2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python -u tests/fuzz_gpu.py 240 9061 > $O/fuzz_n.txt 2>&1; echo "fuzz n rc=$?"; tail -n 1 $O/fuzz_n.txt | cut -c1-300
timeout -k 10 240 python -u tests/fuzz_gpu.py 180 9062 8,12,13,15 > $O/fuzz_o.txt 2>&1; echo "fuzz o rc=$?"; tail -n 1 $O/fuzz_o.txt | cut -c1-300
