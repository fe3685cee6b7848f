#!/bin/bash
# the whole -m gpu suite on the final build, as the driver runs it (one process), with durations
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 ) > $O/t_all_final.txt 2>&1; echo "tests rc=$?"; tail -n 24 $O/t_all_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
