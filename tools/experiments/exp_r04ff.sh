#!/bin/bash
# count_smallk_kernel k = 8: id x 4 (address = one AND), wrap test as bfe / max3
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "k8_lds or reference_kdb_fixture or golden_parsefile or ragged_reads_with_n_at_scale or (random_reads_vs_oracle and 8)" > $O/t_ff.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/t_ff.txt
timeout -k 10 200 python -u tests/fuzz_gpu.py 150 9053 8,8,8,7 > $O/fuzz_m.txt 2>&1; echo "fuzz m rc=$?"; tail -n 1 $O/fuzz_m.txt | cut -c1-400
AB_STEPS=128 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k 8 --no-configs 2>&1 | tee $O/ab_ff_k8.txt
AB_STEPS=128 AB_LIBS="libkdbhip_base.so libkdbhip.so" timeout -k 10 400 bash tools/ab_libs.sh --k 8 --no-configs --forward 2>&1 | tee $O/ab_ff_k8f.txt
