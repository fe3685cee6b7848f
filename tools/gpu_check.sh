#!/bin/bash
# quick GPU regression: parity tests then short benches; writes under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
