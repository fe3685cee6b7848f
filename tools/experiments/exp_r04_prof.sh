#!/bin/bash
# round 4: rocprofv3 kernel stats + PMC passes of the final build, every pass the same command (tools/profile_gpu.sh)
export PYTHONUNBUFFERED=1
for K in "$@"; do
  ST=20; [ $K -ge 13 ] && ST=24; [ $K -eq 17 ] && ST=28
  STEPS=$ST bash tools/profile_gpu.sh r04_k$K --k $K > gpurun_out/prof_r04_k$K.log 2>&1; echo "k=$K rc=$?"
  tail -25 gpurun_out/prof_r04_k$K/summary.md | cut -c1-250
done
