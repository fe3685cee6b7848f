#!/bin/bash
# One device, every k: ms per step of the resident-input region (bench.py --k K), the final build.   -> gpurun_out/bench_all_k.json (one JSON object per line)
OUT=gpurun_out/bench_all_k.json
: > $OUT
for K in 8 9 10 11 12 13 14 15 16 17; do
  STEPS=200; [ $K -ge 13 ] && STEPS=64; [ $K -eq 17 ] && STEPS=128
  python bench.py --k $K --steps $STEPS --warmup 3 --no-cpu-baseline --no-extra-regions --no-configs > gpurun_out/t.json 2> gpurun_out/t.err || { echo "{\"k\": $K, \"failed\": true}" >> $OUT; continue; }
  python - >> $OUT <<PY
import json
d = json.load(open("gpurun_out/t.json"))
r = d["roofline"]
print(json.dumps({"k": $K, "steps": d["steps"], "ms_per_step": d["ms_per_step"], "gbase_per_s": round(d["config"]["reads_per_gpu_per_step"] * d["config"]["read_len"] / d["ms_per_step"] / 1e6, 1) if "reads_per_gpu_per_step" in d["config"] else None,
                  "kernels_ms_per_step": {k: round(v, 4) for k, v in r["kernels_ms_per_step"].items()},
                  "hbm_frac_by_kernel": {k: v.get("hbm_frac") for k, v in r["per_kernel"].items() if v.get("hbm_frac") is not None},
                  "batches_per_flush": (r.get("arena") or {}).get("batches_per_flush")}))
PY
done
cat $OUT
