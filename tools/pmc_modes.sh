#!/bin/bash
# partition_kernel's duration varies by +-8 % from process to process.  Which hardware counter moves with
# it?  Several processes per counter set, kernel trace + counters in the same run; prints (duration, counters) pairs.
OUT=gpurun_out/pmc_modes; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
SETS=("TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
      "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
      "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum")
REPS=${1:-4}
i=0
for S in "${SETS[@]}"; do
  for r in $(seq 1 $REPS); do
    rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT/s${i}_r$r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/s${i}_r$r.json 2> $OUT/s${i}_r$r.err || echo "set $i rep $r failed"
  done
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pmc_modes/s*_r*")):
    if not os.path.isdir(d): continue
    kt = glob.glob(d + "/*/*kernel_trace.csv"); cc = glob.glob(d + "/*/*counter_collection.csv")
    if not kt or not cc: print(d, "missing csv"); continue
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt[0])) if "partition_kernel" in r["Kernel_Name"]]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if "partition_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), "P1 ms %.3f" % (sum(dur) / max(len(dur), 1)), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
