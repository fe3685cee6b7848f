#!/bin/bash
# SQ/LDS utilisation counters for the kdb:: kernels of the bench command -> gpurun_out/pmc_sq_<tag>.txt
TAG=${1:-x}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kdb::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:26s} {v / n:16.0f}")
PY
