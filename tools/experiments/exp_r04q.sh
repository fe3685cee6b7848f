#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKDB_SC_PROF -I include -o kmerdb_amd/libkdbhip_prof.so kmerdb_amd/csrc/kdb_engine.hip -lz -lpthread || exit 1
for P in 0.0005 0.005; do
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_prof.so timeout -k 10 300 python -u tools/experiments/prof_ragged.py 12 $P > $O/prof_ragged2_$P.txt 2>&1; echo rc=$?
grep "sc_prof\|ms per step" $O/prof_ragged2_$P.txt | grep -v "last launch" | cut -c1-420
done
