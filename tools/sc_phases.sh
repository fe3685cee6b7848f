#!/bin/bash
# Diagnostic: build the engine with per-phase cycle stamps in scatter_bases_kernel (-DKDB_SC_PROF) as a second library
# and run one short bench with it.  Usage (on the GPU box): tools/sc_phases.sh [bench args]
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKDB_SC_PROF -I include -o kmerdb_amd/libkdbhip_prof.so kmerdb_amd/csrc/kdb_engine.hip -lz -lpthread
KDB_LIB=$PWD/kmerdb_amd/libkdbhip_prof.so python bench.py --algo 2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-regions "$@" 2>&1 | grep -v "^{" | tail -8
