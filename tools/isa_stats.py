"""Per-kernel instruction statistics of a `hipcc -save-temps` assembly file (gfx950): VALU / v_mov / LDS / waits on lgkmcnt(0), registers, scratch.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -save-temps -o /tmp/isa/lib.so kmerdb_amd/csrc/kdb_engine.hip
    python tools/isa_stats.py /tmp/isa/kdb_engine-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]
A returning LDS atomic followed at once by `s_waitcnt lgkmcnt(0)`, or dozens of v_mov per atomic, is how round 4 found the histogram adds
the compiler had serialised."""
import re
import subprocess
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    funcs, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(line.rstrip("\n"))
    names = [n for n, b in funcs.items() if any('s_endpgm' in x for x in b)]
    dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, d in zip(names, dem):
        if pats and not any(p in d for p in pats):
            continue
        b = funcs[n]
        ins = [x.strip() for x in b if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        cnt = lambda p: sum(1 for x in ins if x.startswith(p))          # noqa: E731
        info = {k: next((x.split(':')[1].strip() for x in b if ('; ' + k + ':') in x), '?') for k in ('NumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize')}
        print("%s\n    insts %d  valu %d  v_mov %d  salu %d  ds %d  ds_rtn_atomics %d  vmem %d  waits_lgkmcnt0 %d  barriers %d | vgprs %s scratch %s occ %s lds %s" % (
            d[:150], len(ins), cnt('v_'), cnt('v_mov'), cnt('s_') - cnt('s_waitcnt'), cnt('ds_'), sum(1 for x in ins if x.startswith('ds_') and '_rtn_' in x),
            cnt('global_') + cnt('buffer_') + cnt('flat_'), sum(1 for x in ins if x.startswith('s_waitcnt') and 'lgkmcnt(0)' in x), cnt('s_barrier'),
            info['NumVgprs'], info['ScratchSize'], info['Occupancy'], info['LDSByteSize'].split(' ')[0]))


if __name__ == "__main__":
    main()
