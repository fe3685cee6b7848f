"""Per-kernel instruction statistics of a `hipcc -save-temps` assembly file (gfx950): VALU / v_mov / LDS / waits on lgkmcnt(0), registers, scratch.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -save-temps -o /tmp/isa/lib.so kmerdb_amd/csrc/kdb_engine.hip
    python tools/isa_stats.py /tmp/isa/kdb_engine-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]
A returning LDS atomic followed at once by `s_waitcnt lgkmcnt(0)`, or dozens of v_mov per atomic, is how round 4 found the histogram adds
the compiler had serialised."""
import re
import subprocess
import sys


def kernel_stats(path):
    """-> {demangled kernel name: {insts, valu, v_mov, salu, ds, ds_rtn_atomics, vmem, waits_lgkmcnt0, barriers, vgprs, scratch, occupancy, lds}}"""
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
    out = {}
    for n, d in zip(names, dem):
        b = funcs[n]
        ins = [x.strip() for x in b if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        cnt = lambda p: sum(1 for x in ins if x.startswith(p))          # noqa: E731
        info = {k: next((x.split(':')[1].strip() for x in b if ('; ' + k + ':') in x), '?') for k in ('NumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize')}
        num = lambda v: int(v.split(' ')[0]) if v.split(' ')[0].isdigit() else None          # noqa: E731
        out[d] = {"insts": len(ins), "valu": cnt('v_'), "v_mov": cnt('v_mov'), "salu": cnt('s_') - cnt('s_waitcnt'), "ds": cnt('ds_'),
                  "ds_rtn_atomics": sum(1 for x in ins if x.startswith('ds_') and '_rtn_' in x), "vmem": cnt('global_') + cnt('buffer_') + cnt('flat_'),
                  "waits_lgkmcnt0": sum(1 for x in ins if x.startswith('s_waitcnt') and 'lgkmcnt(0)' in x), "barriers": cnt('s_barrier'),
                  "vgprs": num(info['NumVgprs']), "scratch": num(info['ScratchSize']), "occupancy": num(info['Occupancy']), "lds": num(info['LDSByteSize'])}
    return out


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for d, v in kernel_stats(path).items():
        if pats and not any(p in d for p in pats):
            continue
        print("%s\n    insts %d  valu %d  v_mov %d  salu %d  ds %d  ds_rtn_atomics %d  vmem %d  waits_lgkmcnt0 %d  barriers %d | vgprs %s scratch %s occ %s lds %s" % (
            d[:150], v["insts"], v["valu"], v["v_mov"], v["salu"], v["ds"], v["ds_rtn_atomics"], v["vmem"], v["waits_lgkmcnt0"], v["barriers"],
            v["vgprs"], v["scratch"], v["occupancy"], v["lds"]))


if __name__ == "__main__":
    main()
