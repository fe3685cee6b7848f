#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-dispatch PMC rows) into a small markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def noop_dispatch(r):
    """The scatter / small-k kernels are compiled per batch shape (records of one length / ragged) and both are launched: the one that
    does not apply returns at once.  Such dispatches (a few microseconds) are left out of every per-launch mean."""
    n = r.get("Kernel_Name", "")
    if "scatter_bases_kernel" not in n and "count_smallk_kernel" not in n:
        return False
    try:
        return int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 8000
    except (KeyError, ValueError):
        return False


print(f"# rocprofv3 summary ({out})\n")
for f in find("stats/**/*kernel_stats.csv"):
    print(f"## kernel stats ({os.path.relpath(f, out)})\n")
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        if ("scatter_bases_kernel" in r["Name"] or "count_smallk_kernel" in r["Name"]) and float(r["AverageNs"]) < 8000:
            r["Name"] = "(returns at once: not this batch's shape) " + r["Name"]
        name = r["Name"].split("(")[0][:70] if not r["Name"].startswith("(returns") else r["Name"][:110]
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
              f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    print()
for d in find("pmc_*"):
    if not os.path.isdir(d):
        continue
    cname = os.path.basename(d)[4:]
    files = [f for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)]
    if not files:
        print(f"## {cname}: no counter_collection.csv\n")
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname or noop_dispatch(r):
                continue
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    print(f"## {cname} per dispatch (mean over dispatches)\n")
    print("| kernel | dispatches | mean value |")
    print("|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {n} | {v / n:.1f} |")
    print()


# ---- machine-readable traffic summary for bench.py (HBM bytes per step of the kdb:: kernels) -------------------
import json
workload, bl = {}, {}
try:
    bl = json.load(open(os.path.join(out, "bench_stats.json")))
    cfg = bl.get("config", {})
    workload = {"k": cfg.get("k"), "reads": cfg.get("reads_per_gpu_per_step"), "read_len": cfg.get("read_len"), "canonical": cfg.get("canonical"),
                "algo": "direct" if str(cfg.get("algo")).startswith(("direct", "1")) else "lds"}
    rf = bl.get("roofline", {})
    workload["steps"] = bl.get("steps")
    workload["distinct_batches"] = rf.get("distinct_batches", 1)
    workload["batches_per_flush"] = (rf.get("arena") or {}).get("batches_per_flush")
    print(f"workload of every pass: k = {workload['k']}, {workload['reads']} reads x {workload['read_len']} bp per step, {workload['steps']} steps rotating through "
          f"{workload['distinct_batches']} distinct batches; batches per histogram flush: {workload['batches_per_flush']}\n")
    print("engine's own account of the stats pass (bench.py roofline.per_kernel): " +
          "; ".join(f"{n}: {v.get('avg_ms')} ms avg, {v.get('gbs')} GB/s" for n, v in rf.get("per_kernel", {}).items()) + "\n")
except Exception:
    pass
# FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced
# streaming reads (MI355X_MICROARCH.md, HBM) -> doubled for the kernels whose reads are such streams; mark_reads_kernel
# does byte loads (uncalibrated) and is taken as reported.
WIDE = ("count_direct_kernel", "count_lds_kernel", "stats_kernel",
        "scatter_bases_kernel", "scatter_ids_kernel", "page_hist_kernel", "hibit_check_kernel", "fold_kernel")
per = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in find(f"pmc_{cname}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname or "kdb::" not in r["Kernel_Name"] or noop_dispatch(r):
                continue
            kname = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "").split("<")[0]
            d = per.setdefault(kname, {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
            d[cname][0] += 1
            d[cname][1] += float(r["Counter_Value"]) * 1024.0
out_k = {}
total = 0.0
for kname, d in per.items():
    if kname == "stats_kernel":
        continue
    fetch = d["FETCH_SIZE"][1] / max(d["FETCH_SIZE"][0], 1) * (2.0 if kname in WIDE else 1.0)
    write = d["WRITE_SIZE"][1] / max(d["WRITE_SIZE"][0], 1)
    out_k[kname] = {"read_bytes": round(fetch), "write_bytes": round(write)}
    total += fetch + write
json.dump({**workload, "hbm_bytes_per_step": round(total), "per_kernel_per_launch": out_k,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, KiB*1024; FETCH doubled for 16 B/lane streaming kernels (gfx950)"},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)


# ---- SQ / LDS utilisation of the kdb:: kernels (separate --pmc passes sq_a, sq_b) -> lds.json ---------------------
sq = {}
for f in find("pmc_sq_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "kdb::" not in r["Kernel_Name"] or noop_dispatch(r):
            continue
        kname = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "").split("<")[0]
        a = sq.setdefault(kname, {}).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
if sq:
    res = {}
    print("## SQ / LDS counters per dispatch (mean; summed over the 8 XCDs)\n")
    for kname, d in sq.items():
        m = {c: v / n for c, (n, v) in d.items()}
        res[kname] = {c: round(v) for c, v in m.items()}
        busy = m.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy and m.get("SQ_LDS_IDX_ACTIVE") is not None:
            # SQ_BUSY_CU_CYCLES sums busy cycles over CUs; SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD (4 SIMDs per CU)
            res[kname]["lds_busy_frac"] = round(m["SQ_LDS_IDX_ACTIVE"] / busy, 4)
            res[kname]["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
            res[kname]["valu_busy_frac"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (busy * 4.0), 4)
        print(f"### {kname}\n")
        for c in sorted(m):
            print(f"- {c}: {m[c]:.0f}")
        for c in ("lds_busy_frac", "lds_bank_conflict_share", "valu_busy_frac"):
            if c in res[kname]:
                print(f"- **{c}: {res[kname][c]}**")
        print()
    json.dump({**workload, "per_kernel_per_launch": res,
               "method": "rocprofv3 --pmc in two passes (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_VALU, SQ_BUSY_CU_CYCLES, ...); "
                         "lds_busy = LDS_IDX_ACTIVE / BUSY_CU_CYCLES; valu_busy = ACTIVE_INST_VALU (quad-cycles per SIMD) x 4 / (4 SIMDs x BUSY_CU_CYCLES)"},
              open(os.path.join(out, "lds.json"), "w"), indent=1)


# ---- the timed region only: the last N dispatches of each kernel (N = its launches in bench.py's timed region) ------------------
# At k >= 13 the histogram pass runs once per flush and its bytes depend on what the flush holds, so a mean over ALL dispatches of
# a run (warm-up flushes of 8 and 16 batches included) does not compare with the line bench.py prints.  Every pass runs the same
# command, hence the same dispatch sequence: the last N dispatches of a kernel are the timed region's in every pass.
def last_n(rows, n):
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    return rows[-n:] if n > 0 else []


try:
    rf = bl.get("roofline", {})
    steps = int(bl.get("steps") or 0)
    want = {}
    for name, v in rf.get("per_kernel", {}).items():
        base = name.split("<")[0]
        if base in ("scatter_bases_kernel", "scatter_ids_kernel", "page_hist_kernel"):
            want[base] = (int(round(v["launches_per_step"] * steps)), v)
    trace = {}
    for f in find("stats/**/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "kdb::" in r["Kernel_Name"] and not noop_dispatch(r):
                kname = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "").split("<")[0]
                trace.setdefault(kname, []).append(r)
    pmc = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in find(f"pmc_{cname}/**/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == cname and "kdb::" in r["Kernel_Name"] and not noop_dispatch(r):
                    kname = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "").split("<")[0]
                    pmc.setdefault((kname, cname), []).append(r)
    if want and trace:
        print("## timed region only (last N dispatches of each kernel; N = launches in bench.py's timed region)\n")
        print("| kernel | N | avg ms (rocprofv3 trace) | avg ms (bench.py, HIP events) | PMC read GB | PMC write GB | PMC GB/s | frac of 8 TB/s | engine-counted GB | engine GB/s (bench.py) | PMC / engine bytes |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        timed = {}
        for base, (n, v) in want.items():
            tr = last_n(trace.get(base, []), n)
            if not tr:
                continue
            avg_ms = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) / len(tr) / 1e6
            rd = last_n(pmc.get((base, "FETCH_SIZE"), []), n)
            wr = last_n(pmc.get((base, "WRITE_SIZE"), []), n)
            rb = sum(float(r["Counter_Value"]) for r in rd) / max(len(rd), 1) * 1024.0 * (2.0 if base in WIDE else 1.0)
            wb = sum(float(r["Counter_Value"]) for r in wr) / max(len(wr), 1) * 1024.0
            eng_b = (v.get("read_bytes_per_step", 0) + v.get("write_bytes_per_step", 0)) / max(v["launches_per_step"], 1e-9)
            gbs = (rb + wb) / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
            timed[base] = {"launches": n, "avg_ms_trace": round(avg_ms, 4), "avg_ms_bench": v.get("avg_ms"), "pmc_read_bytes": round(rb), "pmc_write_bytes": round(wb),
                           "pmc_gbs": round(gbs, 1), "pmc_hbm_frac": round(gbs / 8000.0, 4), "engine_bytes": round(eng_b), "engine_gbs": v.get("gbs")}
            print(f"| {base} | {n} | {avg_ms:.4f} | {v.get('avg_ms')} | {rb / 1e9:.3f} | {wb / 1e9:.3f} | {gbs:.0f} | {gbs / 8000.0:.3f} | {eng_b / 1e9:.3f} | {v.get('gbs')} | "
                  f"{(rb + wb) / eng_b if eng_b else 0:.3f} |")
        print()
        tj = json.load(open(os.path.join(out, "traffic.json")))
        tj["timed_region_per_launch"] = timed
        if steps:
            # HBM bytes of a step of the timed region: every kernel's bytes x its launches / steps (the histogram pass of k >= 13 runs once per flush);
            # the small kernels (page sort, record geometry) as their per-launch means, once per step
            big = sum((v["pmc_read_bytes"] + v["pmc_write_bytes"]) * v["launches"] for v in timed.values()) / steps
            small = sum(v["read_bytes"] + v["write_bytes"] for n, v in tj.get("per_kernel_per_launch", {}).items() if n not in timed)
            tj["hbm_bytes_per_step"] = round(big + small)
            tj["hbm_bytes_per_step_is"] = "timed region: per-kernel PMC bytes x launches / steps"
        json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print(f"(timed-region table not produced: {type(e).__name__}: {e})")
