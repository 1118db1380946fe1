#!/usr/bin/env python3
"""Copy the summaries of the last tools/profile_headline.sh run from gpurun_out/ into profiles/ (tag as argument)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "v4"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"      # round prefix of the file names
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *p: os.path.join(R, *p)
shutil.copy(g("gpurun_out", "pmc_v8", "summary.json"), g("profiles", f"{rnd}_pmc_search_kernel_{tag}.json"))
shutil.copy(g("gpurun_out", "pmc_v8", "summary.json"), g("profiles", "pmc_search_kernel_latest.json"))
shutil.copy(g("gpurun_out", "bench_final_v6.json"), g("profiles", f"{rnd}_bench_1m_{tag}.json"))
shutil.copy(g("gpurun_out", "prof_v6", "bench_under_rocprof.json"), g("profiles", f"{rnd}_bench_1m_{tag}_under_rocprof.json"))
stats = max(glob.glob(g("gpurun_out", "prof_v6", "vm", "*_kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(stats)))
with open(g("profiles", f"{rnd}_bench_1m_{tag}_kernel_stats.csv"), "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
trace = max(glob.glob(g("gpurun_out", "prof_v6", "vm", "*_kernel_trace.csv")), key=os.path.getmtime)
tr = [r for r in csv.DictReader(open(trace)) if "hnsw_search_kernel" in r["Kernel_Name"]]
keys = [k for k in ["Dispatch_Id", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                    "Workgroup_Size", "Grid_Size"] if k in tr[0]]
with open(g("profiles", f"{rnd}_bench_1m_{tag}_search_kernel_dispatches.csv"), "w", newline="") as o:
    w = csv.writer(o); w.writerow(keys + ["duration_ns"])
    for r in tr:
        w.writerow([r[k] for k in keys] + [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
print("search kernel dispatches", len(d), "median ns (the timed steps; the first ones are the ef sweep, the last ones the large batch)", sorted(d)[len(d) // 2])
print("under rocprof:", json.load(open(g("profiles", f"{rnd}_bench_1m_{tag}_under_rocprof.json")))["roofline"])
print("pmc:", {k: v for k, v in json.load(open(g("profiles", "pmc_search_kernel_latest.json"))).items() if k != "counters"})
c = json.load(open(g("profiles", "pmc_search_kernel_latest.json")))["counters"]
print({k: c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "TA_TA_BUSY_sum", "SQ_WAVE_CYCLES") if k in c})
