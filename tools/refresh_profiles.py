#!/usr/bin/env python3
"""Copy the summaries of the last tools/profile_headline.sh run from gpurun_out/ into profiles/ (tag as argument)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "v4"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"      # round prefix of the file names
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *p: os.path.join(R, *p)
shutil.copy(g("gpurun_out", "pmc_r3", "summary.json"), g("profiles", f"{rnd}_pmc_search_kernel_{tag}.json"))
shutil.copy(g("gpurun_out", "pmc_r3", "summary.json"), g("profiles", "pmc_search_kernel_latest.json"))
shutil.copy(g("gpurun_out", "bench_final_r3.json"), g("profiles", f"{rnd}_bench_1m_{tag}.json"))
shutil.copy(g("gpurun_out", "prof_r3", "bench_under_rocprof.json"), g("profiles", f"{rnd}_bench_1m_{tag}_under_rocprof.json"))
stats = max(glob.glob(g("gpurun_out", "prof_r3", "vm", "*_kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(stats)))
with open(g("profiles", f"{rnd}_bench_1m_{tag}_kernel_stats.csv"), "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
trace = max(glob.glob(g("gpurun_out", "prof_r3", "vm", "*_kernel_trace.csv")), key=os.path.getmtime)
tr = [r for r in csv.DictReader(open(trace)) if "hnsw_search_kernel" in r["Kernel_Name"]]
keys = [k for k in ["Dispatch_Id", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                    "Workgroup_Size", "Grid_Size"] if k in tr[0]]
with open(g("profiles", f"{rnd}_bench_1m_{tag}_search_kernel_dispatches.csv"), "w", newline="") as o:
    # overlapped: the dispatch ran while another dispatch of the search kernel was still running (the bench's "pipelined" leg: batches
    # in flight on several streams); such a dispatch takes longer than a lone one and is not a timed step
    w = csv.writer(o); w.writerow(keys + ["duration_ns", "overlapped"])
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in tr]
    rows_out = []
    for i, r in enumerate(tr):
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ov = any(int(o2["Start_Timestamp"]) < e0 and int(o2["End_Timestamp"]) > s0 for j, o2 in enumerate(tr) if j != i)
        rows_out.append((e0 - s0, int(ov)))
        w.writerow([r[k] for k in keys] + [e0 - s0, int(ov)])
# the bench's order of search calls: ef sweep, counters, top-(k+1), warm-up, the timed steps (--steps 10 under the profiler), one call
# for the launch info, then the 40 000-query batch (the first dispatch of about 3 ms)
first_large = next((i for i, (dur, _) in enumerate(rows_out) if dur > 2_500_000), len(rows_out))
timed = [dur for dur, _ in rows_out[max(0, first_large - 11):max(0, first_large - 1)]]
print("search kernel dispatches", len(rows_out), "; the 10 timed steps (the dispatches before the launch-info call and the 40 000-query batch):",
      "mean", sum(timed) / max(1, len(timed)), "ns, min", min(timed or [0]), "max", max(timed or [0]))
print("under rocprof:", json.load(open(g("profiles", f"{rnd}_bench_1m_{tag}_under_rocprof.json")))["roofline"])
print("pmc:", {k: v for k, v in json.load(open(g("profiles", "pmc_search_kernel_latest.json"))).items() if k != "counters"})
c = json.load(open(g("profiles", "pmc_search_kernel_latest.json")))["counters"]
print({k: c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "TA_TA_BUSY_sum", "SQ_WAVE_CYCLES") if k in c})
