#!/usr/bin/env python3
"""Summarise the counter CSVs of tools/pmc_passes.sh: per counter, the value of the LAST
dispatch of the search kernel (10 000 queries), plus a few derived figures."""
import csv
import glob
import json
import sys

root = sys.argv[1]
vals = {}
dur = None
for f in sorted(glob.glob(f"{root}/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "hnsw_search_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            lds = r["LDS_Block_Size"]; vg = r["VGPR_Count"]
out = {"counters": {k: int(v) for k, v in sorted(vals.items())}, "kernel_ns_under_pmc": dur, "lds_block": lds,
       # rocprofv3 reports half of what a wave64 kernel allocates: 56 here = 112 allocated (the code object says 110)
       "vgpr_count_rocprof": vg}
if "FETCH_SIZE" in vals:
    # FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 for wide reads: x2
    out["hbm_read_bytes_raw"] = int(vals["FETCH_SIZE"] * 1024)
    out["hbm_read_bytes_x2"] = int(vals["FETCH_SIZE"] * 1024 * 2)
if "WRITE_SIZE" in vals:
    out["hbm_write_bytes"] = int(vals["WRITE_SIZE"] * 1024)
print(json.dumps(out, indent=1))
