#!/usr/bin/env python
"""Print the kernel timeline of the last full SVI steps from a rocprofv3 kernel_trace.csv."""
import csv, glob, os, sys
src = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "vc_pre_kernel" in r["Kernel_Name"]]
start = idx[-nsteps - 1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start: idx[-1]]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.2f} -> {e/1e3:9.2f}  ({(e-s)/1e3:7.2f} us)  {r['Kernel_Name'][:70]}")
