#!/usr/bin/env python3
"""Prints the kernel timeline of the last association in a rocprofv3 kernel trace (gpurun_out/prof_<tag>)."""
import csv, glob, re, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/trace/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "curvature" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
k = 0
for r in seg:
    m = re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if "associate" in n or "lm_begin" in n:
        print("%9.1f %9.1f %8.1f  q=%s %s" % (s, e, e - s, r.get("Queue_Id", "?"), n))
        k += 1
    if "lm_begin" in n and k > 3:
        break
