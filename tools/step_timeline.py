#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST bench step in a rocprofv3 kernel trace (gpurun_out/prof_<tag>/trace): start,
end, duration (us), queue and kernel of every dispatch, and the idle gaps of the device between dispatches.

    python tools/step_timeline.py gpurun_out/prof_r03a [trace|trace_seq]
"""
import csv, glob, re, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "trace")
files = glob.glob(d + "/" + sub + "/*/*_kernel_trace.csv")
if len(files) != 1:
    sys.exit("step_timeline: %d traces under %s/%s (remove the directory before profiling into it again)" % (len(files), d, sub))
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "curvature" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
busy_until, idle = 0.0, 0.0
for r in seg:
    m = re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - busy_until if busy_until and s > busy_until else 0.0
    idle += gap
    print("%9.1f %9.1f %8.1f  gap %6.1f q=%s %s" % (s, e, e - s, gap, r.get("Queue_Id", "?"), n))
    busy_until = max(busy_until, e)
print("step: %.1f us, device idle between dispatches: %.1f us" % (busy_until, idle))
