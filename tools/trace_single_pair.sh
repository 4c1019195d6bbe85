#!/bin/bash
# kernel timeline of one single-pair registration (GPU box): bash tools/trace_single_pair.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf "$ROOT/gpurun_out/prof_sp"; mkdir -p "$ROOT/gpurun_out/prof_sp"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_sp" --output-format csv -- python3 "$ROOT/tools/single_pair_loop.py" 4 > /dev/null 2>&1
cd "$ROOT" && python3 - <<"PY"
import csv, glob, re
f=sorted(glob.glob("gpurun_out/prof_sp/*/*_kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
starts=[i for i,r in enumerate(rows) if "curvature" in r["Kernel_Name"]]
seg=rows[starts[-2]:starts[-1]]
t0=int(seg[0]["Start_Timestamp"]); pe=t0; busy=0
print("launches", len(seg), "span us", (int(seg[-1]["End_Timestamp"])-t0)/1e3)
for r in seg:
    m=re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r["Kernel_Name"]); n=m.group(1) if m else r["Kernel_Name"][:30]
    s,e=(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3
    busy+=e-s
    print("%8.1f %7.1f gap %6.1f %s"%(s,e-s,s-pe,n))
    pe=e
print("sum of kernel durations", busy)
PY
