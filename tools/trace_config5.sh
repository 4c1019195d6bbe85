#!/bin/bash
# kernel trace of BASELINE config 5 (persistent index) on the GPU box: tools/trace_config5.sh <tag> -> gpurun_out/prof_<tag>/
tag=${1:-c5}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/prof_$tag
rm -rf "$D"; mkdir -p "$D"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$D/trace" --output-format csv -- python3 "$ROOT/tools/trace_config5.py" > "$D/log.txt" 2> "$D/err.txt"
cat "$D/log.txt"
python3 - "$D" <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "state_init" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    m = re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s > 15: print("%9.1f %9.1f %8.1f q=%s %s" % (s, e, e - s, r.get("Queue_Id", "?"), n))
PY
