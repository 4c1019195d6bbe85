#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/probe_fused; rm -rf "$D"; mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
for v in fused fused_nocopy; do
  opt="--opt FUSED_ROWS"; [ $v = fused_nocopy ] && opt="--opt FUSED_ROWS --opt NO_FUSED_COMPACT"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$D/trace_$v" --output-format csv -- python3 "$ROOT/tools/bench_extract.py" $opt > "$D/prof_$v.log" 2>&1
  tail -6 "$D/prof_$v.log"
  f=$(ls $D/trace_$v/*/*_kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=re.search(r"(\w+_kernel(?:<[\w, ]+>)?)",r["Name"]); print("   ", (n.group(1) if n else r["Name"][:40]).ljust(44), r["Calls"].rjust(4), "%9.1f us avg" % (float(r["AverageNs"])/1e3))
PY
done
find "$D" -name "*.db" -delete 2>/dev/null || true
