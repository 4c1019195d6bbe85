#!/bin/bash
# on the GPU box: rocprofv3 kernel stats of the extraction with the row-per-line selection and with select_mis_kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/${1:-ab_select}; rm -rf "$D"; mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
for v in rows mis; do
  opt=""; [ $v = mis ] && opt="--opt NO_ROW_SELECT"
  (cd $ROOT && timeout -k 10 120 python3 tools/bench_extract.py $opt > "$D/$v.txt" 2>&1); echo "$v rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$D/trace_$v" --output-format csv -- python3 "$ROOT/tools/bench_extract.py" $opt > "$D/prof_$v.log" 2>&1
  cat "$D/$v.txt"
  f=$(ls $D/trace_$v/*/*_kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=re.search(r"(\w+_kernel)",r["Name"]); print("   ", (n.group(1) if n else r["Name"][:40]).ljust(28), r["Calls"].rjust(4), "%9.1f us avg" % (float(r["AverageNs"])/1e3))
PY
done
find "$D" -name "*.db" -delete 2>/dev/null || true
