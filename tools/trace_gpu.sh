#!/bin/bash
# quick per-kernel timing on the GPU box: rocprofv3 --kernel-trace --stats of a short bench run
set -e
tag=${1:-t}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/prof_$tag
rm -rf "$D"; mkdir -p "$D"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$D/trace" --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$D/bench_under_rocprof.log" 2> "$D/trace.err"
python3 - "$D" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/trace/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.3: print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
