#!/bin/bash
# one counter pass on the GPU box: tools/pmc_gpu.sh <tag> "<counters>" [kernel regex]
set -e
tag=$1; ctrs=$2; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/pmc_$tag
rm -rf "$D"; mkdir -p "$D"; cd /tmp && export TMPDIR=/tmp
LOAMX_NO_AUX_STREAM=1 timeout -k 10 300 rocprofv3 --pmc $ctrs -d "$D/out" --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-streamed > "$D/log.txt" 2> "$D/err.txt"
python3 - "$D" "${3:-knn|fit|select}" <<'PY'
import csv,glob,sys,re,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(sys.argv[1]+'/out/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:40]
        a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k in sorted(agg):
    if re.search(sys.argv[2],k): print(k, {c:"%.4g"%(v[1]/v[0]) for c,v in agg[k].items()})
PY
