#!/bin/bash
# one counter pass over tools/bench_extract.py on the GPU box: tools/pmc_extract.sh <tag> "<counters>" [bench_extract args]
tag=$1; ctrs=$2; shift; shift; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/pmcx_$tag
rm -rf "$D"; mkdir -p "$D"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctrs -d "$D/out" --output-format csv -- python3 "$ROOT/tools/bench_extract.py" --reps 1 "$@" > "$D/log.txt" 2> "$D/err.txt"
python3 - "$D" <<'PY'
import csv,glob,sys,re,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(sys.argv[1]+'/out/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(\w+_kernel)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:40]
        a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k in sorted(agg):
    if re.search("select|curv",k): print(k, {c:"%.4g"%(v[1]/v[0]) for c,v in agg[k].items()})
PY
find "$D" -name "*.db" -delete 2>/dev/null || true
