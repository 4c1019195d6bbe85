#!/bin/bash
# on the GPU box: rocprofv3 kernel trace of bench.py (3 steps) and the timeline of the last step -> gpurun_out/<tag>/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; D=$ROOT/gpurun_out/${1:-trace_step}; rm -rf "$D"; mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$D/trace" --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-kernel-stats > "$D/bench.log" 2> "$D/trace.err"
find "$D" -name "*.db" -delete 2>/dev/null || true
python3 - "$D" <<'PY'
import csv,glob,re,sys
f=glob.glob(sys.argv[1]+'/trace/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'curvature_valid2' in r['Kernel_Name']]
a=idx[-2]; b=idx[-1]; t0=int(rows[a]['Start_Timestamp']); last=0
for r in rows[a:b]:
    n=re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", r['Kernel_Name']); n=n.group(1) if n else r['Kernel_Name'][:30]
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    if e-s>10: print("%9.1f %9.1f %8.1f  %s"%(s,e,e-s,n))
print("step", (int(rows[b]['Start_Timestamp'])-t0)/1e3)
PY
