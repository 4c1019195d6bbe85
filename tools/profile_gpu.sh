#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 kernel trace + PMC passes of bench.py into
# gpurun_out/prof_<tag>/ (a fresh directory per tag; summarise with tools/summarize_profile.py).
#   bash tools/profile_gpu.sh <tag> ["CTR_A CTR_B" "CTR_C CTR_D" ...]   # extra counter sets, one pass each
set -e
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=$ROOT/gpurun_out/prof_$tag
rm -rf "$D"; mkdir -p "$D"
(cd "$ROOT" && python3 -c "from loam_amd import build; print(build.source_hash())" > "$D/source_sha256.txt")
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$D/trace" --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-streamed > "$D/bench_under_rocprof.log" 2> "$D/trace.err"
echo "trace done"
# the same with the association chains in sequence on one stream: clean per-kernel durations
# (kernels that run concurrently on the auxiliary stream report inflated durations above)
export LOAMX_NO_AUX_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$D/trace_seq" --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-streamed > "$D/bench_under_rocprof_seq.log" 2> "$D/trace_seq.err"
echo "sequential trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$D/pmc_fetch" --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed > "$D/pmc_fetch.log" 2> "$D/pmc_fetch.err"
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$D/pmc_write" --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed > "$D/pmc_write.log" 2> "$D/pmc_write.err"
echo "write done"
i=0
for set in "$@"; do
  timeout -k 10 300 rocprofv3 --pmc $set -d "$D/pmc_x$i" --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-streamed > "$D/pmc_x$i.log" 2> "$D/pmc_x$i.err" || echo "set $i ($set) failed"
  echo "set $i done: $set"
  i=$((i+1))
done
# keep the merged-back payload small: counter CSVs and stats only
find "$D" -name "*.db" -delete 2>/dev/null || true
du -sh "$D"
