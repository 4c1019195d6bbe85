#!/bin/bash
# on the GPU box (inside gpurun): the -m gpu suite + a short bench with the kernel table -> gpurun_out/<tag>/
#   bash tools/gpu_check.sh <tag> [bench args]
tag=${1:-chk}; shift
D=gpurun_out/$tag; mkdir -p "$D"
# -s: a runtime abort message must reach the log
timeout -k 10 700 python -m pytest tests -m gpu -x -q -s > "$D/gputests.log" 2>&1; echo "tests rc=$?"; tail -6 "$D/gputests.log"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-streamed "$@" > "$D/bench.json" 2> "$D/bench.err"; echo "bench rc=$?"
python3 - "$D/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["unit"], j["ms_per_step"], "ms/step", j["results"])
for k, v in j["kernels"].items():
    print("  ", k.ljust(26), str(v["launches"]).rjust(4), str(v["avg_ms"]).rjust(10), str(v["achieved_GBs"]).rjust(9), v["hbm_frac"])
PY
