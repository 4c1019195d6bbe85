#!/bin/bash
# quick A/B on the GPU box: tools/kbench.sh <tag> [bench args] — bench line summary into gpurun_out/<tag>.txt
tag=${1:-kb}; shift
timeout -k 10 300 python bench.py --no-cpu-baseline --no-streamed "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err; echo "bench rc=$?"
python3 - gpurun_out/$tag.json <<'PY' | tee gpurun_out/$tag.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["unit"], j["ms_per_step"], "ms/step", j["results"])
for k, v in j["kernels"].items():
    print("  ", k.ljust(26), str(v["launches"]).rjust(4), str(v["avg_ms"]).rjust(10), str(v["achieved_GBs"]).rjust(9), v["hbm_frac"])
PY
