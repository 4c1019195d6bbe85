#!/bin/bash
# quick kernel table on the GPU box: tools/kbench.sh [bench args]
timeout -k 10 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/kbench.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/kbench.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(j['value'],j['unit'],j['ms_per_step'],'ms/step', j['results'])
for k,v in j['kernels'].items(): print('  ',k.ljust(26),str(v['launches']).rjust(4),str(v['avg_ms']).rjust(10),str(v['achieved_GBs']).rjust(9),v['hbm_frac'])
"
