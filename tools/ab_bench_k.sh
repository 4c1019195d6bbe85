#!/bin/bash
# as tools/ab_bench.sh, with the kernel table (event scopes) of every run
envb=$1; shift
for i in 1 2; do
  for v in A B; do
    if [ $v = A ]; then e=""; else e="$envb"; fi
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-streamed "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j.get('kernels',{})
print('$v', j['value'], j['ms_per_step'], ' '.join('%s=%.4f'%(n[:9],v['avg_ms']) for n,v in k.items() if v['launches']))"
  done
done
