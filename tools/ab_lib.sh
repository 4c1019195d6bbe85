#!/bin/bash
# on the GPU box: A/B of two builds of libloamx.so (loam_amd/lib/libloamx_A.so / _B.so, made beforehand) in one call
# tools/ab_lib.sh [bench args]
for i in 1 2; do
  for v in A B; do
    cp loam_amd/lib/libloamx_$v.so loam_amd/lib/libloamx.so
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-streamed "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j.get('kernels',{})
print('$v', j['value'], j['ms_per_step'], ' '.join('%s=%.4f'%(n[:9],v['avg_ms']) for n,v in k.items() if v['launches']))"
  done
done
