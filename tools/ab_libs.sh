#!/bin/bash
# on the GPU box: several builds of libloamx.so (loam_amd/lib/libloamx_<tag>.so, made beforehand) timed in one call, twice round robin
# tools/ab_libs.sh "A B C" <sed filter for the kernel columns, or ''> [bench args]
tags=$1; shift; keep=$1; shift
for i in 1 2; do
  for v in $tags; do
    cp loam_amd/lib/libloamx_$v.so loam_amd/lib/libloamx.so
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-streamed "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys,re; j=json.loads(sys.stdin.read()); k=j.get('kernels',{})
print('$v', j['value'], j['ms_per_step'], ' '.join('%s=%.4f'%(n[:9],v['avg_ms']) for n,v in k.items() if v['launches'] and re.search('$keep' or '.', n)))"
  done
done
