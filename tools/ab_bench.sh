#!/bin/bash
# on the GPU box: bench.py A/B/A/B in one call (same box): tools/ab_bench.sh "<env for B>" [bench args]
envb=$1; shift
for i in 1 2; do
  for v in A B; do
    if [ $v = A ]; then e=""; else e="$envb"; fi
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-streamed --no-kernel-stats "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', '$e', j['value'], j['ms_per_step'])"
  done
done
