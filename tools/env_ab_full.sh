#!/bin/bash
# the whole default bench (all legs) under two settings of one environment variable, interleaved: tools/env_ab_full.sh VAR a b [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
for r in $(seq 1 $REPS); do
  for v in $A $B; do
    echo "== $VAR=$v"
    env $VAR=$v timeout -k 10 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.2f' % (d['value']/1e6), ' '.join('%s %.2f' % (k.replace('secondary_',''), v['value']/1e6) for k,v in d.items() if k.startswith('secondary')))" || exit 1
  done
done
