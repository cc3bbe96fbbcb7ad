#!/bin/bash
# the whole default bench (all legs) for every experiment build under radian_amd/variants/: one line per leg
for so in radian_amd/variants/lib_*.so; do
  echo "== $so"
  RADIAN_HIP_LIB=$PWD/$so timeout -k 10 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.2f' % (d['value']/1e6), ' '.join('%s %.2f' % (k.replace('secondary_',''), v['value']/1e6) for k,v in d.items() if k.startswith('secondary')))" || exit 1
done
