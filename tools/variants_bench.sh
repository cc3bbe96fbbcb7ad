#!/bin/bash
# bench.py (pipelined, beam search overlapped) for every experiment build under radian_amd/variants/
for so in radian_amd/variants/lib_*.so; do
  echo "== $so"
  RADIAN_HIP_LIB=$PWD/$so timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.3f M samples/s, %.3f ms/step, conv %.4f ms/launch (frac %.3f), decode %.2f ms/step' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['decode_ms_per_step']))" || exit 1
done
