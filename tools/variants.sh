#!/bin/bash
# Times the conv launches of every experiment build under radian_amd/variants/ (python -m radian_amd.build -D... -o...).
for so in radian_amd/variants/lib_*.so; do
  echo "== $so"
  RADIAN_HIP_LIB=$PWD/$so timeout -k 10 120 python tools/layer_times.py || exit 1
done
