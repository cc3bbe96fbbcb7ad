#!/bin/bash
# conv launch times in split-f16 mode for every experiment build under radian_amd/variants/
for so in radian_amd/variants/lib_*.so; do
  echo "== $so"
  RADIAN_HIP_LIB=$PWD/$so timeout -k 10 120 python tools/layer_times.py f16x3 || exit 1
done
