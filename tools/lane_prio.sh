#!/bin/bash
# forward only on two lanes: equal queue priorities against lane 1 at the lowest priority (experiment build), several submit patterns
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export RADIAN_HIP_LIB=$R/tools/variants/libradian_hip_x.so
for P in "" "0,1" "0,0,1" "0,0,0,1" "0,0,0,0,0,0,0,1"; do
  for X in "" 1; do
    if [ -n "$X" ]; then export RD_X_LANE_PRIO=1; else unset RD_X_LANE_PRIO; fi
    if [ -n "$P" ]; then export LANE_PATTERN=$P; else unset LANE_PATTERN; fi
    timeout -k 10 120 python tools/fwd_lanes.py reads 2 120 1 64 | tail -1
  done
done
