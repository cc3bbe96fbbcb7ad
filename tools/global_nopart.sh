#!/bin/bash
# global decode at 64-read steps: the decode partition (auto) against no partition with large groups, long runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export STEPS=320 LAG=128
for H in "" soft; do
  echo "== head: ${H:-saturated}"
  timeout -k 10 200 python tools/global_pipe_bench.py $H g8 2>&1 | tail -1
  timeout -k 10 300 python tools/global_pipe_bench.py $H part0 g8 g16 g32 g64 2>&1 | tail -4
done
