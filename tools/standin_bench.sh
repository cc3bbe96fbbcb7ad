#!/bin/bash
# `python bench.py --gpus N --check` on a ONE-GPU box with the product's whole N > 1 path executing: the ranks find tests/standin_rccl.cpp
# (TEST INFRASTRUCTURE: rccl.h's seven entry points over shared memory) as librccl.so.1, since real RCCL refuses two ranks on one device.
# What the line shows: startup_comm "rccl", rccl_nranks N, every rank's step time, rank 0's host budget.  What it cannot show: RCCL / xGMI.
# Round 6: the line also carries secondary_e2e_fast5_to_fasta -- the N-rank job from a fast5 directory to FASTA files through the multi-GPU route.
# usage (through gpurun): tools/standin_bench.sh [N=4] [more bench.py flags]      (N <= 5: the box's process guard allows six GPU processes)
set -e
N=${1:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d)
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -I/opt/rocm/include -o $D/librccl.so.1 $R/tests/standin_rccl.cpp
shift || true
LD_LIBRARY_PATH=$D:$LD_LIBRARY_PATH python $R/bench.py --gpus $N --steps 10 --warmup 2 --check "$@"
rm -rf $D
