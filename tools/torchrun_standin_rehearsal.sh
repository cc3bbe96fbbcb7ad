#!/bin/bash
# The round-end driver's own multi-GPU command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
# --master-port P bench.py --gpus N --steps K --warmup W` -- on a ONE-GPU box with N = 4: the ranks' collectives go through
# tests/standin_rccl.cpp (TEST INFRASTRUCTURE) as librccl.so.1, since real RCCL refuses two ranks on one device.  One JSON line on stdout:
# startup_comm "rccl", rccl_nranks 4, launcher "foreign".   usage (through gpurun): bash tools/torchrun_standin_rehearsal.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d)
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -I/opt/rocm/include -o $D/librccl.so.1 $R/tests/standin_rccl.cpp
cd $R
LD_LIBRARY_PATH=$D:$LD_LIBRARY_PATH python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 4 --steps 10 --warmup 2
rm -rf $D
