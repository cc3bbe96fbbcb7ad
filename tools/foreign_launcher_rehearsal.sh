#!/bin/bash
# bench.py under a FOREIGN launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT set by someone else, as torch.distributed.run does):
# two ranks on one GPU, started by this shell -- both are children of the same process, which is what names their rendezvous directory.
# (no wrapper process around a rank: the ranks must be direct children of ONE process, as torch.distributed.run's are)
# usage (GPU box): timeout -k 10 300 bash tools/foreign_launcher_rehearsal.sh
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export RD_BENCH_DEVICE=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
RANK=0 LOCAL_RANK=0 python3 bench.py --gpus 2 --steps 6 --warmup 2 --preheat-ms 0 > gpurun_out/foreign_rank0.json 2> gpurun_out/foreign_rank0.err &
P0=$!
RANK=1 LOCAL_RANK=1 python3 bench.py --gpus 2 --steps 6 --warmup 2 --preheat-ms 0 > gpurun_out/foreign_rank1.json 2> gpurun_out/foreign_rank1.err &
P1=$!
wait $P0; R0=$?
wait $P1; R1=$?
echo "rank exit codes $R0 $R1"
python3 -c "
import json
d = json.load(open('gpurun_out/foreign_rank0.json'))
print({k: d[k] for k in ('n_gpus', 'startup_comm', 'rccl_nranks', 'ms_per_step_per_rank')}, d['config']['launcher'])
print('rank 1 stdout bytes:', len(open('gpurun_out/foreign_rank1.json').read()))
"
ls /tmp | grep radian_rccl_uid || echo "rendezvous files removed"
