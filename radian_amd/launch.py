"""Multi-GPU driver: `python -m radian_amd.basecall ... --gpus N` forks one worker process per GPU
(this module with --worker), each basecalls the blocks of reads it claims from a per-node work queue on its own device and
writes its results to a scratch file; the parent (which never touches a GPU) merges them in input order into
the reference's FASTA layout.  The only inter-GPU traffic is the start-up RCCL broadcast of the artefacts."""
import json
import os
import subprocess
import sys
import tempfile


def run_multi_gpu(args, argv):
    from .basecall import FastaWriter
    from .dist import merge_results
    world = args.gpus
    scratch = tempfile.mkdtemp(prefix="radian_mgpu_")
    tag = f"cli{os.getpid()}"
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "radian_amd.launch", "--worker", scratch, tag, "--"] + list(argv)
        procs.append(subprocess.Popen(cmd, env=env))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"worker exit codes {rcs}")
    per_rank = []
    for rank in range(world):
        with open(os.path.join(scratch, f"rank{rank}.jsonl")) as f:
            per_rank.append([tuple(json.loads(l)) for l in f])
    writer = FastaWriter(args.fasta_dir)
    for _, rid, seq in merge_results(per_rank):
        writer.write(rid, seq)
    writer.close()


def worker(scratch, tag, argv):
    from .backend import Backend
    from .basecall import build_parser, make_stitch_pool, run, setup_backend
    from .dist import FileComm, RcclComm, WorkQueue, env_rank_world, uid_path
    args = build_parser().parse_args(argv)
    rank, local_rank, world = env_rank_world()
    pool = make_stitch_pool(args.stitch_workers) if args.decode_type == "chunk" else None   # before the GPU is touched
    be = Backend(int(os.environ.get("RD_CLI_DEVICE", local_rank)))   # override only for rehearsals on a 1-GPU box
    try:
        comm = RcclComm(be, rank, world, uid_path(tag))
    except Exception as e:   # no usable RCCL communicator: every rank loads the artefacts itself, file-based barrier
        print(f"[rank {rank}] RCCL start-up failed ({e}); loading the model per rank", file=sys.stderr)
        comm = FileComm(rank, world, os.path.join(scratch, "fc"))
    # rank 0 parses / repacks the artefacts; everyone gets the device images by one broadcast
    holder = {}

    def load(b):
        setup_backend(args, b)
        holder["lm"] = args._lm_loaded

    comm.bcast_artifacts(be, load)
    be.set_precision(args.precision)   # context state, not part of the broadcast images
    # the flag `_lm_loaded` is host state: recompute it on the other ranks without touching the files' contents
    if rank != 0:
        args._lm_loaded = (args.rna_model != "None" and args.decode_type == "global")
    # reads go to the ranks through a work queue (a counter file in the launcher's scratch directory): dynamic balance
    queue = WorkQueue(os.path.join(scratch, "queue"), args.queue_block) if args.queue_block > 0 else None
    results = run(args, be, writer=None, shard=(rank, world), stitch_pool=pool, queue=queue)
    if queue is not None:
        queue.close()
    with open(os.path.join(scratch, f"rank{rank}.jsonl"), "w") as f:
        for r in results:
            f.write(json.dumps(list(r)) + "\n")
    comm.barrier()
    comm.close()
    be.close()
    if pool is not None:
        pool.shutdown()


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3], sys.argv[5:])
    else:
        raise SystemExit("internal entry point; use python -m radian_amd.basecall ... --gpus N")
