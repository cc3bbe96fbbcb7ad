"""Multi-GPU driver: `python -m radian_amd.basecall ... --gpus N` starts one worker process per GPU (this module with
--worker).  The parent never touches a GPU: it validates the artefacts (host-only parsing, so a bad --sig-model /
--rna-model / --context-len fails here, before any rank can be left waiting inside a broadcast), lists the fast5 files
once, watches the workers (one failing worker stops the job), and merges the per-rank results -- a streaming k-way merge
by (file, read) -- into the reference's FASTA layout (basecall.py:129-138).  Each worker claims blocks of reads, file by
file, from a per-node work queue and opens only the files it claimed from.  The only inter-GPU traffic is the start-up
RCCL broadcast of the artefacts."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time


def merge_to_fasta(scratch, world, fasta_dir):
    """Per-rank result files -> reads-{n}.fasta in input order; one record per rank in memory at a time."""
    from .basecall import FastaWriter
    from .dist import merge_result_files
    writer = FastaWriter(fasta_dir)
    n = 0
    try:
        for _, rid, seq in merge_result_files([os.path.join(scratch, f"rank{r}.jsonl") for r in range(world)]):
            writer.write(rid, seq)
            n += 1
    finally:
        writer.close()
    return n


def wait_all(procs, poll=0.05):
    """Wait for every worker; as soon as one exits non-zero, stop the others (they may be blocked in a collective that
    will never complete) and return the exit codes."""
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t0 = time.time()
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, 10 - (time.time() - t0)))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            return [p.returncode for p in procs]
        if all(rc == 0 for rc in rcs):
            return rcs
        time.sleep(poll)


def run_multi_gpu(args, argv):
    from . import fast5
    from .basecall import load_artifacts
    world = args.gpus
    load_artifacts(args)   # host-only validation: raises here, in the parent, exactly what a single-GPU run would raise
    scratch = tempfile.mkdtemp(prefix="radian_mgpu_")
    try:
        with open(os.path.join(scratch, "files.json"), "w") as f:
            json.dump(fast5.list_files(args.fast5_dir), f)   # one enumeration: every rank sees the same file order
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            cmd = [sys.executable, "-m", "radian_amd.launch", "--worker", scratch, "--"] + list(argv)
            procs.append(subprocess.Popen(cmd, env=env))
        rcs = wait_all(procs)
        if any(rcs):
            raise SystemExit(f"multi-GPU run failed: worker exit codes {rcs}")
        merge_to_fasta(scratch, world, args.fasta_dir)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def run_rank(args, be, comm, scratch, sources, rank, world, stitch_pool=None, backends=None):
    """One rank's share of the job, after the artefacts are on its device: claim work, basecall, write rank{r}.jsonl
    (records in increasing (file, read) order), barrier."""
    from .basecall import run
    from .dist import FileReadQueue
    queue = FileReadQueue(os.path.join(scratch, "queue"), args.queue_block) if args.queue_block > 0 else None
    tmp = os.path.join(scratch, f"rank{rank}.jsonl.tmp")
    with open(tmp, "w") as f:
        def emit(key, rid, seq):
            f.write(json.dumps([list(key) if isinstance(key, tuple) else key, rid, seq]) + "\n")
        run(args, backends or be, writer=None, shard=(rank, world), stitch_pool=stitch_pool, queue=queue, sources=sources, on_result=emit)
    os.replace(tmp, os.path.join(scratch, f"rank{rank}.jsonl"))
    if queue is not None:
        queue.close()
    comm.barrier()
    return queue


def worker(scratch, argv):
    from . import fast5
    from .backend import Backend
    from .basecall import apply_artifacts, build_parser, load_artifacts, make_stitch_pool
    from .dist import connect, env_rank_world, uid_path
    args = build_parser().parse_args(argv)
    rank, local_rank, world = env_rank_world()
    pool = make_stitch_pool(args.stitch_workers) if args.decode_type == "chunk" else None   # before the GPU is touched
    be = Backend(int(os.environ.get("RD_CLI_DEVICE", local_rank)))   # override only for rehearsals on a 1-GPU box
    # every rank uses RCCL or none does (dist.connect); the rendezvous lives in the launcher's private scratch directory
    comm, kind = connect(be, rank, world, uid_path(directory=scratch), force_collective=True)
    # rank 0 parses / repacks the artefacts; everyone gets the device images by one broadcast (file transport: each
    # rank loads them itself).  The parent has validated them already, so rank 0 cannot fail here for a bad argument.
    comm.bcast_artifacts(be, lambda b: apply_artifacts(args, b, load_artifacts(args)))
    be.set_precision(args.precision)   # context state, not part of the broadcast images
    be.set_logits(args.logits)
    be.set_decode_math(args.decode_math)
    args._lm_loaded = (args.rna_model != "None" and args.decode_type == "global" and os.path.exists(args.rna_model))
    with open(os.path.join(scratch, "files.json")) as f:
        sources = [fast5.Fast5Source(p) for p in json.load(f)]
    run_rank(args, be, comm, scratch, sources, rank, world, stitch_pool=pool)
    comm.close()
    be.close()
    if pool is not None:
        pool.shutdown()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[4:])
    else:
        raise SystemExit("internal entry point; use python -m radian_amd.basecall ... --gpus N")
