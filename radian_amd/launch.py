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


class StreamMerger:
    """Merges the per-rank result files into reads-{n}.fasta WHILE the ranks are still writing them, so that what is left
    to do after the last worker exits is the last block, not every read (the merge runs at ~2*10^5 records/s: 1 M reads
    merged afterwards would add seconds behind ~20 s of 8-GPU compute).

    A rank's file is a sequence of lines in the order things happen on that rank: {"claim": [file, lo, hi]} when it takes a
    block from the work queue, one line per finished read -- "R<key>\\t<read_id>\\t<sequence>", or the JSON list [key, read_id,
    sequence] for a read id that does not fit that form -- (increasing key: results leave the driver in input order), {"end": true} last.  A record with key m can be written as soon as no rank can still produce a smaller
    key: a rank with parsed-but-unwritten records is represented by the first of them; one without is bounded from below
    by the first key it may still produce -- the next expected read of its oldest unfinished claim (a claim is finished
    when a record of a later claim, or the end mark, shows up: skipped reads leave no record), else its latest key.
    One record per rank is all the merge needs in memory beyond what it has parsed and not yet been allowed to write.

    The FASTA files are written under a hidden directory inside fasta_dir and moved into place by finish(): a job that fails
    (one rank down stops all) leaves no valid-looking partial reads-*.fasta behind -- abort() removes what was written."""

    INF = (float("inf"),)

    def __init__(self, scratch, world, fasta_dir):
        from collections import deque
        from .basecall import FastaWriter
        self.paths = [os.path.join(scratch, f"rank{r}.jsonl") for r in range(world)]
        self.world = world
        self.fh = [None] * world
        self.tail = [b""] * world
        self.recs = [deque() for _ in range(world)]      # parsed, not yet written
        self.claims = [deque() for _ in range(world)]    # [file, next expected read, hi] of unfinished claims, oldest first
        self.last = [(-1,)] * world                      # latest key seen from the rank
        self.ended = [False] * world
        self.fasta_dir = fasta_dir
        self.partial = tempfile.mkdtemp(prefix=".radian_partial_", dir=fasta_dir)   # same filesystem: finish() renames
        self.writer = FastaWriter(self.partial)
        self.n = 0

    def _read(self, r):
        if self.fh[r] is None:
            try:
                self.fh[r] = open(self.paths[r], "rb")
            except OSError:
                return
        data = self.fh[r].read()
        if not data:
            return
        lines = (self.tail[r] + data).split(b"\n")
        self.tail[r] = lines.pop()
        for ln in lines:
            if not ln:
                continue
            if ln[:1] == b"R":     # a result in the compact form _RankFile writes: R <file>,<read> \t read_id \t sequence as the FASTA holds it
                kf, rid, seq = ln[1:].decode("ascii").split("\t")
                a, _, b = kf.partition(",")
                o = [(int(a), int(b)) if b else (int(a),), rid, seq, True]
            else:
                o = json.loads(ln)
            if isinstance(o, dict):
                if "claim" in o:
                    fi, lo, hi = o["claim"]
                    self.claims[r].append([fi, lo, hi])
                elif o.get("end"):
                    self.ended[r] = True
                    self.claims[r].clear()
                continue
            key = _key(o[0])
            cl = self.claims[r]
            while cl and not (len(key) == 2 and cl[0][0] == key[0] and cl[0][1] <= key[1] < cl[0][2]):
                cl.popleft()                                 # an older claim with nothing more to come
            if cl:
                cl[0][1] = key[1] + 1
            self.last[r] = key
            self.recs[r].append((key, o[1], o[2], len(o) > 3))

    def _bound(self, r):
        """no record that rank r has not handed over yet can have a key below this"""
        if self.recs[r]:
            return self.recs[r][0][0]
        if self.ended[r]:
            return self.INF
        cl = self.claims[r]
        while len(cl) > 1 and cl[0][1] >= cl[0][2]:
            cl.popleft()     # its last read has reported and a newer claim is known: the bound moves on to that claim's first read
        if cl:
            return (cl[0][0], cl[0][1])
        return self.last[r]

    def poll(self):
        for r in range(self.world):
            if not self.ended[r]:
                self._read(r)
        while True:
            bounds = [self._bound(r) for r in range(self.world)]
            best = None
            for r in range(self.world):
                if self.recs[r] and (best is None or bounds[r] < bounds[best]):
                    best = r
            if best is None:
                return
            limit = min((bounds[r] for r in range(self.world) if r != best), default=self.INF)
            if limit < bounds[best]:
                return
            # nothing the other ranks can still produce sorts before `limit`, and their state does not change while this rank's
            # records are written: hand over the whole run up to it (one bounds computation per run of a rank, not per record)
            q = self.recs[best]
            while q and not (limit < q[0][0]):
                _, rid, seq, final = q.popleft()
                if final:
                    self.writer.write_final(rid, seq)      # (the rank has reversed it already)
                else:
                    self.writer.write(rid, seq)
                self.n += 1

    def finish(self):
        """every worker has exited: read what is left, write it, close the FASTA.  -> number of records written"""
        try:
            for r in range(self.world):
                self._read(r)
                self.ended[r] = True      # a rank that died before its end mark has nothing more to say either
                self.claims[r].clear()
            self.poll()
            assert not any(self.recs), "internal: records left after the final merge pass"
        except BaseException:
            self.abort()
            raise
        self.close()
        for name in sorted(os.listdir(self.partial)):
            os.replace(os.path.join(self.partial, name), os.path.join(self.fasta_dir, name))
        os.rmdir(self.partial)
        return self.n

    def close(self):
        self.writer.close()
        for f in self.fh:
            if f is not None:
                f.close()

    def abort(self):
        """the job failed: close and delete what has been written so far"""
        self.close()
        shutil.rmtree(self.partial, ignore_errors=True)


def _key(k):
    return tuple(k) if isinstance(k, (list, tuple)) else (k,)


def merge_to_fasta(scratch, world, fasta_dir):
    """Per-rank result files (complete) -> reads-{n}.fasta in input order."""
    return StreamMerger(scratch, world, fasta_dir).finish()


def merge_watch(scratch, world, fasta_dir, timeout=600.0, poll=0.02):
    """The merging half of run_multi_gpu for a process that did not start the ranks itself (bench.py's N-rank files -> FASTA leg: the ranks
    are the launcher's, the merger is a child of rank 0 that never touches a GPU): merge while the ranks write, until every rank file
    carries its end mark; a rank that never ends makes this raise TimeoutError after `timeout` s with nothing left behind.
    -> {"records", "t_done"}"""
    m = StreamMerger(scratch, world, fasta_dir)
    t0 = time.time()
    try:
        while not all(m.ended):
            m.poll()
            if time.time() - t0 > timeout:
                raise TimeoutError(f"merge_watch: ranks {[r for r in range(world) if not m.ended[r]]} wrote no end mark within {timeout:.0f} s")
            time.sleep(poll)
    except BaseException:
        m.abort()
        raise
    n = m.finish()
    return {"records": n, "t_done": time.time()}


def wait_all(procs, poll=0.05, on_poll=None):
    """Wait for every worker; as soon as one exits non-zero, stop the others (they may be blocked in a collective that
    will never complete) and return the exit codes.  on_poll() runs once per poll interval while workers are alive."""
    while True:
        if on_poll is not None:
            on_poll()
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


def run_ranks(world, cmd, env_extra=None, capture_rank0=False, on_poll=None):
    """Start `cmd` once per rank (RANK / LOCAL_RANK / WORLD_SIZE in the environment; fresh processes -- the parent has not
    touched a GPU and never does), wait with wait_all's one-fails-all-stop rule.  capture_rank0: rank 0's stdout is
    collected and returned (the others' goes to the parent's stderr), for launchers whose rank 0 prints the job's result.
    on_poll(): called every poll while the workers run (the CLI's streaming merge).  -> (exit codes, rank 0's stdout)."""
    import threading
    procs, chunks = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(env_extra or {}))
        # numpy's BLAS starts a pool of one idle thread per core of the NODE at import, before the rank can bind itself to its slice
        # (hostbudget.py); nothing in a rank calls BLAS
        env.setdefault("OPENBLAS_NUM_THREADS", "1")
        env.setdefault("MKL_NUM_THREADS", "1")
        out = None
        if capture_rank0:
            out = subprocess.PIPE if rank == 0 else sys.stderr
        procs.append(subprocess.Popen(list(cmd), env=env, stdout=out))
    reader = None
    if capture_rank0:
        def pump():
            for line in procs[0].stdout:
                chunks.append(line)
        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
    rcs = wait_all(procs, on_poll=on_poll)
    if reader is not None:
        reader.join(timeout=5.0)
    return rcs, b"".join(chunks).decode("utf-8", "replace")


def package_env():
    """PYTHONPATH for child processes that must import this package whatever the working directory is: the reference is run from the directory
    that holds models/ (basecall.py:28-30), the top-level basecall.py finds the package beside itself through sys.path -- which a child process
    started with `python -m radian_amd.launch` does not inherit."""
    parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    old = os.environ.get("PYTHONPATH", "")
    return {"PYTHONPATH": parent + (os.pathsep + old if old else "")}


def run_multi_gpu(args, argv):
    from . import fast5
    from .basecall import load_artifacts, save_artifacts
    world = args.gpus
    art = load_artifacts(args)   # host-only validation: raises here, in the parent, exactly what a single-GPU run would raise
    scratch = tempfile.mkdtemp(prefix="radian_mgpu_")
    merger = None
    try:
        save_artifacts(art, scratch)   # (the rank that feeds the broadcast reads this back instead of parsing again)
        del art
        with open(os.path.join(scratch, "files.json"), "w") as f:
            json.dump(fast5.list_files(args.fast5_dir), f)   # one enumeration: every rank sees the same file order
        merger = StreamMerger(scratch, world, args.fasta_dir)
        cmd = [sys.executable, "-m", "radian_amd.launch", "--worker", scratch, "--"] + list(argv)
        rcs, _ = run_ranks(world, cmd, env_extra=package_env(), on_poll=merger.poll)   # the FASTA grows while the ranks work
        if any(rcs):
            raise SystemExit(f"multi-GPU run failed: worker exit codes {rcs}")
        n = merger.finish()
        merger = None
        ranks = []
        for r in range(world):
            with open(os.path.join(scratch, f"contexts{r}.json")) as f:
                ranks.append(json.load(f))
        return {"records": n, "ranks": ranks}   # what each rank ran with: device, device contexts, agreed transport
    finally:
        if merger is not None:
            merger.abort()
        shutil.rmtree(scratch, ignore_errors=True)


class _RankFile:
    """rank{r}.jsonl as StreamMerger reads it; flushed at claims and every 0.1 s so the parent can merge as the rank goes"""

    def __init__(self, path):
        import threading
        self.f = open(path, "w")
        self.t = time.time()
        self.lock = threading.Lock()   # claims come from the driver's reading thread, results from its host stage

    def claim(self, fi, lo, hi):
        with self.lock:
            self.f.write(json.dumps({"claim": [fi, lo, hi]}) + "\n")
            self.f.flush()

    def emit(self, key, rid, seq):
        with self.lock:
            if "\t" in rid or "\n" in rid or not rid.isascii():     # (not a fast5 read id; the JSON form carries anything)
                self.f.write(json.dumps([list(key) if isinstance(key, tuple) else key, rid, seq]) + "\n")
            else:   # compact form: 8 ranks x 27 M samples/s are ~53 k records/s for the merging parent to parse
                kf = ",".join(str(int(x)) for x in key) if isinstance(key, tuple) else str(int(key))
                self.f.write(f"R{kf}\t{rid}\t{seq[::-1]}\n")    # (reversed here, on the rank: 5' to 3' as basecall.py:129 writes it)
            now = time.time()
            if now - self.t > 0.1:
                self.f.flush()
                self.t = now

    def end(self):
        with self.lock:
            self.f.write(json.dumps({"end": True}) + "\n")
            self.f.close()


def run_rank(args, be, comm, scratch, sources, rank, world, stitch_pool=None, backends=None, stats=None):
    """One rank's share of the job, after the artefacts are on its device: claim work, basecall on `backends` (the device
    contexts of this rank's GPU; default: the one that received the broadcast), write rank{r}.jsonl as it goes, barrier
    (comm None: no barrier -- the caller synchronises its ranks by other means).  stats: see basecall.run."""
    from .basecall import run
    from .dist import FileReadQueue
    out = _RankFile(os.path.join(scratch, f"rank{rank}.jsonl"))
    queue = None
    if args.queue_block > 0:
        queue = FileReadQueue(os.path.join(scratch, "queue"), args.queue_block, on_claim=out.claim)
    try:
        run(args, backends or be, writer=None, shard=(rank, world), stitch_pool=stitch_pool, queue=queue, sources=sources, on_result=out.emit, stats=stats)
    finally:
        out.end()      # (a rank that fails still says that it is done: the merger must not wait for it)
        if queue is not None:
            queue.close()
    if comm is not None:
        comm.barrier()
    return queue


def rank_budget(args, local_rank, local_world):
    """Host budget of one rank, BEFORE any GPU call or thread (hostbudget.py holds the placement rule): bind to this rank's slice of the
    usable cores -- NUMA-local to its GPU when the topology is readable -- and size the threads that scale (the chunk-mode stitch) from
    the slice, never from the node: eight ranks on 64 cores run 8 x 6 stitch threads, not 8 x 16.  Fills args.stitch_workers when the
    user left it open.  -> the plan (cpus, how, bound, numa_node, cores_for_threads)."""
    from . import hostbudget
    # (RD_CLI_DEVICE puts every local rank on ONE device -- a rehearsal on a 1-GPU box: the ranks then share that GPU's NUMA node)
    devices = [int(os.environ["RD_CLI_DEVICE"])] * local_world if "RD_CLI_DEVICE" in os.environ else None
    budget = hostbudget.apply(local_rank, local_world, getattr(args, "cpu_affinity", "auto"), devices=devices)
    n_mine = len(budget["cpus"]) if (budget["bound"] or budget["how"] != "all") else max(1, len(budget["cpus"]) // max(1, local_world))
    budget["cores_for_threads"] = n_mine
    if args.stitch_workers is None and args.decode_type == "chunk":
        args.stitch_workers = (min(4, max(1, n_mine // 4)) if args.no_pipeline else hostbudget.threads_for(n_mine, "chunk")["stitch_threads"])
    return budget


def worker(scratch, argv):
    from . import fast5
    from .backend import Backend
    from .basecall import apply_artifacts, build_parser, load_artifacts, make_stitch_pool, n_contexts
    from .dist import StartupFailed, connect, env_rank_world, uid_path
    args = build_parser().parse_args(argv)
    rank, local_rank, world = env_rank_world()
    budget = rank_budget(args, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    pool = make_stitch_pool(args.stitch_workers) if args.decode_type == "chunk" and args.no_pipeline else None   # before the GPU is touched
    from .backend import device_for_rank
    device = int(os.environ["RD_CLI_DEVICE"]) if "RD_CLI_DEVICE" in os.environ else device_for_rank(local_rank)   # (override: rehearsals on a 1-GPU box)
    be = Backend(device)
    # every rank uses RCCL or none does (dist.connect); the rendezvous lives in the launcher's private scratch directory
    try:
        comm, kind = connect(be, rank, world, uid_path(directory=scratch), force_collective=True)
        # rank 0 parses / repacks the artefacts; everyone gets the device images by one broadcast (file transport: each
        # rank loads them itself).  The parent has validated them already, so rank 0 cannot fail here for a bad argument.
        comm.bcast_artifacts(be, lambda b: apply_artifacts(args, b, load_artifacts(args, cache_dir=scratch)))
    except StartupFailed as e:
        print(f"[radian_amd.launch] rank {rank}: {e}", file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)   # a helper thread is stuck inside a collective; the parent stops the other ranks
    # the rank's further device contexts (--device-contexts, as in a single-GPU run: one context's forward overlaps the
    # other's beam search) take the images from the one that received the broadcast -- a device copy, no second parse
    backends = [be] + [Backend(device) for _ in range(n_contexts(args) - 1)]
    for b in backends[1:]:
        b.clone_artifacts_from(be)
    for b in backends:
        b.set_precision(args.precision)   # context state, not part of the broadcast images
        b.set_logits(args.logits)
        b.set_decode_math(args.decode_math)
        b.set_decode_partition(args.decode_partition)
    args._lm_loaded = (args.rna_model != "None" and args.decode_type == "global" and os.path.exists(args.rna_model))
    with open(os.path.join(scratch, "files.json")) as f:
        sources = [fast5.Fast5Source(p) for p in json.load(f)]
    with open(os.path.join(scratch, f"contexts{rank}.json"), "w") as f:
        json.dump({"device": device, "contexts": len(backends), "transport": kind, "cpus": budget["cpus"], "cpu_split": budget["how"],
                   "cpu_bound": budget["bound"], "numa_node": budget["numa_node"], "stitch_workers": args.stitch_workers}, f)   # (what the worker-route tests read)
    run_rank(args, be, comm, scratch, sources, rank, world, stitch_pool=pool, backends=backends)
    comm.close()
    for b in reversed(backends):
        b.close()
    if pool is not None:
        pool.shutdown()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[4:])
    elif len(sys.argv) >= 6 and sys.argv[1] == "--merge":     # --merge scratch world fasta_dir timeout: merge_watch as a process of its own
        print(json.dumps(merge_watch(sys.argv[2], int(sys.argv[3]), sys.argv[4], float(sys.argv[5]))))
    else:
        raise SystemExit("internal entry point; use python -m radian_amd.basecall ... --gpus N")
