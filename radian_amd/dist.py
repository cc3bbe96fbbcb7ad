"""Process-group plumbing for the multi-GPU path: one process per GPU, reads sharded over the ranks, ONE start-up
broadcast of the artefacts, no data-path collective.

Transports behind the same four calls (barrier, allreduce_max, bcast_artifacts, close):
  * RcclComm  -- RCCL over xGMI through libradian_hip.so's rd_rccl_* (the GPU path; no PyTorch in the process);
  * FileComm  -- files in a directory shared by the ranks of one node (no collective library): the agreed fallback when
                 RCCL cannot be used by EVERY rank.
`connect()` makes the choice collectively: every rank reports whether its RCCL start-up worked and RCCL is used only if
all of them did -- a rank never sits in ncclBroadcast while another waits at a file barrier.
(The torch.distributed `gloo` transport used by the CPU tests lives in tests/_gloo_comm.py, not in the product.)
"""
import heapq
import json
import os
import tempfile
import threading
import time

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _proc_start_time(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time in clock ticks since boot
    except Exception:
        return "0"


def uid_path(tag=None, directory=None):
    """Rendezvous path for one launch.  With `directory` (the launcher's private mkdtemp scratch: radian_amd.launch and
    bench.py's own launcher) the name is unique by construction.  Otherwise (a foreign launcher such as
    torch.distributed.run): all ranks are children of the same launcher process, so its pid + start time names the
    launch, and the launcher's restart counter names the attempt -- a worker group restarted by the same agent never
    reads the files (unique id, init outcomes) of the attempt before it."""
    if directory is not None:
        return os.path.join(directory, "rccl_uid")
    if tag is None:
        ppid = os.getppid()
        attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        tag = f"{os.environ.get('MASTER_PORT', '29500')}_{ppid}_{_proc_start_time(ppid)}_a{attempt}"
    return os.path.join(tempfile.gettempdir(), f"radian_rccl_uid_{tag}")


class StartupFailed(RuntimeError):
    """The collective start-up cannot complete (a peer reported a failure while this rank is still inside
    ncclCommInitRank, which cannot be cancelled): the process should exit non-zero NOW so that its launcher stops the job
    (launch.wait_all / torch.distributed.run both do).  `stuck` tells the caller that a helper thread is still blocked in
    the RCCL call: leave with os._exit, not through interpreter shutdown."""

    def __init__(self, msg, stuck=False):
        super().__init__(msg)
        self.stuck = stuck


class Rendezvous:
    """All-gather of one short string per rank through files in a directory (atomic rename, polling)."""

    def __init__(self, directory, rank, world, timeout=120.0):
        self.dir, self.rank, self.world, self.timeout = directory, rank, world, timeout
        os.makedirs(directory, exist_ok=True)

    def publish(self, phase, value):
        mine = os.path.join(self.dir, f"{phase}.{self.rank}")
        with open(mine + ".tmp", "w") as f:
            f.write(value)
        os.replace(mine + ".tmp", mine)

    def peek(self, phase):
        """{rank: value} of what has been published for `phase` so far (no waiting)"""
        out = {}
        for r in range(self.world):
            try:
                with open(os.path.join(self.dir, f"{phase}.{r}")) as f:
                    out[r] = f.read()
            except OSError:
                pass
        return out

    def gather(self, phase, value):
        self.publish(phase, value)
        out = []
        t0 = time.time()
        for r in range(self.world):
            p = os.path.join(self.dir, f"{phase}.{r}")
            while not os.path.exists(p):
                if time.time() - t0 > self.timeout:
                    raise RuntimeError(f"rendezvous '{phase}': timed out after {self.timeout:.0f}s waiting for rank {r}")
                time.sleep(0.002)
            with open(p) as f:
                out.append(f.read())
        return out


def exchange_uid(make_uid, rank, path, timeout=120.0):
    """rank 0 publishes the 128-byte RCCL unique id through a file (atomic rename); the others poll for it."""
    if rank == 0:
        uid = make_uid()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("timed out waiting for the RCCL unique id from rank 0")
        time.sleep(0.01)
    with open(path, "rb") as f:
        return f.read()


class SingleComm:
    rank, world = 0, 1

    def barrier(self):
        pass

    def allreduce_max(self, values):
        return np.asarray(values, dtype=np.float64)

    def bcast_artifacts(self, be, load_fn):
        load_fn(be)

    def allgather(self, value):
        return [float(value)]

    def nranks_seen(self):
        return 1

    def close(self):
        pass


def _allgather_by_max(comm, value):
    """every rank's non-negative value, by one max-reduction of a one-hot vector (the comms only offer max)"""
    v = np.zeros(comm.world, dtype=np.float64)
    v[comm.rank] = float(value)
    return [float(x) for x in comm.allreduce_max(v)]


class RcclComm:
    """RCCL communicator owned by the Backend's rd_ctx (created by connect(): every rank has one, or none has)."""

    def __init__(self, be, rank, world, rdv=None):
        self.be, self.rank, self.world, self.rdv = be, rank, world, rdv

    def barrier(self):
        self.be.rccl_barrier()

    def allreduce_max(self, values):
        return self.be.rccl_allreduce_max(values)

    def bcast_artifacts(self, be, load_fn, timeout=600.0, load_timeout=3600.0):
        """rank 0 loads + repacks weights / LM, then one broadcast puts the device images on every rank.  Two watched phases:
        (1) the load: rank 0 publishes "loaded" (ok / fail) through the rendezvous when load_fn has returned; the others wait for that
        mark under `load_timeout` WITHOUT entering the collective -- a slow parse on rank 0 (the 420 MB JSON through the fallback parser
        on a cold filesystem) does not eat the broadcast's deadline, and a load that fails leaves nobody inside a collective;
        (2) the broadcast, on a helper thread under `timeout` counted from the mark: a rank that a peer left alone inside the
        collective raises StartupFailed(stuck=True) instead of waiting for ever; an error of this rank's own call is re-raised as it is."""
        if self.rank == 0:
            try:
                load_fn(be)
            except BaseException as e:
                if self.rdv is not None:
                    self.rdv.publish("loaded", f"fail: {type(e).__name__}: {e}")
                raise
            if self.rdv is not None:
                self.rdv.publish("loaded", "ok")
        elif self.rdv is not None:
            t0 = time.time()
            while True:
                mark = self.rdv.peek("loaded").get(0)
                if mark is not None:
                    break
                if time.time() - t0 > load_timeout:
                    raise StartupFailed(f"rank {self.rank}: rank 0 did not finish loading the artefacts within {load_timeout:.0f} s")
                time.sleep(0.01)
            if mark.startswith("fail"):
                raise StartupFailed(f"rank {self.rank}: rank 0 failed to load the artefacts ({mark[5:].strip()}); nothing to receive")
        box = {}

        def run():
            try:
                be.rccl_bcast_model(0)
            except BaseException as e:
                box["err"] = e

        th = threading.Thread(target=run, name="rccl-bcast", daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            raise StartupFailed(f"rank {self.rank}: the artefact broadcast did not finish within {timeout:.0f} s", stuck=True)
        if "err" in box:
            raise box["err"]

    def allgather(self, value):
        return _allgather_by_max(self, value)

    def nranks_seen(self):
        """the communicator's size as RCCL reports it (ncclCommCount)"""
        return self.be.rccl_comm_count()

    def close(self):
        pass


class FileComm:
    """Transport through a shared directory on one node (no collective library).  Every rank loads the artefacts itself."""

    def __init__(self, rank, world, base, timeout=600.0):
        self.rank, self.world, self._base, self._seq, self._timeout = rank, world, base, 0, timeout

    def _exchange(self, value):
        self._seq += 1
        mine = f"{self._base}.x{self._seq}.{self.rank}"
        with open(mine + ".tmp", "w") as f:
            f.write(repr(float(value)))
        os.replace(mine + ".tmp", mine)
        vals = []
        t0 = time.time()
        for r in range(self.world):
            p = f"{self._base}.x{self._seq}.{r}"
            while not os.path.exists(p):
                if time.time() - t0 > self._timeout:
                    raise RuntimeError("FileComm: timed out waiting for rank %d" % r)
                time.sleep(0.0005)
            with open(p) as f:
                vals.append(float(f.read()))
        return vals

    def barrier(self):
        self._exchange(0.0)

    def allreduce_max(self, values):
        return np.asarray([max(self._exchange(v)) for v in np.asarray(values, dtype=np.float64).ravel()])

    def bcast_artifacts(self, be, load_fn):
        load_fn(be)

    def allgather(self, value):
        return self._exchange(value)

    def nranks_seen(self):
        """ranks that answered the latest exchange (there is no communicator to ask)"""
        return len(self._exchange(0.0))

    def close(self):
        self.barrier()
        # once everyone has passed exchange k, every file of exchanges < k has been read by all ranks; the files of the
        # last exchange stay (a few bytes, named after the launch, never reused)
        for q in range(1, self._seq):
            try:
                os.remove(f"{self._base}.x{q}.{self.rank}")
            except OSError:
                pass


def _run_watched(fn, rdv, phase, what, rank, timeout, stuck_grace):
    """Run one collective start-up step on a helper thread while this thread watches the rendezvous -> 'ok' | 'fail: ...'.
    A collective cannot be cancelled: when a peer has published a failure for this phase (or the time is up) while fn is still
    running, StartupFailed(stuck=True) is raised -- the caller leaves the process without interpreter shutdown."""
    box = {}

    def run():
        try:
            fn()
            box["out"] = "ok"
        except Exception as e:
            box["out"] = f"fail: {e}"

    th = threading.Thread(target=run, name=f"rccl-{phase}", daemon=True)
    th.start()
    t0, seen_fail = time.time(), None
    while th.is_alive():
        th.join(0.02)
        if not th.is_alive():
            break
        now = time.time()
        failed = {r: v for r, v in rdv.peek(phase).items() if v.startswith("fail")}
        if failed and seen_fail is None:
            seen_fail = now
        if seen_fail is not None and now - seen_fail > stuck_grace:
            raise StartupFailed("RCCL start-up failed on " + "; ".join(f"rank {r}: {v[5:].strip()}" for r, v in sorted(failed.items()))
                                + f"; rank {rank} is still inside {what} and cannot leave it", stuck=True)
        if now - t0 > timeout:
            raise StartupFailed(f"rank {rank}: {what} did not return within {timeout:.0f} s", stuck=True)
    return box["out"]


def connect(be, rank, world, uid_file, allow_file_fallback=True, timeout=120.0, force_collective=False, stuck_grace=10.0):
    """Collective choice of the transport -> (comm, kind) with kind 'rccl' | 'file-fallback' | 'single'.

    Phase 1 (nothing collective has been called yet): every rank checks that librccl loads (rank 0 also draws the unique
    id) and publishes ok / fail; if anyone failed, NO rank calls ncclCommInitRank and all use the file transport.
    Phase 2: ncclCommInitRank runs on a helper thread while the rank's main thread watches the rendezvous.  Outcomes:
      * every rank's init returned ok                      -> RCCL everywhere;
      * every rank's init RETURNED and some failed         -> everyone drops its communicator, file transport everywhere
        (this is what two ranks on one GPU do: RCCL refuses the duplicate device on both sides at once);
      * a peer published a failure (or vanished) while this rank is STILL inside ncclCommInitRank -- the collective cannot
        complete and cannot be cancelled -> StartupFailed after `stuck_grace` seconds: fail fast, the launcher stops the job.
    Phase 3: the communicator's first collective (a max-reduction of the rank numbers, result checked), watched the same way
    and with the same three outcomes: RCCL builds its transports lazily, so this is where a broken one shows.
    With allow_file_fallback=False every failure raises instead (the launcher reports and stops the job)."""
    if world == 1 and not force_collective:   # (force_collective: a one-rank RCCL communicator, for the worker-route test)
        return SingleComm(), "single"
    rdv = Rendezvous(uid_file + ".rdv", rank, world, timeout)
    uid, mine = None, "ok"
    try:
        if rank == 0 or not hasattr(be, "rccl_probe"):
            uid = be.rccl_unique_id()      # dlopen(librccl) + ncclGetUniqueId (starts the bootstrap root: rank 0 only)
        else:
            be.rccl_probe()                # dlopen(librccl) + symbols: a purely local test
    except Exception as e:
        mine = f"fail: {e}"
    if rank == 0 and uid is not None:
        mine = "ok:" + uid.hex()
    phase1 = rdv.gather("pre", mine)
    errors = [f"rank {r}: {v[5:].strip()}" for r, v in enumerate(phase1) if v.startswith("fail")]
    if not errors:
        uid0 = bytes.fromhex(phase1[0][3:])

        def first_collective():
            # RCCL sets its transports up lazily: a communicator that initialised can still fail (or answer wrongly) in its
            # first collective.  One max-reduction of the rank numbers, watched like the init, before anything depends on it.
            got = be.rccl_allreduce_max([float(rank)])
            if int(got[0]) != world - 1:
                raise RuntimeError(f"first all-reduce returned {got[0]!r}, expected {world - 1}")

        for phase, what, fn in (("init", "ncclCommInitRank", lambda: be.rccl_init(rank, world, uid0)),
                                ("first", "the first collective", first_collective)):
            mine = _run_watched(fn, rdv, phase, what, rank, timeout, stuck_grace)
            outcome = rdv.gather(phase, mine)
            errors = [f"rank {r}: {v[5:].strip()}" for r, v in enumerate(outcome) if v.startswith("fail")]
            if errors:
                if phase == "first" or mine == "ok":    # this rank holds a communicator: give it back
                    try:
                        be.rccl_finalize()
                    except Exception:
                        pass
                break
        if not errors:
            return RcclComm(be, rank, world, rdv), "rccl"
    msg = "RCCL start-up failed on " + "; ".join(errors)
    if not allow_file_fallback:
        raise RuntimeError(msg)
    if rank == 0:
        import sys
        print(f"[radian_amd.dist] {msg}; all {world} ranks use the file transport and load the artefacts themselves", file=sys.stderr)
    return FileComm(rank, world, uid_file + ".fc"), "file-fallback"


def leave(rank, world, uid_file, timeout=10.0):
    """After comm.close(): remove the launch's rendezvous files when nobody needs them any more (a foreign launcher has no
    scratch directory to delete).  Every rank says goodbye; rank 0 waits for the others (bounded), then deletes."""
    import glob
    import shutil
    d = uid_file + ".rdv"
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"bye.{rank}"), "w"):
            pass
        if rank != 0:
            return
        t0 = time.time()
        while time.time() - t0 < timeout and not all(os.path.exists(os.path.join(d, f"bye.{r}")) for r in range(world)):
            time.sleep(0.005)
        shutil.rmtree(d, ignore_errors=True)
        for p in glob.glob(glob.escape(uid_file) + "*"):
            try:
                os.remove(p)
            except OSError:
                pass
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------ work distribution
class WorkQueue:
    """Per-node work queue over read INDICES: the ranks share a counter file; a rank claims the next block of `block`
    consecutive read indices by advancing the counter under a lock.  Every rank walks the same read sequence and asks
    owns(idx) in increasing idx order.  Used when the reads come from an in-memory sequence; for fast5 directories the
    file-level FileReadQueue below is used, so that a rank only opens the files it claimed from."""

    def __init__(self, path, block=256):
        import fcntl
        self._fcntl = fcntl
        self.path, self.block = path, int(block)
        if self.block < 1:
            raise ValueError("work-queue block must be >= 1")
        self._fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
        self._start = self._stop = 0
        self.claimed = []

    def _claim(self):
        f = self._fcntl
        f.flock(self._fd, f.LOCK_EX)
        try:
            os.lseek(self._fd, 0, os.SEEK_SET)
            raw = os.read(self._fd, 32)
            k = int(raw) if raw.strip() else 0
            os.lseek(self._fd, 0, os.SEEK_SET)
            os.write(self._fd, b"%-31d\n" % (k + 1))
        finally:
            f.flock(self._fd, f.LOCK_UN)
        self._start, self._stop = k * self.block, (k + 1) * self.block
        self.claimed.append(k)

    def owns(self, idx):
        """True iff this rank basecalls read `idx`; idx must not decrease between calls."""
        while idx >= self._stop:
            self._claim()
        return idx >= self._start

    def close(self):
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None


class FileReadQueue:
    """Per-node work queue over FILES, THEN READS ("per-rank work queue over fast5 files/reads"; the
    reference's loop nest is files -> reads, basecall.py:70-72).  Shared state, under an fcntl lock: (file index, first
    unclaimed read of that file).  A claim takes the next `block` reads of the current file, or what is left of it, and
    moves the cursor to the next file when the file is used up.  Only the claimant needs the file's read count, so a
    rank opens exactly the files it takes work from; claims come out in increasing (file, read) order."""

    def __init__(self, path, block=256, on_claim=None):
        import fcntl
        self._fcntl = fcntl
        self.block = int(block)
        if self.block < 1:
            raise ValueError("work-queue block must be >= 1")
        self._fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
        self._path = path
        self._counts = {}
        self.claimed = []          # [(file, lo, hi)]
        self.opened = set()        # files whose read count this rank looked up
        self.prefetched = set()    # ... of which ahead of time (the file after one whose first block this rank took)
        self.on_claim = on_claim   # on_claim(file, lo, hi) as soon as a block is this rank's (the launcher's streaming merge)

    # Read counts are looked up OUTSIDE the queue's lock (an HDF5 open + key listing) and shared between the ranks through
    # small files beside the queue file, so that a file is counted once per node, by the rank that gets there first; the
    # rank that takes the FIRST block of a file also counts the next file ahead of time, so that at a file boundary the
    # count is already there and nobody waits.  (the lookup used to sit inside the lock.)
    def _count_path(self, fi):
        return f"{self._path}.count.{fi}"

    def _known(self, fi):
        n = self._counts.get(fi)
        if n is None:
            try:
                with open(self._count_path(fi)) as f:
                    n = self._counts[fi] = int(f.read())
            except (OSError, ValueError):
                return None
        return n

    def _lookup(self, sources, fi, keep_open):
        if self._known(fi) is not None:
            return
        # one rank counts, the others that need the same count wait on the FILE's own lock (not on the queue's)
        lk = os.open(self._count_path(fi) + ".lock", os.O_RDWR | os.O_CREAT, 0o600)
        try:
            self._fcntl.flock(lk, self._fcntl.LOCK_EX)
            if self._known(fi) is not None:
                return
            self.opened.add(fi)
            n = self._counts[fi] = sources[fi].n_reads()
            if not keep_open or n == 0:
                sources[fi].close()          # counted ahead of time (or empty): whoever claims from it opens it again
            tmp = f"{self._count_path(fi)}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                f.write(str(n))
            os.replace(tmp, self._count_path(fi))
        finally:
            os.close(lk)                     # (closing the descriptor releases the lock)

    def claims(self, sources):
        """Generator of (file_index, lo, hi) blocks owned by this rank; sources[i].n_reads() gives a file's read count."""
        f = self._fcntl
        while True:
            need = None
            f.flock(self._fd, f.LOCK_EX)
            try:
                os.lseek(self._fd, 0, os.SEEK_SET)
                raw = os.read(self._fd, 64).split()
                fi, r0 = (int(raw[0]), int(raw[1])) if len(raw) >= 2 else (0, 0)
                got = None
                while fi < len(sources):
                    n = self._known(fi)
                    if n is None:
                        need = fi
                        break
                    if r0 >= n:
                        fi, r0 = fi + 1, 0
                        continue
                    hi = min(n, r0 + self.block)
                    got = (fi, r0, hi)
                    fi, r0 = (fi + 1, 0) if hi == n else (fi, hi)
                    break
                os.lseek(self._fd, 0, os.SEEK_SET)
                os.write(self._fd, b"%-31d %-31d\n" % (fi, r0))
            finally:
                f.flock(self._fd, f.LOCK_UN)
            if need is not None:
                self._lookup(sources, need, keep_open=True)
                continue
            if got is None:
                return
            self.claimed.append(got)
            if self.on_claim is not None:
                self.on_claim(*got)
            if got[1] == 0 and got[0] + 1 < len(sources) and self._known(got[0] + 1) is None:
                self.prefetched.add(got[0] + 1)
                self._lookup(sources, got[0] + 1, keep_open=False)
            yield got

    def close(self):
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None


def shard_indices(n, rank, world):
    """Round-robin shard of read indices (what basecall.run uses without a queue: index % world == rank)."""
    return list(range(rank, n, world))


def merge_results(per_rank_results):
    """[(key, read_id, sequence)] lists from every rank -> one list in input order (small jobs / tests)."""
    merged = [r for rr in per_rank_results for r in rr]
    merged.sort(key=lambda r: _key(r[0]))
    return merged


def _key(k):
    return tuple(k) if isinstance(k, (list, tuple)) else (k,)


def iter_results_file(path):
    """records of a rank's result file; the launcher's claim / end marks (JSON objects) are not records"""
    with open(path) as f:
        for line in f:
            o = json.loads(line)
            if isinstance(o, list):
                yield o


def merge_result_files(paths):
    """Streaming k-way merge of per-rank result files (JSON lines [key, read_id, sequence], each file in increasing key
    order -- a rank works through its claims in order) into input order, one record in memory per rank."""
    return heapq.merge(*[iter_results_file(p) for p in paths], key=lambda r: _key(r[0]))
