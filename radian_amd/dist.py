"""Process-group plumbing for the multi-GPU path: one process per GPU, reads sharded by index, ONE start-up
broadcast of the artefacts, no data-path collective.

Two interchangeable transports behind the same four calls (barrier, allreduce_max, bcast_artifacts, close):
  * RcclComm  -- RCCL over xGMI through libradian_hip.so's rd_rccl_* (the GPU path; no PyTorch in the process);
  * GlooComm  -- torch.distributed `gloo` on CPU tensors (used by the CPU tests to exercise the same
                 sharding / merge logic with world_size 2, and usable on hosts without RCCL).
"""
import os
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


def uid_path(tag=None):
    """Rendezvous file for the RCCL unique id.  All ranks of one launch are children of the same launcher process
    (torch.distributed.run's agent, or radian_amd.launch), so its pid + start time names the launch uniquely and a
    file left behind by an earlier, crashed launch can never be picked up."""
    if tag is None:
        ppid = os.getppid()
        tag = f"{os.environ.get('MASTER_PORT', '29500')}_{ppid}_{_proc_start_time(ppid)}"
    return f"/tmp/radian_rccl_uid_{tag}"


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

    def close(self):
        pass


class RcclComm:
    """RCCL communicator owned by the Backend's rd_ctx."""

    def __init__(self, be, rank, world, uid_file):
        self.be, self.rank, self.world, self._uid_file = be, rank, world, uid_file
        uid = exchange_uid(be.rccl_unique_id, rank, uid_file)
        be.rccl_init(rank, world, uid)

    def barrier(self):
        self.be.rccl_barrier()

    def allreduce_max(self, values):
        return self.be.rccl_allreduce_max(values)

    def bcast_artifacts(self, be, load_fn):
        """rank 0 loads + repacks weights / LM, then one broadcast puts the device images on every rank."""
        if self.rank == 0:
            load_fn(be)
        be.rccl_bcast_model(0)

    def close(self):
        if self.rank == 0 and os.path.exists(self._uid_file):
            os.remove(self._uid_file)


class FileComm:
    """Last-resort transport through a shared directory on one node (no collective library): used by bench.py only if
    the RCCL communicator cannot be created, so that a scaling run still reports.  Every rank loads the artefacts itself."""

    def __init__(self, rank, world, base):
        self.rank, self.world, self._base, self._seq = rank, world, base, 0

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
                if time.time() - t0 > 600:
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

    def close(self):
        self.barrier()
        # once everyone has passed exchange k, every file of exchanges < k has been read by all ranks; the files of the
        # last exchange stay (a few bytes, named after the launcher's pid + start time, never reused)
        for q in range(1, self._seq):
            try:
                os.remove(f"{self._base}.x{q}.{self.rank}")
            except OSError:
                pass


class GlooComm:
    """torch.distributed gloo; artefacts travel as host bytes and every rank loads them itself."""

    def __init__(self, rank, world, init_method=None):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = rank, world
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world, init_method=init_method)

    def barrier(self):
        self._dist.barrier()

    def allreduce_max(self, values):
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return t.numpy()

    def bcast_artifacts(self, be, load_fn):
        """load_fn(be) must call be.load_weights(flat, dilations) / be.load_lm(table, k); rank 0 runs it against a
        recorder, the recorded host arrays are broadcast, every rank replays them into its own backend."""
        rec = _Recorder()
        if self.rank == 0:
            load_fn(rec)
        box = [rec.calls if self.rank == 0 else None]
        self._dist.broadcast_object_list(box, src=0)
        for name, a, kw in box[0]:
            getattr(be, name)(*a, **kw)

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


class _Recorder:
    def __init__(self):
        self.calls = []

    def load_weights(self, *a, **kw):
        self.calls.append(("load_weights", a, kw))

    def load_lm(self, *a, **kw):
        self.calls.append(("load_lm", a, kw))


class WorkQueue:
    """Per-node work queue over read indices ("per-rank work queue over fast5 files/reads for real input").
    The ranks of one node share a counter file; a rank claims the next block of `block` consecutive read indices by
    advancing the counter under an fcntl lock, so a rank with long reads simply claims fewer blocks.  Every rank walks
    the same read sequence and asks owns(idx) in increasing idx order; results are merged by index (merge_results)."""

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


def shard_indices(n, rank, world):
    """Round-robin shard of read indices (what basecall.run uses: index % world == rank)."""
    return list(range(rank, n, world))


def merge_results(per_rank_results):
    """[(read_index, read_id, sequence)] lists from every rank -> one list in input order."""
    merged = [r for rr in per_rank_results for r in rr]
    merged.sort(key=lambda r: r[0])
    return merged
