"""torch.distributed `gloo` transport with radian_amd.dist's comm interface (barrier, allreduce_max, bcast_artifacts,
close).  TESTS ONLY: lets the world_size-N CPU tests drive the product's sharded driver through a real collective
library; the product itself never imports torch (RCCL through libradian_hip.so, or the file transport)."""
import numpy as np


class GlooComm:
    """artefacts travel as host bytes and every rank loads them itself."""

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

    def set_precision(self, *a, **kw):
        pass
