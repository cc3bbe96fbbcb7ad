"""Worker of tests/test_dist_cpu.py::test_world8_*: one rank of the multi-GPU launcher's worker route
(radian_amd.launch.run_rank: file-then-read work queue, per-rank result file, barrier) with a trivial stand-in for the
device so that >= 10 k reads go through the host logic in seconds on CPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


class TrivialBackend:
    """labels = first samples mod 4: enough to tell every read apart; status 1 for an all-equal signal"""

    def basecall_raw_global(self, raws, outlier_clip, chunk_len, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        status = np.array([1 if len(r) and (np.asarray(r) == r[0]).all() else 0 for r in raws], dtype=np.int32)
        return [(np.asarray(r[:16]) % 4).astype(np.uint8) for r in raws], status


def main():
    scratch = sys.argv[1]
    slow = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    from radian_amd import basecall, dist, fast5, launch
    rank, _, world = dist.env_rank_world()
    args = basecall.build_parser().parse_args(["in", "out", "--decode-type", "global", "--rna-model", "None", "--queue-block", "96",
                                               "--gpu-batch-windows", "64"])
    args._lm_loaded = False
    comm = dist.FileComm(rank, world, os.path.join(scratch, "fc"), timeout=120.0)
    with open(os.path.join(scratch, "files.json")) as f:
        sources = [fast5.Fast5Source(p) for p in json.load(f)]
    be = TrivialBackend()
    if slow and rank % 2:
        orig = be.basecall_raw_global

        def slow_call(*a, **k):
            time.sleep(slow)
            return orig(*a, **k)
        be.basecall_raw_global = slow_call
    with open(os.devnull, "w") as dn:
        so, sys.stdout = sys.stdout, dn
        try:
            q = launch.run_rank(args, be, comm, scratch, sources, rank, world)
        finally:
            sys.stdout = so
    comm.close()
    json.dump({"claimed": q.claimed, "opened": sorted(q.opened), "prefetched": sorted(q.prefetched)}, open(os.path.join(scratch, f"queue{rank}.json"), "w"))


if __name__ == "__main__":
    main()
