"""Stand-in for one rank of bench.py's own launcher (tests/test_dist_cpu.py::test_bench_self_launch_*): goes through
the product's start-up (radian_amd.dist.connect with a device-less Backend stand-in, the launcher's rendezvous directory)
and prints what a rank prints; argv[1] = 'ok' | 'fail1' (rank 1 exits 3 while the others would wait forever)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NoDevice:
    def rccl_unique_id(self):
        raise RuntimeError("no librccl in the stub")


def main():
    from radian_amd import dist
    mode = sys.argv[1]
    rank, _, world = dist.env_rank_world()
    assert "RD_BENCH_RDV" in os.environ and os.path.isdir(os.environ["RD_BENCH_RDV"])
    if mode == "fail1":
        if rank == 1:
            time.sleep(0.3)
            sys.exit(3)
        time.sleep(600)
    comm, kind = dist.connect(NoDevice(), rank, world, dist.uid_path(directory=os.environ["RD_BENCH_RDV"]), timeout=60)
    assert kind == "file-fallback"
    n = comm.nranks_seen()
    per_rank = comm.allgather(1.0 + rank)
    comm.barrier()
    print(f"rank {rank} chatter")           # rank 0: to the launcher's pipe (must not become the result); others: stderr
    if rank == 0:
        print(json.dumps({"metric": "stub", "n_gpus": world, "rccl_nranks": n, "ms_per_step_per_rank": per_rank, "startup_comm": kind}))
    comm.close()


if __name__ == "__main__":
    main()
