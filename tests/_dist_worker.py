"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job running the product's sharded driver
(radian_amd.basecall.run + radian_amd.dist) against the oracle-backed test double."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    queue_block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    from radian_amd import basecall, dist, weights
    from _gloo_comm import GlooComm
    from _oracle_backend import OracleBackend
    from _reads import golden_reads
    rank, _, world = dist.env_rank_world()
    comm = GlooComm(rank, world, init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}")
    args = basecall.build_parser().parse_args(
        ["unused_in", "unused_out", "--chunk-len", "256", "--step-size", "128", "--beam-width", "4", "--decode-type", mode,
         "--gpu-batch-windows", "24", "--context-len", "3"])
    be = OracleBackend()

    def load(b):
        b.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2, 4)), (1, 2, 4))
        if mode == "global":
            b.load_lm(np.random.default_rng(9).dirichlet([0.3] * 4, size=64), 3)

    comm.bcast_artifacts(be, load)
    assert be.w is not None and be.dil == (1, 2, 4)
    args._lm_loaded = mode == "global"
    queue = dist.WorkQueue(os.path.join(out_dir, "queue"), queue_block) if queue_block else None
    res = basecall.run(args, be, reads=golden_reads(1500), writer=None, shard=(rank, world), queue=queue)
    t = comm.allreduce_max([float(rank)])
    assert t[0] == world - 1
    comm.barrier()
    with open(os.path.join(out_dir, f"rank{rank}.jsonl"), "w") as f:
        for r in res:
            f.write(json.dumps(list(r)) + "\n")
    comm.close()


if __name__ == "__main__":
    main()
