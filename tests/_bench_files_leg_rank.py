"""One rank of bench.py's N-rank files -> FASTA leg with the oracle-computed Backend stand-in (tests/test_dist_cpu.py): the product's
host side -- work queue, native fast5 reader, driver loop, rank files, merger process -- runs exactly as on the GPU box; the device calls
are the test double's.  argv[1] = 'ok' | 'fail1' (rank 1 raises inside its leg) ; argv[2] = where rank 0 keeps the merged FASTA."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CLI = ["--decode-type", "chunk", "--chunk-len", "128", "--step-size", "64", "--beam-width", "3", "--rna-model", "None", "--no-pipeline",
       "--device-contexts", "1", "--gpu-batch-windows", "64", "--queue-block", "7"]


def main():
    import bench
    from radian_amd import dist, hostbudget, weights
    from _oracle_backend import OracleBackend
    mode, keep = sys.argv[1], sys.argv[2]
    rank, local_rank, world = dist.env_rank_world()
    budget = hostbudget.apply(local_rank, world)
    src = OracleBackend()
    src.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2)), (1, 2))

    def factory():
        if mode == "fail1" and rank == 1:
            raise RuntimeError("rank 1 cannot make its context")
        return OracleBackend()
    leg = bench.files_leg(rank, world, src, budget, os.environ["RD_BENCH_RDV"], CLI, files_per_rank=2, reads_per_file=20, read_len=500, warm_reads=4,
                          timeout=40.0, backend_factory=factory, comm_info={"startup_comm": "stub", "rccl_nranks": world}, keep=keep if rank == 0 else None)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "secondary_e2e_fast5_to_fasta": leg}))


if __name__ == "__main__":
    main()
