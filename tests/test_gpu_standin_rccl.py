"""The product's N > 1 code with N > 1 on the one GPU that is reachable.

Real RCCL refuses two ranks on one device, so on a one-GPU box tests/test_gpu_two_ranks.py can only show the ranks agreeing on the
file transport.  Here the ranks find tests/standin_rccl.cpp (TEST INFRASTRUCTURE: the seven RCCL entry points the product resolves,
with rccl.h's signatures and semantics, over a shared-memory segment) as `librccl.so.1` through LD_LIBRARY_PATH.  Everything on the
product's side of the seam then runs as it will on the 8-GPU node: rd_rccl_unique_id on rank 0, the three-phase transport agreement
(load, ncclCommInitRank, the watched first collective), rd_rccl_bcast_model with rank 0 as the sender and the OTHER ranks executing the
receiver's half (header -> storage reserved and bound -> weight images -> LM table, entropies, sparse mask), the barriers and the
max-reductions of the timed region, ncclCommCount -- and the receivers then basecall with what they received:
  * `python bench.py --gpus 4`: one JSON line, startup_comm == "rccl", rccl_nranks == 4, every rank's step time;
  * `python -m radian_amd.basecall ... --gpus 4`, chunk mode and global mode with an RNA model (dense, and sparse: the "absent" mask is
    part of the broadcast image): FASTA identical to the single-process run, every worker reporting the rccl transport.
What this does NOT show is RCCL itself (its transports over xGMI): that needs the 8-GPU node.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def standin_env(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is needed to build the stand-in library")
    d = tmp_path_factory.mktemp("standin_rccl")
    lib = d / "librccl.so.1"
    r = subprocess.run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include", "-o", str(lib), os.path.join(ROOT, "tests", "standin_rccl.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and lib.exists(), r.stderr.decode()[-3000:]
    env = dict(os.environ, PYTHONPATH=ROOT, LD_LIBRARY_PATH=str(d) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("RD_BENCH_DEVICE", None)     # ranks map their local rank onto the one visible device themselves (backend.device_for_rank)
    env.pop("RD_CLI_DEVICE", None)
    return env


# Ranks per rehearsal.  The 8-GPU node runs eight; a GPU box of this pool allows SIX processes on its card at once (gpurun's process guard,
# which ends the whole run beyond that) and the pytest process itself holds a context: five ranks ran green in round 5, FOUR is what the
# suite starts -- one process of margin against a guard that would take every later test with it.
# Nothing in the product's start-up depends on the count beyond what 4 > 2 exercises (rank 0 sends, three ranks execute the receiver's half,
# the work queue deals blocks to four claimants, the merger interleaves four streams); eight ranks run on the CPU (tests/test_dist_cpu.py).
WORLD = 4


def test_bench_four_ranks_broadcast_through_the_collective_seam(standin_env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "4", "--warmup", "1", "--preheat-ms", "0", "--check",
                        "--regions", "2", "--nrank-files-per-rank", "1", "--nrank-legs", "chunk"],
                       env=standin_env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == WORLD and d["startup_comm"] == "rccl" and d["rccl_nranks"] == WORLD, {k: d[k] for k in ("n_gpus", "startup_comm", "rccl_nranks")}
    assert len(d["ms_per_step_per_rank"]) == WORLD and all(x > 0 for x in d["ms_per_step_per_rank"])
    hb = d["host_budget_rank0"]      # every rank bound itself to its share of the cores before its first GPU call (radian_amd/hostbudget.py)
    usable = len(os.sched_getaffinity(0))
    assert hb["split"] in ("numa", "even") and (not hb["bound"] or hb["cores"] <= max(1, usable // WORLD + 1)), (hb, usable)
    assert abs(d["ms_per_step"] - max(d["ms_per_step_per_rank"])) < 1e-6        # (the reported region's own per-rank clocks)
    assert d["value"] > 0 and d["scaling"] == "weak" and len(d["value_runs"]) == 2 and min(d["value_runs"]) <= d["value"] <= max(d["value_runs"])
    # round 6: the N > 1 line carries the files -> FASTA leg through the multi-GPU route (work queue, native reader, per-rank core slice,
    # rank files, merger process), every rank's share and rate in it
    leg = d["secondary_e2e_fast5_to_fasta"]
    assert "skipped" not in leg, leg
    assert leg["n_ranks"] == WORLD and leg["reads"] == WORLD * 4096 == leg["records_written"] and leg["samples"] == WORLD * 4096 * 4096
    assert leg["value"] > 1e6 and leg["value_to_merged_fasta"] > 1e6 and leg["rccl_nranks"] == WORLD and leg["startup_comm"] == "rccl"
    assert [p["rank"] for p in leg["per_rank"]] == list(range(WORLD)) and sum(p["reads"] for p in leg["per_rank"]) == WORLD * 4096
    assert all(p["cores"] >= 1 for p in leg["per_rank"]) and sum(1 for p in leg["per_rank"] if p["reads"] > 0 and p["value"] > 0) >= WORLD - 1



@pytest.mark.parametrize("mode,model", [("chunk", "none"), ("global", "dense"), ("global", "sparse")])
def test_cli_four_ranks_receive_the_artefacts_by_broadcast(tmp_path, standin_env, monkeypatch, mode, model):
    from radian_amd import fast5, synthetic
    reads = synthetic.synthetic_reads(120, 3000, seed=5)
    rng = np.random.default_rng(2)
    in_dir = tmp_path / "in"
    in_dir.mkdir()
    for f in range(3):
        lo, hi = [0, 30, 90][f], [30, 90, 120][f]
        fast5.write_multi_fast5(str(in_dir / f"r{f}.fast5"), {f"{i:06d}": reads[i][: int(rng.integers(1200, 3000))] for i in range(lo, hi)})
    from radian_amd import basecall, launch
    k = 3
    keys = ["".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k)) for i in range(4 ** k)]
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    dense = {c: [float(x) for x in table[i]] for i, c in enumerate(keys)}

    def common_args(lm_path):
        lm_args = ["--rna-model", "None"] if lm_path is None else ["--rna-model", lm_path, "--context-len", str(k), "--rna-threshold", "5.0",
                                                                  "--sig-threshold", "0.0"]
        return ["--decode-type", mode, "--step-size", "512", "--beam-width", "6", "--sig-model", "synthetic:1234", "--sig-config", "none",
                "--queue-block", "16", "--gpu-batch-windows", "64"] + lm_args

    def single(lm_path, tag):
        out = tmp_path / f"out1_{tag}"
        out.mkdir()
        basecall.main([str(in_dir), str(out)] + common_args(lm_path))          # one process, no communicator
        return open(out / "reads-0.fasta").read()

    lm_path = None
    if model != "none":
        lm_path = str(tmp_path / "lm.json")
        (tmp_path / "lm.json").write_text(json.dumps(dense))
    a = single(lm_path, "dense")
    if model == "sparse":
        # remove a context no read's search reaches (the mask must travel with the broadcast; a read that reached it would raise on every
        # route alike): candidates are the 3-mers that occur in no emitted sequence in either direction
        seqs = [ln for ln in a.splitlines() if not ln.startswith(">")]
        free = [c for c in keys if not any(c in sq or c[::-1] in sq for sq in seqs)]
        assert free, "every 3-mer occurs in the output: pick other weights for this test"
        for tries, c in enumerate(free[:6]):
            sparse = dict(dense)
            del sparse[c]
            (tmp_path / "lm.json").write_text(json.dumps(sparse))
            try:
                a = single(lm_path, f"sparse{tries}")
                break
            except KeyError:
                continue
        else:
            pytest.fail("every candidate context was reached by some read's search")
    common = common_args(lm_path)
    out3 = tmp_path / "out3"
    out3.mkdir()
    argv3 = [str(in_dir), str(out3)] + common + ["--gpus", str(WORLD)]
    args3 = basecall.build_parser().parse_args(argv3)
    monkeypatch.setenv("LD_LIBRARY_PATH", standin_env["LD_LIBRARY_PATH"])     # the ranks are children of this process
    monkeypatch.delenv("RD_CLI_DEVICE", raising=False)
    report = launch.run_multi_gpu(args3, argv3)
    assert [r["transport"] for r in report["ranks"]] == ["rccl"] * WORLD, report
    assert [r["device"] for r in report["ranks"]] == [0] * WORLD or len({r["device"] for r in report["ranks"]}) == WORLD, report
    # the host budget every rank took before its first GPU call: disjoint core slices; chunk mode's stitch threads sized from the slice
    cpus = [r["cpus"] for r in report["ranks"]]
    usable = len(os.sched_getaffinity(0))
    if usable >= WORLD:
        assert all(r["cpu_bound"] for r in report["ranks"]) and len({c for cs in cpus for c in cs}) == sum(len(cs) for cs in cpus) <= usable, report
    if mode == "chunk":
        assert sum(r["stitch_workers"] for r in report["ranks"]) <= max(WORLD, usable), report
    b = open(out3 / "reads-0.fasta").read()
    assert a == b and a.count(">") == 120 and report["records"] == 120


def test_bench_under_the_drivers_own_launcher(standin_env):
    """the round-end driver's command, rank for rank: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` (a foreign launcher: RANK / LOCAL_RANK / WORLD_SIZE come from it, the rendezvous
    directory is named after its pid) -- two ranks, the collectives through the stand-in library: ONE JSON line on the job's stdout."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--preheat-ms", "0"]
    r = subprocess.run(cmd, env=standin_env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["startup_comm"] == "rccl" and d["rccl_nranks"] == 2 and len(d["ms_per_step_per_rank"]) == 2
    assert d["config"]["launcher"].startswith("foreign") and d["value"] > 0 and d["steps"] == 4 and d["warmup"] == 1
    # round 6: under the driver's launcher the line carries BOTH files -> FASTA legs through the multi-GPU route (default flags: 4 files of 4096 reads
    # per rank): configs[2] (chunk) and configs[3]'s geometry (global decode, 12-mer LM) -- the configuration BASELINE.json quotes its scaling curve on
    for key, mode in (("secondary_e2e_fast5_to_fasta", "chunk"), ("secondary_e2e_fast5_to_fasta_global_lm", "global")):
        leg = d[key]
        assert "skipped" not in leg, leg
        assert leg["n_ranks"] == 2 and leg["records_written"] == 2 * 4 * 4096 and leg["value"] > 1e6 and f"--decode-type {mode}" in leg["cli"], (key, leg)
        assert len(leg["per_rank"]) == 2 and sum(p["reads"] for p in leg["per_rank"]) == leg["reads"]
