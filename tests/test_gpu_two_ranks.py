"""The N > 1 route on the hardware that is reachable: TWO ranks on ONE GPU (RD_BENCH_DEVICE / RD_CLI_DEVICE pin both to device 0).
RCCL refuses the duplicate device on both ranks at once, so what runs here is everything around the broadcast: the launchers
(bench.py's own, radian_amd.launch), the rendezvous, the collective choice of the transport (both ranks leave ncclCommInitRank with
an error and agree on the file transport), sharding / the work queue, the streaming merge.  (a >= 2-rank GPU test of
the connect and launch.worker route.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_two_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--preheat-ms", "0", "--regions", "2",
                        "--nrank-files-per-rank", "1", "--nrank-legs", "chunk"],
                       env=dict(os.environ, RD_BENCH_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE JSON line on stdout, whatever the ranks and RCCL print
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["startup_comm"] in ("file-fallback", "rccl") and d["rccl_nranks"] == 2
    assert len(d["ms_per_step_per_rank"]) == 2 and all(x > 0 for x in d["ms_per_step_per_rank"])
    assert abs(d["ms_per_step"] - max(d["ms_per_step_per_rank"])) < 1e-6      # the line's time is the slowest rank's
    assert d["config"]["launcher"] == "bench.py (own)" and d["value"] > 0 and len(d["value_runs"]) == 2
    leg = d["secondary_e2e_fast5_to_fasta"]      # round 6: the files -> FASTA leg of the N > 1 line, here on the file transport
    assert "skipped" not in leg and leg["n_ranks"] == 2 and leg["records_written"] == 2 * 4096 and len(leg["per_rank"]) == 2, leg


@pytest.mark.parametrize("mode", ["chunk", "global"])
def test_cli_two_ranks_equal_single_process(tmp_path, mode):
    from radian_amd import basecall, fast5, synthetic
    reads = synthetic.synthetic_reads(150, 3000, seed=3)
    rng = np.random.default_rng(1)
    in_dir = tmp_path / "in"
    in_dir.mkdir()
    for f in range(3):     # three files of uneven size: the queue crosses file boundaries
        lo, hi = [0, 20, 110][f], [20, 110, 150][f]
        fast5.write_multi_fast5(str(in_dir / f"r{f}.fast5"), {f"{i:06d}": reads[i][: int(rng.integers(1200, 3000))] for i in range(lo, hi)})
    outs = {}
    for g in (1, 2):
        out = tmp_path / f"out{g}"
        out.mkdir()
        cmd = [sys.executable, "-m", "radian_amd.basecall", str(in_dir), str(out), "--decode-type", mode, "--step-size", "512", "--beam-width", "6",
               "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", "None", "--gpus", str(g), "--queue-block", "16",
               "--gpu-batch-windows", "64"]
        env, cwd = dict(os.environ, RD_CLI_DEVICE="0", PYTHONPATH=ROOT), ROOT
        if g == 2 and mode == "global":
            # round 6: the two-rank run through the reference's literal entry point from a foreign working directory, nothing on PYTHONPATH -- the
            # rank processes must find the package by themselves (launch.package_env)
            cmd[1:3] = [os.path.join(ROOT, "basecall.py")]
            env = {k: v for k, v in env.items() if k != "PYTHONPATH"}
            cwd = str(tmp_path)
        r = subprocess.run(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs[g] = open(out / "reads-0.fasta").read()
    assert outs[1] == outs[2] and outs[1].count(">") == 150


def test_device_for_rank_follows_visibility():
    """a rank's device is its local rank, or -- when the launcher narrowed every process's view to fewer devices than local ranks --
    the local rank modulo what is visible (a one-GPU box sees one device: every local rank maps to device 0)"""
    from radian_amd.backend import device_count, device_for_rank
    n = device_count()
    assert n >= 1 and device_for_rank(0) == 0
    assert [device_for_rank(r) for r in range(n, n + 3)] == [r % n for r in range(n, n + 3)]
