"""N>1 path on CPU: world_size-2 gloo job running the product's sharded driver; merged output must equal the
single-process run, in input order."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,queue_block", [("chunk", 0), ("global", 0), ("chunk", 2)])
def test_two_rank_gloo_equals_single(tmp_path, mode, queue_block, oracle):
    """queue_block 0: static round-robin shard; > 0: the ranks claim blocks of reads from the per-node work queue"""
    world = 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="3")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path), mode, str(queue_block)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    from radian_amd import basecall, dist, weights
    from _oracle_backend import OracleBackend
    from _reads import golden_reads
    import numpy as np
    per_rank = []
    for rank in range(world):
        with open(tmp_path / f"rank{rank}.jsonl") as f:
            per_rank.append([tuple(json.loads(l)) for l in f])
    if queue_block == 0:
        assert sorted(i for i, _, _ in per_rank[0]) == dist.shard_indices(5, 0, 2)
        assert sorted(i for i, _, _ in per_rank[1]) == dist.shard_indices(5, 1, 2)
    else:   # whoever claimed what: every read exactly once, in blocks of queue_block consecutive indices per claim
        owned = sorted(i for rr in per_rank for i, _, _ in rr)
        assert owned == [0, 1, 2, 3, 4]
        for rr in per_rank:
            blocks = {i // queue_block for i, _, _ in rr}
            assert all(b not in {j // queue_block for j, _, _ in other} for other in per_rank if other is not rr for b in blocks)
    merged = dist.merge_results(per_rank)
    # single process reference run of the same driver
    args = basecall.build_parser().parse_args(
        ["a", "b", "--chunk-len", "256", "--step-size", "128", "--beam-width", "4", "--decode-type", mode,
         "--gpu-batch-windows", "24", "--context-len", "3"])
    be = OracleBackend()
    be.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2, 4)), (1, 2, 4))
    if mode == "global":
        be.load_lm(np.random.default_rng(9).dirichlet([0.3] * 4, size=64), 3)
    args._lm_loaded = mode == "global"
    single = basecall.run(args, be, reads=golden_reads(1500), writer=None)
    assert [tuple(x) for x in merged] == [tuple(x) for x in single]
    assert [i for i, _, _ in merged] == [0, 1, 2, 3, 4]
    assert all(len(s) > 0 for _, _, s in merged)


def test_uid_rendezvous_and_filecomm_under_torchrun(tmp_path):
    """torch.distributed.run (the driver's launcher) with 3 ranks on CPU: the launcher-pid-keyed rendezvous file gives
    every rank the same id, and the file-based fallback transport computes barrier + max correctly."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, hashlib\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from radian_amd import dist\n"
        "rank, lr, world = dist.env_rank_world()\n"
        "p = dist.uid_path()\n"
        "uid = dist.exchange_uid(lambda: os.urandom(128), rank, p)\n"
        "c = dist.FileComm(rank, world, p + '.fc')\n"
        "c.barrier()\n"
        "m = c.allreduce_max([float(rank), 10.0 - rank])\n"
        "assert list(m) == [world - 1.0, 10.0], m\n"
        "c.close()\n"
        f"open(os.path.join({str(tmp_path)!r}, f'uid{{rank}}'), 'w').write(hashlib.md5(uid).hexdigest())\n")
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    uids = {open(tmp_path / f"uid{i}").read() for i in range(3)}
    assert len(uids) == 1


def test_work_queue_many_claimers(tmp_path):
    """WorkQueue under contention: 4 processes walk 0..N-1 at different speeds; every index is owned exactly once."""
    script = tmp_path / "q.py"
    script.write_text(
        "import os, sys, time, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from radian_amd.dist import WorkQueue\n"
        "me, n = int(sys.argv[1]), int(sys.argv[2])\n"
        f"q = WorkQueue(os.path.join({str(tmp_path)!r}, 'queue'), 7)\n"
        "mine = []\n"
        "for i in range(n):\n"
        "    if q.owns(i):\n"
        "        mine.append(i)\n"
        "        if me % 2: time.sleep(0.0003)\n"
        "q.close()\n"
        f"json.dump(mine, open(os.path.join({str(tmp_path)!r}, f'own{{me}}.json'), 'w'))\n")
    n = 1000
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(n)]) for r in range(4)]
    assert all(p.wait(timeout=120) == 0 for p in procs)
    owned = [json.load(open(tmp_path / f"own{r}.json")) for r in range(4)]
    assert sorted(i for o in owned for i in o) == list(range(n))   # (a late starter may find nothing left: fine)
