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


def _spawn(script_args, world, extra_env=None, timeout=300):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable] + script_args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
    return procs, outs


def test_world8_file_queue_order_and_fasta_rotation(tmp_path):
    """8 ranks (file transport), 10 500 reads in 6 fast5 files of uneven size + one empty file, through the launcher's
    worker route: every read exactly once, a rank opens only files it claimed reads from, the streaming merge restores
    input order and the FASTA rotates every 1000 reads (basecall.py:134-138)."""
    import numpy as np
    from radian_amd import fast5, launch
    world = 8
    in_dir = tmp_path / "in"
    (in_dir / "sub").mkdir(parents=True)
    rng = np.random.default_rng(11)
    counts = [4000, 1, 2999, 0, 1500, 1900, 100]
    for fi, n in enumerate(counts):
        reads = {f"f{fi}-{i:05d}": rng.integers(0, 2000, size=24).astype(np.int16) for i in range(n)}
        if fi == 2:
            reads["f2-00007"] = np.full(24, 9, dtype=np.int16)   # MAD == 0 -> skipped, leaves a gap in the keys
        fast5.write_multi_fast5(str((in_dir / "sub" if fi % 2 else in_dir) / f"r{fi}.fast5"), reads)
    files = fast5.list_files(str(in_dir))
    assert len(files) == len(counts)
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    json.dump(files, open(scratch / "files.json", "w"))
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    # the launcher's route: the parent merges WHILE the ranks work (launch.StreamMerger fed by wait_all's poll)
    import time
    merger = launch.StreamMerger(str(scratch), world, str(out_dir))
    progress = []
    def poll():
        merger.poll()
        progress.append(merger.n)
    rcs, _ = launch.run_ranks(world, [sys.executable, os.path.join(ROOT, "tests", "_queue_worker.py"), str(scratch), "0.002"],
                              env_extra={"OMP_NUM_THREADS": "1"}, on_poll=poll)
    t_last = time.time()
    assert rcs == [0] * world, rcs
    n = merger.finish()
    tail = time.time() - t_last
    assert tail <= 0.5, f"merge tail after the last worker: {tail:.2f} s"
    assert progress[-1] >= 0.5 * n and any(0 < x < n for x in progress), (progress[-1], n)   # it really merged as they went
    # expected: the single-process order = files in list order, reads in name order
    exp = []
    for path in files:
        for r in fast5.iter_reads(path):
            raw = r.get_raw_data()
            if (raw == raw[0]).all():
                continue
            exp.append((r.read_id, "".join("ACGT"[c] for c in (raw[:16] % 4))[::-1]))
    assert n == len(exp) == sum(counts) - 1
    got = []
    nfiles = n // 1000 + 1
    assert sorted(os.listdir(out_dir)) == sorted(f"reads-{i}.fasta" for i in range(nfiles))
    for i in range(nfiles):
        lines = open(out_dir / f"reads-{i}.fasta").read().split("\n")[:-1]
        assert len(lines) == (2000 if i < nfiles - 1 else 2 * (n - 1000 * (nfiles - 1)))
        got += [(lines[j][1:], lines[j + 1]) for j in range(0, len(lines), 2)]
    assert got == exp
    # queue accounting
    all_claims, lookups = [], []
    for rank in range(world):
        q = json.load(open(scratch / f"queue{rank}.json"))
        claimed_files = {c[0] for c in q["claimed"]}
        empty = {i for i, p in enumerate(files) if p.endswith("r3.fast5")}
        # a rank counts the reads of a file it needs a block of, or -- ahead of time -- of the file after one it took the first block of
        assert set(q["opened"]) <= claimed_files | empty | set(q["prefetched"]) | {0}, (rank, q["opened"], claimed_files, q["prefetched"])
        all_claims += [tuple(c) for c in q["claimed"]]
        lookups += q["opened"]
    # counts are shared through the queue's side files: every file is counted exactly once per node, not once per rank
    assert sorted(lookups) == list(range(len(files))), lookups
    all_claims.sort()
    covered = {}
    for fi, lo, hi in all_claims:
        assert covered.get(fi, 0) == lo and hi - lo <= 96
        covered[fi] = hi
    sizes = {i: fast5.Fast5Source(p).n_reads() for i, p in enumerate(files)}
    assert covered == {i: nreads for i, nreads in sizes.items() if nreads}
    assert len({c[0] for c in all_claims}) == 6 and len(all_claims) >= sum(counts) // 96


def test_transport_agreement(tmp_path):
    """dist.connect with stand-in rccl_* calls, 3 ranks: (a) all fine -> 'rccl' everywhere; (b) rank 1's
    ncclCommInitRank fails -> every rank drops RCCL and the file transport's barrier / max work; (c) rank 2 cannot even
    load librccl -> NO rank calls ncclCommInitRank; (d) every init succeeds but rank 0's first collective returns an error,
    (e) ... or a wrong answer -> every rank gives its communicator back, file transport everywhere.  Never a mix."""
    script = tmp_path / "c.py"
    script.write_text(
        "import os, sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from radian_amd import dist\n"
        "rank, _, world = dist.env_rank_world()\n"
        "case = sys.argv[1]\n"
        "class Be:\n"
        "    inits = 0; finals = 0\n"
        "    def rccl_unique_id(self):\n"
        "        if case == 'c' and rank == 2: raise RuntimeError('cannot dlopen librccl')\n"
        "        return bytes([rank + 1]) * 128\n"
        "    def rccl_init(self, r, w, uid):\n"
        "        Be.inits += 1\n"
        "        assert uid == bytes([1]) * 128 and (r, w) == (rank, world)\n"
        "        if case == 'b' and rank == 1: raise RuntimeError('ncclCommInitRank: invalid usage')\n"
        "    def rccl_finalize(self): Be.finals += 1\n"
        "    def rccl_barrier(self): pass\n"
        "    def rccl_allreduce_max(self, v):\n"
        "        if case == 'd' and rank == 0 and Be.finals == 0: raise RuntimeError('ncclAllReduce: unhandled cuda error')\n"
        "        if case == 'e' and rank == 2 and Be.finals == 0: return [0.0]\n"
        "        return [float(world - 1)]\n"
        f"comm, kind = dist.connect(Be(), rank, world, os.path.join({str(tmp_path)!r}, 'uid_' + case), timeout=60)\n"
        "comm.barrier()\n"
        "m = comm.allreduce_max([float(rank)])\n"
        "if kind != 'rccl': assert m[0] == world - 1\n"
        "comm.close()\n"
        f"json.dump([kind, type(comm).__name__, Be.inits, Be.finals], open(os.path.join({str(tmp_path)!r}, f'res_{{case}}_{{rank}}.json'), 'w'))\n")
    for case, exp_kind, exp_inits in (("a", "rccl", 1), ("b", "file-fallback", 1), ("c", "file-fallback", 0), ("d", "file-fallback", 1),
                                      ("e", "file-fallback", 1)):
        procs, outs = _spawn([str(script), case], 3, timeout=120)
        assert all(p.returncode == 0 for p in procs), outs
        res = [json.load(open(tmp_path / f"res_{case}_{r}.json")) for r in range(3)]
        assert [r[0] for r in res] == [exp_kind] * 3, res
        assert [r[2] for r in res] == [exp_inits] * 3, res
        if case == "b":   # the ranks whose init had succeeded gave their communicator back
            assert [r[3] for r in res] == [1, 0, 1], res
        if case in "de":  # every rank held one
            assert [r[3] for r in res] == [1, 1, 1], res
    # without the fallback the failure is raised on every rank (the launcher then stops the job)
    from radian_amd import dist

    class Bad:
        def rccl_unique_id(self):
            raise RuntimeError("no librccl")
    with pytest.raises(RuntimeError, match="RCCL start-up failed on rank 0"):
        dist.connect(Bad(), 0, 1, str(tmp_path / "uid_d"), allow_file_fallback=False, force_collective=True)


def test_launcher_stops_job_when_a_worker_fails(tmp_path):
    """launch.wait_all: one worker exits non-zero while the others would block forever -> they are terminated and the
    exit codes come back promptly; a missing artefact is reported by the parent before anything is spawned."""
    import time
    from radian_amd import basecall, launch
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"]) for _ in range(2)]
    procs.insert(1, subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"]))
    rcs = launch.wait_all(procs)
    assert rcs[1] == 3 and all(rc not in (0, None) for rc in rcs) and time.time() - t0 < 30
    (tmp_path / "in").mkdir()
    (tmp_path / "out").mkdir()
    argv = [str(tmp_path / "in"), str(tmp_path / "out"), "--sig-model", str(tmp_path / "nope.h5"), "--gpus", "2"]
    with pytest.raises((FileNotFoundError, OSError, RuntimeError)):
        launch.run_multi_gpu(basecall.build_parser().parse_args(argv), argv)
    argv = [str(tmp_path / "in"), str(tmp_path / "out"), "--sig-model", "synthetic:1", "--sig-config", "none", "--gpus", "2"]
    with pytest.raises(FileNotFoundError):   # default --rna-model is not shipped; global mode needs it
        launch.run_multi_gpu(basecall.build_parser().parse_args(argv), argv)


def test_bench_self_launch_with_stub_ranks(tmp_path, capfd):
    """`python bench.py --gpus N` without a launcher: bench.self_launch starts N fresh ranks (here a device-less stand-in
    that runs the product's start-up through the launcher's private rendezvous directory), forwards rank 0's ONE JSON line,
    removes its scratch directory; a failing rank stops the others promptly and the launcher returns non-zero."""
    import glob
    import tempfile
    import time
    sys.path.insert(0, ROOT)
    import bench
    stub = os.path.join(ROOT, "tests", "_bench_stub_rank.py")
    before = set(glob.glob(os.path.join(tempfile.gettempdir(), "radian_bench_*")))
    rc = bench.self_launch(3, [], worker_cmd=[sys.executable, stub, "ok"])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1, out
    line = json.loads(out[0])
    assert line["n_gpus"] == 3 and line["rccl_nranks"] == 3 and line["ms_per_step_per_rank"] == [1.0, 2.0, 3.0]
    t0 = time.time()
    rc = bench.self_launch(3, [], worker_cmd=[sys.executable, stub, "fail1"])
    cap = capfd.readouterr()
    assert rc == 1 and cap.out.strip() == "" and "exit codes" in cap.err and time.time() - t0 < 30
    assert set(glob.glob(os.path.join(tempfile.gettempdir(), "radian_bench_*"))) == before


def test_bench_nrank_files_leg_with_stub_ranks(tmp_path, capfd, oracle):
    """the N > 1 bench line carries a files -> FASTA leg through the multi-GPU route (FileReadQueue, native reader, per-rank
    core slice, rank files, StreamMerger in a process of its own).  Three stub ranks (device calls by the test double): every rank's share,
    rate and cores are reported, the merged FASTA holds every read once, in input order, and equals a single-process run of the same files;
    a rank that fails inside the leg costs the leg ({"skipped": ...}), never the line."""
    import hashlib
    sys.path.insert(0, ROOT)
    import bench
    from radian_amd import basecall, fast5, weights
    from _oracle_backend import OracleBackend
    stub = os.path.join(ROOT, "tests", "_bench_files_leg_rank.py")
    keep = str(tmp_path / "merged")
    rc = bench.self_launch(3, [], worker_cmd=[sys.executable, stub, "ok", keep])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1, out
    leg = json.loads(out[0])["secondary_e2e_fast5_to_fasta"]
    assert "skipped" not in leg, leg
    assert leg["n_ranks"] == 3 and leg["reads"] == 120 and leg["records_written"] == 120 and leg["samples"] == 120 * 500
    assert leg["value"] > 0 and leg["value_to_merged_fasta"] > 0 and leg["seconds_to_merged_fasta"] >= leg["seconds"] * 0.5
    pr = leg["per_rank"]
    # (the work queue deals 18 blocks of <= 7 reads to whoever asks first: with stub ranks whose warm-up takes a different time each, one rank may
    # find it empty -- a rank without reads reports 0 reads and a rate of 0, it does not fail the leg)
    assert [p["rank"] for p in pr] == [0, 1, 2] and sum(p["reads"] for p in pr) == 120 and all(p["cores"] >= 1 and p["value"] >= 0 for p in pr)
    assert sum(1 for p in pr if p["reads"] > 0 and p["value"] > 0) >= 2
    if len(os.sched_getaffinity(0)) >= 3:
        from radian_amd.hostbudget import parse_cpulist
        sets = [set(parse_cpulist(p["cpus"])) for p in pr]
        assert all(p["cpu_bound"] for p in pr) and not (sets[0] & sets[1]) and not (sets[1] & sets[2]) and all(len(x) == p["cores"] for x, p in zip(sets, pr))
        assert bench.cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert leg["rccl_nranks"] == 3 and leg["startup_comm"] == "stub"
    # the merged FASTA == one process over the same six files
    in_dir, ref_dir = tmp_path / "in", tmp_path / "ref"
    in_dir.mkdir()
    ref_dir.mkdir()
    for fi in range(6):
        fast5.write_multi_fast5(str(in_dir / f"batch_{fi:04d}.fast5"), bench.files_leg_reads(fi, 20, 500))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _bench_files_leg_rank as stubmod
    args = basecall.build_parser().parse_args([str(in_dir), str(ref_dir)] + stubmod.CLI)
    args._lm_loaded = False
    be = OracleBackend()
    be.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2)), (1, 2))
    w = basecall.FastaWriter(str(ref_dir))
    import contextlib
    assert sorted(leg["file_order"]) == [f"batch_{fi:04d}.fast5" for fi in range(6)]
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):     # (the files in the order the leg's directory enumerated them: rglob order is the filesystem's)
        basecall.run(args, be, writer=w, sources=[fast5.Fast5Source(str(in_dir / name)) for name in leg["file_order"]])
    w.close()
    assert open(os.path.join(keep, "reads-0.fasta")).read() == open(ref_dir / "reads-0.fasta").read()
    assert leg["fasta_sha256"] == hashlib.sha256(open(ref_dir / "reads-0.fasta", "rb").read()).hexdigest()
    ids = [ln[1:].strip() for ln in open(os.path.join(keep, "reads-0.fasta")) if ln.startswith(">")]
    assert ids == [rid for name in leg["file_order"] for rid in sorted(bench.files_leg_reads(int(name[6:10]), 20, 500))]
    # a rank that fails inside the leg: the line still comes, the leg says why, nothing is left in /dev/shm
    import glob
    rc = bench.self_launch(3, [], worker_cmd=[sys.executable, stub, "fail1", str(tmp_path / "none")])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1, out
    line = json.loads(out[0])
    assert line["value"] == 1.0 and "rank 1" in line["secondary_e2e_fast5_to_fasta"]["skipped"]
    assert not glob.glob("/dev/shm/radian_bench_nrank_*") and not os.path.exists(str(tmp_path / "none"))


def test_workers_find_the_package_from_a_foreign_working_directory(tmp_path):
    """`python3 /path/to/repo/basecall.py in out --gpus N` from the directory that holds models/ (the reference's way, basecall.py:28-30): the
    parent finds the package beside the script, but its rank processes are `python -m radian_amd.launch --worker ...` children that inherit
    neither sys.path nor the cwd's luck -- launch.package_env() puts the package's parent on their PYTHONPATH (round 6: they failed with
    ModuleNotFoundError before)."""
    from radian_amd import launch
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    code = ("import os, sys; from radian_amd import launch; os.chdir(sys.argv[1]); "
            "rcs, _ = launch.run_ranks(2, [sys.executable, '-c', 'import radian_amd.launch, radian_amd.basecall'], env_extra=launch.package_env()); "
            "bad, _ = launch.run_ranks(2, [sys.executable, '-c', 'import radian_amd.launch']); print(rcs, bad)")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "[0, 0] [1, 1]", (r.stdout, r.stderr[-500:])
    assert launch.package_env()["PYTHONPATH"].split(os.pathsep)[0] == ROOT


def test_merge_watch_ends_with_the_ranks_or_at_its_deadline(tmp_path):
    """launch.merge_watch (round 6: the merging half of the multi-GPU route as a process of its own, bench.py's N-rank legs): merges while the rank
    files grow, returns when every rank has written its end mark; a rank that never ends makes it raise TimeoutError with no partial FASTA left
    behind (the hidden directory is removed)."""
    import threading
    import time
    from radian_amd import launch
    scratch, out = tmp_path / "s", tmp_path / "o"
    scratch.mkdir()
    out.mkdir()

    def rank(r, n, end=True, delay=0.0):
        f = launch._RankFile(str(scratch / f"rank{r}.jsonl"))
        f.claim(r, 0, n)
        for i in range(n):
            f.emit((r, i), f"read-{r}-{i}", "ACGT" * (i + 1))
            time.sleep(delay)
        if end:
            f.end()
        else:
            f.f.flush()
    ths = [threading.Thread(target=rank, args=(r, 30, True, 0.002)) for r in range(3)]
    for t in ths:
        t.start()
    res = launch.merge_watch(str(scratch), 3, str(out), timeout=30.0)
    for t in ths:
        t.join()
    assert res["records"] == 90 and sorted(os.listdir(out)) == ["reads-0.fasta"]
    ids = [ln[1:].strip() for ln in open(out / "reads-0.fasta") if ln.startswith(">")]
    assert ids == [f"read-{r}-{i}" for r in range(3) for i in range(30)]
    # a rank that never writes its end mark: TimeoutError, nothing left in the output directory
    for f in os.listdir(scratch):
        os.remove(scratch / f)
    os.remove(out / "reads-0.fasta")
    rank(0, 5, end=True)
    rank(1, 5, end=False)
    t0 = time.time()
    with pytest.raises(TimeoutError, match=r"ranks \[1\]"):
        launch.merge_watch(str(scratch), 2, str(out), timeout=1.0)
    assert time.time() - t0 < 10 and os.listdir(out) == []


def test_bench_parent_never_loads_the_hip_library(tmp_path):
    """The launcher process of `bench.py --gpus N` must not touch a GPU: it runs to completion (here: to the failure of its
    ranks, which have no GPU) without libradian_hip.so ever being mapped -- RADIAN_HIP_LIB points at a file that is not a
    library, so any load in the PARENT would raise there instead of in the ranks."""
    bogus = tmp_path / "not_a_lib.so"
    bogus.write_text("x")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(os.environ, RADIAN_HIP_LIB=str(bogus)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    err = r.stderr.decode()
    assert r.returncode == 1 and r.stdout.decode().strip() == ""
    assert "multi-GPU run failed: rank exit codes" in err          # the parent got as far as reaping its ranks
    assert err.count("invalid ELF header") >= 1 or "not_a_lib" in err   # ... and the ranks were the ones that tried to load it


def test_connect_fails_fast_when_a_peer_fails_while_this_rank_is_inside_init(tmp_path):
    """a rank whose ncclCommInitRank fails leaves its peers blocked inside theirs.  The peers' main threads watch
    the rendezvous and raise StartupFailed(stuck=True) after the grace period instead of hanging for the full timeout."""
    script = tmp_path / "s.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from radian_amd import dist\n"
        "rank, _, world = dist.env_rank_world()\n"
        "class Be:\n"
        "    def rccl_unique_id(self): return bytes(128)\n"
        "    def rccl_probe(self): pass\n"
        "    def rccl_init(self, r, w, uid):\n"
        "        if rank == 1: raise RuntimeError('ncclCommInitRank: unhandled system error')\n"
        "        time.sleep(600)   # the collective never completes without rank 1\n"
        "t0 = time.time()\n"
        "try:\n"
        f"    dist.connect(Be(), rank, world, os.path.join({str(tmp_path)!r}, 'uid'), timeout=60, stuck_grace=0.5)\n"
        "except dist.StartupFailed as e:\n"
        "    assert e.stuck and 'rank 1' in str(e) and time.time() - t0 < 20, e\n"
        "    os._exit(3)\n"
        "except RuntimeError as e:\n"
        "    sys.exit(4 if rank == 1 else 5)   # rank 1 waits for the others' outcome, which never comes\n"
        "sys.exit(0)\n")
    import time
    from radian_amd import launch
    t0 = time.time()
    rcs, _ = launch.run_ranks(3, [sys.executable, str(script)])
    assert rcs[0] == 3 or rcs[2] == 3, rcs       # a stuck rank left first; wait_all then stopped the rest
    assert all(rc != 0 for rc in rcs) and time.time() - t0 < 40, rcs


def test_stream_merger_any_interleaving(tmp_path):
    """launch.StreamMerger against adversarial timing: rank files that grow in random increments (partial lines included), claims
    announced long before their records, claims whose reads are all skipped, ranks that end early, ranks without a queue (static
    sharding: no claim marks).  Whatever the interleaving: every record once, in key order, and a record is never written before a
    smaller key that some rank can still produce."""
    import random
    from radian_amd import launch
    rnd = random.Random(5)
    for trial in range(30):
        world = rnd.randint(1, 6)
        with_claims = rnd.random() < 0.7
        # the job: files with reads; blocks handed to ranks in key order
        blocks, fi = [], 0
        for fi in range(rnd.randint(1, 5)):
            n, lo = rnd.randint(0, 60), 0
            while lo < n:
                hi = min(n, lo + rnd.randint(1, 16))
                blocks.append((fi, lo, hi))
                lo = hi
        owner = [rnd.randrange(world) for _ in blocks]
        skipped = {(f, r) for (f, lo, hi) in blocks for r in range(lo, hi) if rnd.random() < 0.15}
        lines = [[] for _ in range(world)]
        for (f, lo, hi), o in zip(blocks, owner):
            if with_claims:
                lines[o].append(json.dumps({"claim": [f, lo, hi]}))
            for r in range(lo, hi):
                if (f, r) not in skipped:
                    lines[o].append(json.dumps([[f, r], f"id{f}-{r}", "ACGT"[(f + r) % 4] * 3]))
        if with_claims:      # a rank announces a claim as soon as it makes it: move the marks ahead of earlier records at random
            for o in range(world):
                ls, i = lines[o], 0
                while i < len(ls):
                    if ls[i].startswith('{"claim"') and i > 0 and rnd.random() < 0.5:
                        j = rnd.randint(max(0, i - 20), i)
                        # (never ahead of an EARLIER claim mark: claims are made in order)
                        while j < i and any(x.startswith('{"claim"') for x in ls[j:i]):
                            j += 1
                        ls.insert(j, ls.pop(i))
                    i += 1
        data = [("\n".join(ls + [json.dumps({"end": True})]) + "\n").encode() for ls in lines]
        scratch = tmp_path / f"s{trial}"
        out = tmp_path / f"o{trial}"
        scratch.mkdir()
        out.mkdir()
        pos = [0] * world
        fhs = [open(scratch / f"rank{r}.jsonl", "wb") for r in range(world)]
        m = launch.StreamMerger(str(scratch), world, str(out))
        written = 0
        while any(pos[r] < len(data[r]) for r in range(world)):
            r = rnd.choice([x for x in range(world) if pos[x] < len(data[x])])
            k = rnd.randint(1, 200)
            fhs[r].write(data[r][pos[r]: pos[r] + k])
            fhs[r].flush()
            pos[r] += k
            if rnd.random() < 0.5:
                m.poll()
                assert m.n >= written
                written = m.n
        for f in fhs:
            f.close()
        n = m.finish()
        exp = [(f"id{f}-{r}", ("ACGT"[(f + r) % 4] * 3)[::-1]) for (f, lo, hi) in blocks for r in range(lo, hi) if (f, r) not in skipped]
        got = []
        for i in range(n // 1000 + 1):
            ls = open(out / f"reads-{i}.fasta").read().split("\n")[:-1]
            got += [(ls[j][1:], ls[j + 1]) for j in range(0, len(ls), 2)]
        assert n == len(exp) and got == exp, (trial, world, with_claims)


def test_stream_merger_bound_moves_past_finished_claims_and_abort_leaves_nothing(tmp_path):
    """ A claim whose last read has reported no longer holds the rank's lower bound once a newer claim is known,
    so other ranks' records are written instead of piling up in the parent; the FASTA appears in fasta_dir only at finish(),
    and a failed job (abort) leaves no reads-*.fasta behind."""
    from radian_amd import launch
    scratch, out = tmp_path / "s", tmp_path / "o"
    scratch.mkdir()
    out.mkdir()
    r0 = [{"claim": [0, 0, 4]}, {"claim": [0, 8, 12]}] + [[[0, i], f"id{i}", "ACGT"] for i in range(4)]
    r1 = [{"claim": [0, 4, 8]}] + [[[0, i], f"id{i}", "ACGT"] for i in range(4, 8)]
    for r, lines in enumerate((r0, r1)):
        with open(scratch / f"rank{r}.jsonl", "w") as f:
            f.write("".join(json.dumps(x) + "\n" for x in lines))
    m = launch.StreamMerger(str(scratch), 2, str(out))
    m.poll()
    assert m.n == 8 and not any(m.recs)             # rank 0's bound is its second claim's first read, (0, 8)
    assert not [n for n in os.listdir(out) if n.startswith("reads-")]
    m.abort()
    assert os.listdir(out) == []
    m = launch.StreamMerger(str(scratch), 2, str(out))
    assert m.finish() == 8
    assert os.listdir(out) == ["reads-0.fasta"]
    assert open(out / "reads-0.fasta").read().count(">") == 8


def test_host_feed_dry_run_rate_floor(tmp_path):
    """ The host side of a node WITHOUT its GPUs: tools/host_feed_bench.py runs the CLI's multi-GPU route -- launcher,
    per-node file/read queue, fast5 parsing on each rank's read-ahead thread, the driver loop's batching and pipeline tickets, rank
    result streams, the parent's streaming merge, FASTA rotation -- with a null device that returns at once.  Two ranks, 8000 reads
    of 4096 samples: every read once, in input order, nine FASTA files, and a rate floor an order of magnitude under what this
    measures on an idle 8-core box (2 ranks: ~45 M samples/s) -- a regression to per-read Python parsing
    (0.3 M samples/s per core) or to a merge that runs after the ranks would fall through it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import host_feed_bench as hb
    res = hb.bench_ranks(2, 8000, 4096, 3, "global", keep=str(tmp_path))
    assert res["records"] == 8000 and res["fasta_files"] == 9
    assert res["M_samples_per_s"] >= 4.0, res
    assert res["merge_after_last_rank_s"] <= 2.0, res
    ids = []
    for i in range(9):
        with open(tmp_path / "out" / f"reads-{i}.fasta") as f:
            ids += [ln[1:].strip() for ln in f if ln.startswith(">")]
    exp = [f"{fi:03d}-{i:07d}" for fi in range(3) for i in range(2667 if fi < 2 else 2666)]
    assert ids == exp


def test_rccl_bcast_waits_for_rank0_load_outside_the_collective(tmp_path):
    """RcclComm.bcast_artifacts: the receivers wait for rank 0's "loaded" mark WITHOUT entering the broadcast and start the
    broadcast's deadline there -- a load slower than that deadline is fine; a load that fails leaves no rank inside a collective (the
    receivers raise StartupFailed(stuck=False), rank 0 re-raises its own error); a rank 0 that never finishes runs into load_timeout."""
    import threading
    import time
    from radian_amd import dist

    class FakeBackend:
        def __init__(self):
            self.bcasts = 0

        def rccl_bcast_model(self, root):
            self.bcasts += 1

    def run(load, **kw):
        d = tmp_path / f"rdv{len(os.listdir(tmp_path))}"
        bes = [FakeBackend(), FakeBackend()]
        out = [None, None]

        def rank(r):
            comm = dist.RcclComm(bes[r], r, 2, dist.Rendezvous(str(d), r, 2))
            try:
                comm.bcast_artifacts(bes[r], load, **kw)
                out[r] = "ok"
            except BaseException as e:
                out[r] = e

        ths = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(30)
        return out, bes

    # a load that takes longer than the broadcast's own deadline
    out, bes = run(lambda b: time.sleep(0.6), timeout=0.3, load_timeout=10.0)
    assert out == ["ok", "ok"] and [b.bcasts for b in bes] == [1, 1]

    def boom(b):
        raise FileNotFoundError("models/rnamodel_12mer_pc.json")
    out, bes = run(boom, timeout=5.0, load_timeout=10.0)
    assert isinstance(out[0], FileNotFoundError)
    assert isinstance(out[1], dist.StartupFailed) and not out[1].stuck and "rnamodel_12mer_pc.json" in str(out[1])
    assert [b.bcasts for b in bes] == [0, 0]
    # rank 0 stalls inside its load: the receiver gives up on the load's own deadline, still outside the collective
    out, bes = run(lambda b: time.sleep(1.5), timeout=5.0, load_timeout=0.3)
    assert isinstance(out[1], dist.StartupFailed) and not out[1].stuck and "did not finish loading" in str(out[1]) and bes[1].bcasts == 0
