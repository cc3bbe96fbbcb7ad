#!/usr/bin/env python3
"""Host-feed budget of a multi-GPU node, measured WITHOUT a GPU (SURVEY section 7 "Host feed at 8 GPUs").

  read     fast5 parsing alone: samples/s of fast5.iter_reads + get_raw_data on one core, ctypes->libhdf5 and the pure-Python reader
  ranks    a dry run of the CLI's multi-GPU route with N worker processes whose device is a NULL backend that returns at once:
           launch.run_ranks -> per-node FileReadQueue claims -> Fast5Source parsing -> basecall.run's batching / pipeline tickets ->
           rank result streams -> launch.StreamMerger -> FASTA rotation.  What it reports is the rate at which the HOST side of
           the node could feed and drain its GPUs: reads/s, samples/s, and per worker process.
  merger   launch.StreamMerger alone: N rank result files with 2-kb sequences written beforehand, merged into the FASTA layout
usage: host_feed_bench.py read [n_reads=16384] [len=4096]
       host_feed_bench.py ranks [world=8] [n_reads=100000] [len=4096] [files=8] [mode=global|chunk]
       host_feed_bench.py merger [world=8] [n_records=200000] [seq_len=2048]
Every worker of `ranks` takes its host budget exactly as launch.worker does (launch.rank_budget: core slice + thread counts) and reports
the threads it had alive at its busiest.  FEED_GZIP=1: the files' signals are deflate-compressed (pre-VBZ MinKNOW files)."""
import json, os, shutil, sys, tempfile, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


class NullTicket:
    def __init__(self, mode, raws, chunk, step):
        self.mode, self.raws, self.chunk, self.step = mode, raws, chunk, step

    def result(self):          # global: one labeling per read (~ a base per 8 samples)
        return [np.zeros(len(r) // 8, dtype=np.uint8) for r in self.raws], np.zeros(len(self.raws), dtype=np.int32)

    def result_raw(self):      # chunk: the label matrix as the device leaves it (~ 64 bases per window), stitched natively
        nw = [(0 if len(r) < self.chunk else (len(r) - self.chunk) // self.step + 1) + 1 for r in self.raws]
        tot = sum(nw)
        lab = np.zeros((tot, self.chunk), dtype=np.uint8)
        lab[:, :64] = np.arange(64, dtype=np.uint8) % 4
        return lab, np.full(tot, 64, dtype=np.int32), nw, np.zeros(len(self.raws), dtype=np.int32)


class NullBackend:
    """the Backend surface basecall.run uses, with a device that takes no time"""

    def pipe_submit_raw(self, decode_type, raws, outlier_clip, chunk_len, step, beam_width, use_lm=False, s_threshold=0.0, r_threshold=0.0):
        np.concatenate(raws)   # (what Backend._pack_raw costs: one copy of the batch's samples)
        return NullTicket(decode_type, raws, chunk_len, step)

    def close(self):
        pass


class NullComm:
    def barrier(self):
        pass

    def close(self):
        pass


def write_files(d, n_reads, length, n_files):
    from radian_amd import fast5
    rng = np.random.default_rng(0)
    per = -(-n_reads // n_files)
    base = np.round(rng.normal(500, 80, size=length * 64)).astype(np.int16)
    done = 0
    for fi in range(n_files):
        k = min(per, n_reads - done)
        if k <= 0:
            break
        fast5.write_multi_fast5(os.path.join(d, f"f{fi:03d}.fast5"),
                                {f"{fi:03d}-{i:07d}": base[(i % 63) * length: (i % 63) * length + length] for i in range(k)},
                                filters=(("deflate", 1),) if os.environ.get("FEED_GZIP") == "1" else ())
        done += k
    return done


def bench_read(n_reads, length):
    from radian_amd import fast5, h5
    d = tempfile.mkdtemp(prefix="rd_feed_")
    try:
        write_files(d, n_reads, length, 1)
        path = os.path.join(d, "f000.fast5")
        out = {"reads": n_reads, "samples_per_read": length, "file_MB": round(os.path.getsize(path) / 1e6)}
        for name in ("libhdf5", "pure"):
            if name == "pure":
                os.environ["RADIAN_HDF5_PURE"] = "1"
            t0 = time.perf_counter()
            n = s = 0
            for r in fast5.iter_reads(path):
                s += r.get_raw_data().shape[0]
                n += 1
            dt = time.perf_counter() - t0
            out[name] = {"reads_per_s": round(n / dt), "M_samples_per_s": round(s / dt / 1e6, 1), "us_per_read": round(dt / n * 1e6, 1)}
        os.environ.pop("RADIAN_HDF5_PURE", None)
        print(json.dumps(out))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def worker(scratch, argv):
    from radian_amd import fast5, launch
    from radian_amd.basecall import build_parser
    from radian_amd.dist import env_rank_world
    args = build_parser().parse_args(argv)
    rank, local_rank, world = env_rank_world()
    budget = launch.rank_budget(args, local_rank, world)        # the first act of a rank, as in launch.worker
    args._lm_loaded = False
    import threading
    peak = {"threads": 0}
    stop = threading.Event()

    def census():
        while not stop.wait(0.05):
            try:
                tids = os.listdir("/proc/self/task")
                if len(tids) - 1 > peak["threads"]:
                    peak["threads"] = len(tids) - 1   # (minus this census thread)
                    peak["names"] = sorted(n_.name for n_ in threading.enumerate()) + sorted(open(f"/proc/self/task/{t}/comm").read().strip() for t in tids if os.path.exists(f"/proc/self/task/{t}/comm"))
            except OSError:
                pass
    threading.Thread(target=census, daemon=True).start()
    with open(os.path.join(scratch, "files.json")) as f:
        sources = [fast5.Fast5Source(p) for p in json.load(f)]
    sys.stdout = open(os.devnull, "w")      # (the per-read "Basecalled read ..." lines: the CLI prints them; a job redirects them)
    launch.run_rank(args, NullBackend(), NullComm(), scratch, sources, rank, world)
    stop.set()
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    with open(os.path.join(scratch, f"budget{rank}.json"), "w") as f:
        json.dump({"cpu_s": round(ru.ru_utime + ru.ru_stime, 2), "cpus": budget["cpus"], "split": budget["how"], "bound": budget["bound"], "stitch_workers": args.stitch_workers,
                   "peak_threads": peak["threads"], "names": peak.get("names")}, f)


def bench_ranks(world, n_reads, length, n_files, mode, keep=None):
    from radian_amd import fast5, launch
    d = keep or tempfile.mkdtemp(prefix="rd_feed_")
    try:
        in_dir, out_dir, scratch = os.path.join(d, "in"), os.path.join(d, "out"), os.path.join(d, "scratch")
        for x in (in_dir, out_dir, scratch):
            os.makedirs(x, exist_ok=True)
        n = write_files(in_dir, n_reads, length, n_files)
        argv = [in_dir, out_dir, "--decode-type", mode, "--step-size", "512", "--beam-width", "10", "--rna-model", "None",
                "--sig-model", "synthetic", "--sig-config", "none"]
        with open(os.path.join(scratch, "files.json"), "w") as f:
            json.dump(sorted(fast5.list_files(in_dir)), f)
        merger = launch.StreamMerger(scratch, world, out_dir)
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", scratch, "--"] + argv
        t0 = time.perf_counter()
        rcs, _ = launch.run_ranks(world, cmd, on_poll=merger.poll)
        t_ranks = time.perf_counter() - t0
        assert not any(rcs), rcs
        written = merger.finish()
        dt = time.perf_counter() - t0
        budgets = [json.load(open(os.path.join(scratch, f"budget{r}.json"))) for r in range(world)]
        if os.environ.get("RD_FEED_VERBOSE"):
            print(budgets[0]["names"], file=sys.stderr)
        usable = len(os.sched_getaffinity(0))
        res = {"world": world, "mode": mode, "reads": n, "samples_per_read": length, "cores": os.cpu_count(), "usable_cores": usable,
               "cores_per_rank": [len(b["cpus"]) for b in budgets], "cpu_split": budgets[0]["split"], "bound": all(b["bound"] for b in budgets),
               "slices_disjoint": len({c for b in budgets for c in b["cpus"]}) == sum(len(b["cpus"]) for b in budgets),
               "stitch_threads_per_rank": [b["stitch_workers"] for b in budgets], "stitch_threads_total": sum((b["stitch_workers"] or 0) for b in budgets),
               "peak_threads_per_rank": [b["peak_threads"] for b in budgets], "seconds": round(dt, 2),
               "rank_cpu_seconds": [b["cpu_s"] for b in budgets], "cores_busy_ranks": round(sum(b["cpu_s"] for b in budgets) / dt, 2),
               "fast5_reader": "libhdf5" if os.environ.get("RADIAN_FAST5_NATIVE") == "0" else "native",
               "merge_after_last_rank_s": round(dt - t_ranks, 2), "records": written, "reads_per_s": round(n / dt),
               "M_samples_per_s": round(n * length / dt / 1e6, 1), "M_samples_per_s_per_rank": round(n * length / dt / 1e6 / world, 1),
               "fasta_files": len([x for x in os.listdir(out_dir) if x.startswith("reads-")])}
        print(json.dumps(res))
        return res
    finally:
        if keep is None:
            shutil.rmtree(d, ignore_errors=True)


def bench_merger(world, n_records, seq_len):
    """StreamMerger alone: the eight ranks' result streams exist already (round-robin blocks of 256 reads over the ranks, as the work
    queue deals them), sequences of seq_len bases; time poll-until-done + finish.  Target (VERDICT r4 #2c): >= 60 k records/s with 2-kb
    sequences = 8 x 27 M samples/s / 4096 samples per read."""
    from radian_amd import launch
    d = tempfile.mkdtemp(prefix="rd_merge_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.environ.get("RD_MERGE_ON_DISK") is None else None)
    try:   # (memory-backed by default: the container's overlay disk adds 2-3x run-to-run noise that is not the merger's)
        scratch, out_dir = os.path.join(d, "scratch"), os.path.join(d, "out")
        os.makedirs(scratch)
        os.makedirs(out_dir)
        rng = np.random.default_rng(0)
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=seq_len))
        files = [launch._RankFile(os.path.join(scratch, f"rank{r}.jsonl")) for r in range(world)]
        block = 256
        for b0 in range(0, n_records, block):
            r = (b0 // block) % world
            files[r].claim(0, b0, min(n_records, b0 + block))
            for i in range(b0, min(n_records, b0 + block)):
                files[r].emit((0, i), f"read-{i:08d}", seq)
        for f in files:
            f.end()
        t0, c0 = time.perf_counter(), time.process_time()
        m = launch.StreamMerger(scratch, world, out_dir)
        n = m.finish()
        dt, dc = time.perf_counter() - t0, time.process_time() - c0
        assert n == n_records
        print(json.dumps({"merger_alone": {"world": world, "records": n, "seq_len": seq_len, "seconds": round(dt, 2), "cpu_seconds": round(dc, 2), "records_per_s": round(n / dt),
                                           "MB_per_s": round(n * (seq_len + 16) / dt / 1e6), "fasta_files": len(os.listdir(out_dir))}}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--worker":
        worker(a[1], a[3:])
    elif a and a[0] == "merger":
        bench_merger(int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 200000, int(a[3]) if len(a) > 3 else 2048)
    elif a and a[0] == "read":
        bench_read(int(a[1]) if len(a) > 1 else 16384, int(a[2]) if len(a) > 2 else 4096)
    else:
        a = a[1:] if a and a[0] == "ranks" else a
        bench_ranks(int(a[0]) if len(a) > 0 else 8, int(a[1]) if len(a) > 1 else 100000, int(a[2]) if len(a) > 2 else 4096,
                    int(a[3]) if len(a) > 3 else 8, a[4] if len(a) > 4 else "global")
