#!/usr/bin/env python3
"""Host-feed budget of a multi-GPU node, measured WITHOUT a GPU (SURVEY section 7 "Host feed at 8 GPUs").

  read     fast5 parsing alone: samples/s of fast5.iter_reads + get_raw_data on one core, ctypes->libhdf5 and the pure-Python reader
  ranks    a dry run of the CLI's multi-GPU route with N worker processes whose device is a NULL backend that returns at once:
           launch.run_ranks -> per-node FileReadQueue claims -> Fast5Source parsing -> basecall.run's batching / pipeline tickets ->
           rank result streams -> launch.StreamMerger -> FASTA rotation.  What it reports is the rate at which the HOST side of
           the node could feed and drain its GPUs: reads/s, samples/s, and per worker process.
usage: host_feed_bench.py read [n_reads=16384] [len=4096]
       host_feed_bench.py ranks [world=8] [n_reads=100000] [len=4096] [files=8] [mode=global|chunk]"""
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
                                {f"{fi:03d}-{i:07d}": base[(i % 63) * length: (i % 63) * length + length] for i in range(k)})
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
    rank, _, world = env_rank_world()
    args._lm_loaded = False
    with open(os.path.join(scratch, "files.json")) as f:
        sources = [fast5.Fast5Source(p) for p in json.load(f)]
    sys.stdout = open(os.devnull, "w")      # (the per-read "Basecalled read ..." lines: the CLI prints them; a job redirects them)
    launch.run_rank(args, NullBackend(), NullComm(), scratch, sources, rank, world)


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
        res = {"world": world, "mode": mode, "reads": n, "samples_per_read": length, "cores": os.cpu_count(), "seconds": round(dt, 2),
               "merge_after_last_rank_s": round(dt - t_ranks, 2), "records": written, "reads_per_s": round(n / dt),
               "M_samples_per_s": round(n * length / dt / 1e6, 1), "M_samples_per_s_per_rank": round(n * length / dt / 1e6 / world, 1),
               "fasta_files": len([x for x in os.listdir(out_dir) if x.startswith("reads-")])}
        print(json.dumps(res))
        return res
    finally:
        if keep is None:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--worker":
        worker(a[1], a[3:])
    elif a and a[0] == "read":
        bench_read(int(a[1]) if len(a) > 1 else 16384, int(a[2]) if len(a) > 2 else 4096)
    else:
        a = a[1:] if a and a[0] == "ranks" else a
        bench_ranks(int(a[0]) if len(a) > 0 else 8, int(a[1]) if len(a) > 1 else 100000, int(a[2]) if len(a) > 2 else 4096,
                    int(a[3]) if len(a) > 3 else 8, a[4] if len(a) > 4 else "global")
