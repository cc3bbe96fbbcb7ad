#!/usr/bin/env python3
"""Soak run of the one-rank CLI route: many ragged reads in several multi-read fast5 files through radian_amd.basecall.run (chunk mode,
then global mode with a 4^9-row LM, soft head so that labelings are long), with the process's resident memory, the device's used memory
and the rate printed every `every` reads -- a leak or a slow-down over a long job shows as a trend.  usage: soak.py [n_reads=120000] [every=20000] [route=run|main]
route=main drives radian_amd.basecall.main(argv) itself -- the command line's own route, writer and all (round 6: run() kept every
record of a job although the writer had written it; on_result, which this tool passed, hid that) -- and samples the resident memory
from a watcher thread every 2 s instead of from on_result."""
import os, resource, sys, tempfile, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import fast5, basecall, weights
from radian_amd.backend import Backend


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    route = sys.argv[3] if len(sys.argv) > 3 else "run"
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "in"))
    rng = np.random.default_rng(11)
    lens = np.clip(np.exp(rng.normal(np.log(4500.0), 0.6, size=n_reads)).astype(np.int64), 600, 40000)
    n_files = 6
    t0 = time.time()
    per = (n_reads + n_files - 1) // n_files
    for fi in range(n_files):
        lo, hi = fi * per, min(n_reads, (fi + 1) * per)
        reads = {f"{i:08d}-soak": np.round(rng.normal(500.0, 80.0, size=int(lens[i]))).astype(np.int16) for i in range(lo, hi)}
        fast5.write_multi_fast5(os.path.join(d, "in", f"f{fi}.fast5"), reads)
    total = int(lens.sum())
    print(f"{n_reads} reads, {total / 1e6:.0f} M samples in {n_files} files written in {time.time() - t0:.0f} s", flush=True)
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)
    k = 9
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** k)
    if route == "main":
        import threading
        for mode, extra in (("chunk", ["--step-size", "512", "--beam-width", "10"]), ("global", ["--step-size", "128", "--beam-width", "6"])):
            out = os.path.join(d, "out_main_" + mode)
            os.makedirs(out)
            stop = threading.Event()
            t_start = time.time()

            def watch():
                while not stop.wait(2.0):
                    n = sum(1 for f in os.listdir(out))
                    print(f"  main/{mode}: {time.time() - t_start:6.1f} s | ~{max(0, n - 1) * 1000:7d} reads written | host RSS {rss_mb():7.0f} MB",
                          file=sys.__stdout__, flush=True)
            th = threading.Thread(target=watch, daemon=True)
            th.start()
            so = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                basecall.main([os.path.join(d, "in"), out, "--decode-type", mode, "--sig-model", "synthetic:1234", "--sig-config", "none",
                               "--rna-model", "None"] + extra)
            finally:
                sys.stdout = so
                stop.set()
                th.join()
            dt = time.time() - t_start
            print(f"main/{mode}: {total / dt / 1e6:.2f} M samples/s over {dt:.1f} s incl. start-up; host RSS {rss_mb():.0f} MB, max RSS "
                  f"{resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MB", flush=True)
        return
    for mode, extra in (("chunk", ["--step-size", "512", "--beam-width", "10"]), ("global", ["--step-size", "128", "--beam-width", "6", "--context-len", str(k)])):
        out = os.path.join(d, "out_" + mode)
        os.makedirs(out)
        args = basecall.build_parser().parse_args([os.path.join(d, "in"), out, "--decode-type", mode, "--sig-model", "synthetic:1234",
                                                   "--sig-config", "none", "--rna-model", "None"] + extra)
        be = Backend(args.device)
        be.load_weights(w)
        if mode == "global":
            be.load_lm(table, k)
            args._lm_loaded = True
        state = {"n": 0, "samples": 0, "t": time.time(), "t0": time.time(), "bases": 0}
        free0, tot = be.mem_info()
        print(f"  {mode}: start | host RSS {rss_mb():7.0f} MB | device used {(tot - free0) / 1e6:8.0f} MB", flush=True)

        def on_result(key, rid, seq):
            state["n"] += 1
            state["bases"] += len(seq)
            if state["n"] % every == 0:
                now = time.time()
                free, _ = be.mem_info()
                print(f"  {mode}: {state['n']:7d} reads | {every / (now - state['t']):7.0f} reads/s in the last block | host RSS {rss_mb():7.0f} MB | "
                      f"device used {(tot - free) / 1e6:8.0f} MB | bases so far {state['bases']}", file=sys.__stdout__, flush=True)
                state["t"] = now
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.run(args, [be], on_result=on_result)
        finally:
            sys.stdout = so
        dt = time.time() - state["t0"]
        print(f"{mode}: {state['n']} reads, {total / dt / 1e6:.2f} M samples/s over {dt:.1f} s; host RSS {rss_mb():.0f} MB, max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MB", flush=True)
        be.close()


if __name__ == "__main__":
    main()
