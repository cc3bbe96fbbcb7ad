"""One-context pipelined global decode (rd_pipe_submit_reads_global) at the bench's 64-read steps: samples/s against the
group size.  usage: python tools/global_pipe_bench.py [soft] [fast] [hashed] [f16] [partK] [nREADS] [gN ...] [W] """
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from radian_amd import Backend, synthetic, weights

def main():
    soft = "soft" in sys.argv
    W = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 10
    n, L = 64, 4096
    for a in sys.argv[1:]:
        if a.startswith('n') and a[1:].isdigit():
            n = int(a[1:])      # reads per step (default 64)
    be = Backend(0)
    be.load_weights(bench.soft_head_weights() if soft else weights.synthetic_weights(seed=1234))
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11)
    if "hashed" in sys.argv:
        be.load_lm_hashed(table, 11, 256)      # configs[4]: --context-len 256 over a 4^11-row table
    else:
        be.load_lm(table, 11)
    if "f16" in sys.argv:
        be.set_logits("f16")
    if "fast" in sys.argv:
        be.set_decode_math("fast")
    for a in sys.argv[1:]:
        if a.startswith("part"):
            be.set_decode_partition(int(a[4:]))
    bufs = []
    for b in range(4):
        norm = np.stack([synthetic.mad_normalise(r, 4) for r in synthetic.synthetic_reads(n, L, seed=1000 + b)]).astype(np.float32)
        d = be.dev_alloc(norm.nbytes); be.h2d(d, norm); bufs.append(d)
    off = np.arange(n + 1, dtype=np.int64) * L
    lab_off = np.ascontiguousarray(off[:-1])
    lag = int(os.environ.get("LAG", "32"))          # submits the caller lets the pipeline run ahead before it waits for results
    steps = int(os.environ.get("STEPS", "64"))
    ring = [(np.zeros(n * L + 1, np.uint8), np.zeros(n, np.int32)) for _ in range(lag + 8)]
    groups = [int(a[1:]) for a in sys.argv[1:] if a.startswith('g') and a[1:].isdigit()] or [1, 2, 3, 4, 6, 8, 16]
    for group in groups:
        be.pipe_flush(); be.pipe_config(group)
        def run(k):
            for i in range(k):
                lab, ln = ring[i % len(ring)]
                be.pipe_submit_reads_global(bufs[i % 4], off, n, 1024, 512, W, True, 0.5, 0.5, lab, lab_off, ln)
                if i >= lag:
                    be.pipe_progress(be.pipe_submitted() - lag)
            be.pipe_flush(); be.sync()
        run(8)
        t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
        print(f"group {group:2d}: {steps * n * L / dt / 1e6:6.2f} M samples/s  ({dt / steps * 1e3:.2f} ms per {n}-read step, {dt / steps / (n * L) * 1e9:.2f} ns per row)"
              f"   policy: {[(m, round(q['ns_per_row'], 1), round(q['us_per_step'], 2), q['rows_per_step']) for m in (1, 2, 3, 0) for q in [be.pipe_policy(W, m, True)]]}", flush=True)
    be.close()
main()
