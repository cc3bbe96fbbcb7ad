#!/usr/bin/env python3
"""Is the bench's timed region bound by the host's kernel submission?  Times every rd_pipe_submit_reads call of a 20-step region on the
host, the flush, and the whole region (the bench's default configuration, fast to run)."""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
CHUNK, STEP, READ_LEN, N = 1024, 512, 4096, 64
dn = []
for b in range(4):
    reads = synthetic.synthetic_reads(N, READ_LEN, seed=b)
    norm = np.stack([synthetic.mad_normalise(r, 4) for r in reads]).astype(np.float32)
    d = be.dev_alloc(norm.nbytes); be.h2d(d, norm); dn.append(d)
read_off = np.arange(N + 1, dtype=np.int64) * READ_LEN
be.pipe_config(8); be.pipe_set_lanes(2)
out = [(np.zeros((512, CHUNK), np.uint8), np.full(512, -1, np.int32)) for _ in range(16)]
def sub(i):
    lab, ln = out[i % 16]
    be.pipe_submit_reads(dn[i % 4], read_off, N, CHUNK, STEP, 10, lab, ln)
for rep in range(3):
    for i in range(3): sub(i)
    be.pipe_flush(); be.sync()
    t0 = time.perf_counter(); ts = []
    for i in range(20):
        a = time.perf_counter(); sub(i); ts.append(time.perf_counter() - a)
    a = time.perf_counter(); be.pipe_flush(); be.sync(); fl = time.perf_counter() - a
    el = time.perf_counter() - t0
    ts = np.array(ts) * 1e3
    print(f"region {el*1e3:.1f} ms ({20*N*READ_LEN/el/1e6:.2f} M samples/s); submit calls: sum {ts.sum():.1f} ms, median {np.median(ts):.2f}, max {ts.max():.2f}; flush+sync {fl*1e3:.1f} ms")
    print("   per call ms:", " ".join(f"{t:.1f}" for t in ts))
