#!/usr/bin/env python3
"""Forward only, the headline's batch (64 reads x 4096, chunk 1024 / step 512: stream + per-layer head tiles) or uniform windows, on
1..4 lanes of ONE context or on several contexts -- separates what the tile structure costs from what the pipeline around the forward
costs.  usage: fwd_lanes.py [mode=reads|windows] [lanes=2] [iters=80] [contexts=1] [n_reads=64]"""
import os, sys, threading, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "reads"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 80
nctx = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n_reads = int(sys.argv[5]) if len(sys.argv) > 5 else 64
N, chunk, step = 4096, 1024, 512
pattern = [int(x) for x in os.environ["LANE_PATTERN"].split(",")] if os.environ.get("LANE_PATTERN") else None   # e.g. 0,0,0,1
w = weights.synthetic_weights(seed=1234)
rng = np.random.default_rng(0)
sig = np.clip(rng.normal(size=n_reads * N), -4, 4).astype(np.float32)
off = (np.arange(n_reads + 1, dtype=np.int64) * N)
bes, bufs = [], []
for _ in range(nctx):
    be = Backend(0)
    be.load_weights(w)
    be.set_conv_fuse(int(os.environ.get("RD_FUSE", "1")))
    d = be.dev_alloc(sig.nbytes)
    be.h2d(d, sig)
    bes.append(be)
    bufs.append(d)
nW = n_reads * N // chunk      # windows mode: the same samples as uniform windows


def loop(i, n):
    rows = 0
    for k in range(n):
        if mode == "reads":
            rows = bes[i].forward_reads_resident(bufs[i], off, n_reads, chunk, step, "chunk", lane=(pattern[k % len(pattern)] if pattern else k % lanes))
        else:
            bes[i].forward_resident(bufs[i], nW, chunk)
            rows = nW * chunk
    bes[i].sync()
    return rows


rows = loop(0, 1)
for rep in range(3):
    ths = [threading.Thread(target=loop, args=(i, 2 * lanes)) for i in range(nctx)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    t0 = time.time()
    ths = [threading.Thread(target=loop, args=(i, iters)) for i in range(nctx)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.time() - t0
    per = dt / (nctx * iters)
    print(f"fuse={os.environ.get('RD_FUSE', '1')} prio={os.environ.get('RD_X_LANE_PRIO', '-')} pattern={os.environ.get('LANE_PATTERN', '-')} {mode} lanes={lanes} contexts={nctx} reads={n_reads}: {per * 1e3:.3f} ms per forward = {n_reads * N / per / 1e6:.2f} M samples/s "
          f"({rows} rows evaluated at the head layer)", flush=True)
for be in bes:
    be.close()
