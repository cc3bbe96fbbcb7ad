#!/usr/bin/env python3
"""EXPERIMENT (round 4): what could overlapping a layer's last round of workgroups with the next layer's first win?
Forward only (uniform 1024-row windows, nW chosen so that a launch is 5.75 rounds of 512 workgroups like the headline's), back to
back on one context or on two contexts from two host threads (= two lanes).  With RD_X_ALT_STREAMS=1 the library puts neighbouring
layers on two streams with NO ordering between them: results are garbage, the time is an upper bound for any scheme that lets
layer l+1 start while layer l drains.   usage: alt_streams.py [n_windows=368] [iters=60] [contexts=1]"""
import os, sys, threading, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights

nW = int(sys.argv[1]) if len(sys.argv) > 1 else 368
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T = 1024
w = weights.synthetic_weights(seed=1234)
x = np.random.default_rng(0).normal(size=(nW, T)).astype(np.float32)
bes, bufs = [], []
for _ in range(nctx):
    be = Backend(0)
    be.load_weights(w)
    d = be.dev_alloc(x.nbytes)
    be.h2d(d, x)
    bes.append(be)
    bufs.append(d)
probe = np.zeros(4, dtype=np.float32)


def loop(i, n):
    for _ in range(n):
        bes[i].forward_resident(bufs[i], nW, T)
    bes[i].d2h(probe, bufs[i])      # (a blocking copy on the context's stream: waits for its forwards)


for rep in range(3):
    ths = [threading.Thread(target=loop, args=(i, 10)) for i in range(nctx)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    t0 = time.time()
    ths = [threading.Thread(target=loop, args=(i, iters)) for i in range(nctx)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.time() - t0
    rows = nctx * iters * nW * T
    print(f"alt={os.environ.get('RD_X_ALT_STREAMS', '0')} contexts={nctx} nW={nW}: {dt / (nctx * iters) * 1e3:.3f} ms per forward, {rows / dt / 1e6:.2f} M rows/s "
          f"= {rows * 11 * 393216 / dt / 157.3e12:.4f} of the fp32-MFMA peak (conv FLOPs only)", flush=True)
for be in bes:
    be.close()
