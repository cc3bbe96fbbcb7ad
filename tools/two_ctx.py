#!/usr/bin/env python3
"""Experiment: the bench job through ONE context (forwards back to back on one stream) vs TWO contexts taking steps
alternately (two independent kernel chains, so one chain's partially filled last round overlaps the other's launches)."""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.preprocess import mad_normalise

READS, L, CHUNK, STEP, BEAM, GROUP = 64, 4096, 1024, 512, 10, 8
nW = READS * 8
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
for nctx in (1, 2, 3):
    bes = []
    for c in range(nctx):
        be = Backend(0)
        be.load_weights(weights.synthetic_weights(seed=1234))
        be.set_precision(prec)
        be.pipe_config(GROUP)
        data = []
        for b in range(4):
            reads = synthetic.synthetic_reads(READS, L, seed=b)
            norm = np.stack([mad_normalise(r, 4) for r in reads]).astype(np.float32)
            d = be.dev_alloc(norm.nbytes)
            be.h2d(d, norm)
            data.append(d)
        out = [(np.zeros((nW, CHUNK), np.uint8), np.full(nW, -1, np.int32)) for _ in range(2 * GROUP)]
        bes.append((be, data, out))
    read_off = np.arange(READS + 1, dtype=np.int64) * L

    def run(steps):
        for i in range(steps):
            be, data, out = bes[i % nctx]
            j = i // nctx
            lab, ln = out[j % len(out)]
            be.pipe_submit_reads(data[j % 4], read_off, READS, CHUNK, STEP, BEAM, lab, ln)
        for be, _, _ in bes:
            be.pipe_flush()
        for be, _, _ in bes:
            be.sync()

    run(6)
    t0 = time.perf_counter()
    n = 48
    run(n)
    el = time.perf_counter() - t0
    print(f"{prec} contexts={nctx}: {n * READS * L / el / 1e6:.2f} M samples/s, {el / n * 1e3:.3f} ms/step")
    for be, _, _ in bes:
        be.close()
