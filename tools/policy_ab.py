#!/usr/bin/env python3
"""Group policy of the reads-level pipeline (pipe_reads.hip): self-calibrated (product library) against the round-3 constants (an
experiment build with RD_NO_CALIB=1), global decode with the 4^11-row LM, per matrix-product mode and batch shape.
usage: policy_ab.py [W=10] [math=glibc]     (RADIAN_HIP_LIB=<experiment build> RD_NO_CALIB=1 for the constants)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from radian_amd import Backend, synthetic, weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
math = sys.argv[2] if len(sys.argv) > 2 else "glibc"
tag = "constants" if os.environ.get("RD_NO_CALIB") else "calibrated"
be = Backend(0)
be.load_weights(bench.soft_head_weights())
be.load_lm(np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11), 11)
be.set_decode_math(math)
for prec in ("fp32", "bf16x3", "f16x3"):
    be.pipe_flush()
    be.set_precision(prec)
    for n, L, steps in ((64, 4096, 96), (512, 4096, 24), (8, 100000, 48), (2048, 1500, 12)):
        bufs = []
        for b in range(2):
            norm = np.stack([synthetic.mad_normalise(r, 4) for r in synthetic.synthetic_reads(n, L, seed=1000 + b)]).astype(np.float32)
            d = be.dev_alloc(norm.nbytes)
            be.h2d(d, norm)
            bufs.append(d)
        off = np.arange(n + 1, dtype=np.int64) * L
        lab_off = np.ascontiguousarray(off[:-1])
        ring = [(np.zeros(n * L + 1, np.uint8), np.zeros(n, np.int32)) for _ in range(34)]

        def run(k):
            for i in range(k):
                lab, ln = ring[i % len(ring)]
                be.pipe_submit_reads_global(bufs[i % 2], off, n, 1024, 512, W, True, 0.5, 0.5, lab, lab_off, ln)
                if i >= 32:
                    be.pipe_progress(be.pipe_submitted() - 32)
            be.pipe_flush()
            be.sync()
        run(max(8, steps // 3))
        t0 = time.perf_counter()
        run(steps)
        dt = time.perf_counter() - t0
        pol = " | ".join(f"{'part m=%d' % op if op else 'chip'}: {q['us_per_step']:.2f} us/step -> {q['rows_per_step']} rows/step" for op in (1, 2, 3, 0) for q in [be.pipe_policy(W, op, True)])
        print(f"{tag:10s} W={W} {math} {prec:7s} {n:5d} reads x {L:6d}: {steps * n * L / dt / 1e6:6.2f} M samples/s ({dt / steps * 1e3:.2f} ms per step)"
              f"   [{be.pipe_policy(W, 1, True)['ns_per_row']:.1f} ns/row | {pol}]", flush=True)
        for d in bufs:
            be.dev_free(d)
be.close()
