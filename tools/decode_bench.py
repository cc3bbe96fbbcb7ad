#!/usr/bin/env python3
"""Beam search alone (no concurrent forward): n windows x T time steps resident in HBM, W = 10; HIP-event time per launch."""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.backend import RD_TIMER_DECODE
from radian_amd.preprocess import mad_normalise

be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
T = 1024
for n in (512, 4096):
    # the bench's input distribution: Gaussian int16 reads, MAD-normalised, cut into windows (near-uniform softmax rows:
    # long labelings, many live beams -- SURVEY 8d's worst case for the beam search)
    reads = synthetic.synthetic_reads(n // 8, 4096, seed=5)
    w, valid_w = synthetic.reads_to_windows(reads, T, 512)[:2]
    w = np.ascontiguousarray(w, dtype=np.float32)
    assert w.shape == (n, T)
    d_w = be.dev_alloc(w.nbytes)
    be.h2d(d_w, w)
    d_p = be.dev_alloc(n * T * 5 * 4)
    be.forward_resident(d_w, n, T, d_p)
    valid = np.ascontiguousarray(valid_w, dtype=np.int32)
    labels = np.zeros((n, T), np.uint8)
    lens = np.zeros(n, np.int32)
    for W in (10, 25):
        be.decode_resident(d_p, n, T, valid, W, labels, lens)
        be.timer_enable(RD_TIMER_DECODE, 8)
        t0 = time.perf_counter()
        for _ in range(3):
            be.decode_resident(d_p, n, T, valid, W, labels, lens)
        wall = (time.perf_counter() - t0) / 3
        t = be.timer_read(RD_TIMER_DECODE)
        be.timer_enable(RD_TIMER_DECODE, 0)
        ms = t["total_ms"] / max(1, t["launches"])
        print(f"n={n} T={T} W={W}: kernel {ms:.3f} ms ({valid.sum() / ms / 1e3:.1f} M timesteps/s, {ms * 1e3 / T:.2f} us per time step), call {wall * 1e3:.2f} ms, mean len {lens.mean():.0f}")
    be.dev_free(d_w)
    be.dev_free(d_p)
