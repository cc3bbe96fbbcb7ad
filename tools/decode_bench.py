#!/usr/bin/env python3
"""Beam search alone (no concurrent forward): n windows x T time steps resident in HBM; HIP-event time per launch.
Two input distributions: the bench's (He-normal weights: saturated softmax rows, labelings a few bases long) and the same
model with the last Dense kernel x 0.05 (soft rows, mean entropy 0.85 nat: long labelings, many live beams, merges and
re-entries -- the decoder's hard case)."""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.backend import RD_TIMER_DECODE

T = 1024
for name, scale in (("bench weights (saturated rows)", 1.0), ("soft head x0.05", 0.05)):
    be = Backend(0)
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(scale)
    be.load_weights(w)
    print(f"== {name}")
    for n in (512, 4096):
        reads = synthetic.synthetic_reads(n // 8, 4096, seed=5)
        win, valid_w = synthetic.reads_to_windows(reads, T, 512)[:2]
        win = np.ascontiguousarray(win, dtype=np.float32)
        assert win.shape == (n, T)
        d_w = be.dev_alloc(win.nbytes)
        be.h2d(d_w, win)
        d_p = be.dev_alloc(n * T * 5 * 4)
        be.forward_resident(d_w, n, T, d_p)
        valid = np.ascontiguousarray(valid_w, dtype=np.int32)
        labels = np.zeros((n, T), np.uint8)
        lens = np.zeros(n, np.int32)
        for W, form, math in ((1, "one", "fast"), (1, "two", "fast"), (6, "one", "fast"), (6, "two", "fast"), (6, "one", "glibc"), (6, "two", "glibc"), (10, "auto", "fast"), (10, "auto", "glibc"), (25, "waves", "fast"),
                              (25, "lanes", "fast"), (25, "auto", "fast"), (25, "auto", "glibc")):
            be.set_decode_form(form)
            be.set_decode_math(math)
            be.decode_resident(d_p, n, T, valid, W, labels, lens)
            be.timer_enable(RD_TIMER_DECODE, 8)
            t0 = time.perf_counter()
            for _ in range(3):
                be.decode_resident(d_p, n, T, valid, W, labels, lens)
            wall = (time.perf_counter() - t0) / 3
            t = be.timer_read(RD_TIMER_DECODE)
            be.timer_enable(RD_TIMER_DECODE, 0)
            ms = t["total_ms"] / max(1, t["launches"])
            print(f"n={n} T={T} W={W} ({form}, {math}): kernel {ms:.3f} ms ({valid.sum() / ms / 1e3:.1f} M timesteps/s, {ms * 1e3 / T:.2f} us per time step), "
                  f"call {wall * 1e3:.2f} ms, mean len {lens.mean():.0f}")
        be.dev_free(d_w)
        be.dev_free(d_p)
    be.close()
