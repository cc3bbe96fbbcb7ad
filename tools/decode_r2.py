#!/usr/bin/env python3
"""What would two sequences per wave cost at 7 <= W <= 12?  Its instruction stream is (about) that of the one-wave kernel with two
candidates per lane (R = 2), which rd_set_decode_form(2) runs for ONE sequence: if that takes x times the R = 1 kernel's time, the
two-sequence form would advance 2 / x sequences per unit of issue.   usage: decode_r2.py [W=10]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.backend import RD_TIMER_DECODE

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
        d_w = be.dev_alloc(win.nbytes)
        be.h2d(d_w, win)
        d_p = be.dev_alloc(n * T * 5 * 4)
        be.forward_resident(d_w, n, T, d_p)
        valid = np.ascontiguousarray(valid_w, dtype=np.int32)
        labels = np.zeros((n, T), np.uint8)
        lens = np.zeros(n, np.int32)
        for form in ("auto", "lanes"):
            for math in ("fast", "glibc"):
                be.set_decode_form(form)
                be.set_decode_math(math)
                be.decode_resident(d_p, n, T, valid, W, labels, lens)
                be.timer_enable(RD_TIMER_DECODE, 8)
                for _ in range(3):
                    be.decode_resident(d_p, n, T, valid, W, labels, lens)
                t = be.timer_read(RD_TIMER_DECODE)
                be.timer_enable(RD_TIMER_DECODE, 0)
                ms = t["total_ms"] / max(1, t["launches"])
                print(f"n={n} W={W} ({form}, {math}): kernel {ms:.3f} ms ({valid.sum() / ms / 1e3:.1f} M timesteps/s, {ms * 1e3 / T:.2f} us per time step)", flush=True)
        be.dev_free(d_w)
        be.dev_free(d_p)
    be.close()
