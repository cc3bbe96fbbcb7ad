#!/usr/bin/env python3
"""Workload of tools/prof_decode.sh: a few beam-search launches over n windows x 1024 rows (soft = 0: bench weights, 1: the soft head,
2: SURVEY 8d's peaky rows -- softmax(4 N(0,1), blank + 2), no model involved: bench.py's secondary_decode_only_peaky)."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic

n, W, soft = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = 1024
be = Backend(0)
w = weights.synthetic_weights(seed=1234).copy()
if soft:
    w[-645:-5] *= np.float32(0.05)
be.load_weights(w)
if hasattr(be, "set_decode_math"):
    be.set_decode_math(sys.argv[4] if len(sys.argv) > 4 else "fast")   # profiles/r02d_*: the fast arithmetic (round 1's decoder has no other)
if len(sys.argv) > 5 and hasattr(be, "set_decode_form"):
    be.set_decode_form(sys.argv[5])        # "one" / "two": W <= 6 as one / two sequences per wave (round 3)
reads = synthetic.synthetic_reads(n // 8, 4096, seed=5)
win, valid = synthetic.reads_to_windows(reads, T, 512)[:2]
win = np.ascontiguousarray(win, dtype=np.float32)
d_w = be.dev_alloc(win.nbytes)
be.h2d(d_w, win)
d_p = be.dev_alloc(n * T * 5 * 4)
if soft == 2:
    be.h2d(d_p, synthetic.peaky_probs(n, T, seed=7))
    valid = np.full(n, T, dtype=np.int32)
else:
    be.forward_resident(d_w, n, T, d_p)
valid = np.ascontiguousarray(valid, dtype=np.int32)
labels = np.zeros((n, T), np.uint8)
lens = np.zeros(n, np.int32)
for _ in range(3):
    be.decode_resident(d_p, n, T, valid, W, labels, lens)
print(f"n={n} W={W} soft={soft} timesteps_per_launch={int(valid.sum())} mean_len={lens.mean():.1f}")
be.close()
