"""Beam search alone, 512 windows x 1024 rows: us per time step against the beam width, the launch shapes of decode.hip and the general
kernel of decode_wide.hip.  usage: decode_wscan.py [fast|glibc] [soft]      (soft: the soft head, ~200 bases per window)"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.backend import RD_TIMER_DECODE
T = 1024
be = Backend(0)
w = weights.synthetic_weights(seed=1234).copy()
if len(sys.argv) > 2 and sys.argv[2] == "soft":
    w[-645:-5] *= np.float32(0.05)
be.load_weights(w)
be.set_decode_math(sys.argv[1] if len(sys.argv) > 1 else "fast")   # the figures in DESIGN.md 4.4 are of the fast arithmetic
n = 512
reads = synthetic.synthetic_reads(n // 8, 4096, seed=5)
win, valid_w = synthetic.reads_to_windows(reads, T, 512)[:2]
win = np.ascontiguousarray(win, dtype=np.float32)
d_w = be.dev_alloc(win.nbytes); be.h2d(d_w, win)
d_p = be.dev_alloc(n * T * 5 * 4)
be.forward_resident(d_w, n, T, d_p)
valid = np.ascontiguousarray(valid_w, dtype=np.int32)
labels = np.zeros((n, T), np.uint8); lens = np.zeros(n, np.int32)
for W, form in ((10,"auto"),(12,"auto"),(13,"waves"),(13,"lanes"),(20,"waves"),(25,"waves"),(26,"waves"),(40,"waves"),(51,"waves"),(51,"lanes"),(52,"auto"),(64,"auto"),(65,"auto"),(100,"auto"),(128,"auto"),(129,"auto"),(200,"auto"),(256,"auto"),(257,"auto"),(1024,"auto"),(10,"queue")):   # (65 ... 128: five waves, the beam set in two halves -- round 6; above 128: decode_wide.hip; queue: 16 waves of resident workgroups)
    be.set_decode_form(form)
    be.decode_resident(d_p, n, T, valid, W, labels, lens)
    be.timer_enable(RD_TIMER_DECODE, 8)
    for _ in range(3):
        be.decode_resident(d_p, n, T, valid, W, labels, lens)
    t = be.timer_read(RD_TIMER_DECODE); be.timer_enable(RD_TIMER_DECODE, 0)
    ms = t["total_ms"] / max(1, t["launches"])
    print(f"W={W} {form}: {ms*1e3/T:.2f} us/step")
