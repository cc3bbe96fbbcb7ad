#!/usr/bin/env python3
"""Does a global-mode stream whose batches sit on both sides of the "few reads -> decode partition" threshold (n_reads <= part_seq_limit / 2:
192 reads at W <= 12) flip between the partitioned and the unpartitioned lanes and lose its groups?  Streams of ~4 M-sample batches (the CLI's
4096-unit batches) with 180 reads x 22 k samples (partition), 200 reads x 20 k (whole chip), and the two alternating.
usage (gpurun): python tools/probe/partition_threshold.py [W=10]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
be.set_decode_math("glibc")
a = pp.reads_of(180, 22000, 1)
b = pp.reads_of(200, 20000, 2)
pp.stream(be, [a, b], W, 12, window_rows=96 << 20)
for name, bs in (("180 reads x 22 k (partition)", [a]), ("200 reads x 20 k (whole chip)", [b]), ("alternating", [a, b]), ("alternating", [a, b])):
    s0 = be.pipe_stats()
    v = pp.stream(be, bs, W, 48, window_rows=96 << 20)
    s1 = be.pipe_stats()
    print(f"{name:32s} {v / 1e6:6.2f} M samples/s  groups {s1['launches'] - s0['launches']}  queue launches {s1['queue_launches'] - s0['queue_launches']}  limit closes {s1['limit_closes'] - s0['limit_closes']}", flush=True)
be.close()
