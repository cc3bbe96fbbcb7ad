#!/usr/bin/env python3
"""Where does the decode partition pay?  Global-mode streams of ~4 M-sample batches (the CLI's batch size) made of n reads of 4 M / n samples,
with the library's partition choice (few reads -> partition) and with the partition off.  usage (gpurun): python tools/probe/partition_sweep.py [W=10] [samples per batch = 4000000] [reads per batch, comma separated]"""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000      # samples per batch
SUBMITS = max(40, 160000000 // B)                            # ~160 M samples per stream
for n in [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (10, 20, 41, 90, 180):
    out = []
    for part in (-1, 0):
        be = Backend(0)
        be.load_weights(weights.synthetic_weights(seed=1234))
        be.set_decode_math("glibc")
        be.set_decode_partition(part)
        b = pp.reads_of(n, B // n, 1)
        pp.stream(be, [b], W, max(10, SUBMITS // 4), window_rows=128 << 20)
        s0 = be.pipe_stats()
        v = pp.stream(be, [b], W, SUBMITS, window_rows=128 << 20)
        s1 = be.pipe_stats()
        out.append(f"{'auto' if part < 0 else 'off '}: {v / 1e6:6.2f} M ({s1['launches'] - s0['launches']} groups)")
        be.close()
    print(f"W = {W}, {n:4d} reads x {B // n:7d} samples per batch   " + "   ".join(out), flush=True)
