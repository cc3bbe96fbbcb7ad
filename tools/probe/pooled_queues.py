#!/usr/bin/env python3
"""Do the pooled CU-masked streams of a long-lived process slow its blocking calls?  (round 6: inside `pytest -m gpu` tests/test_gpu_policy.py measured
57-80 ns per forward row with HIP-event timers around blocking calls, 27-36 in a fresh process.)  The independent figure of tools/policy_probe.py before and
after the process has used decode partitions of 1 / 4 / 8 / 12 / 16 CUs per XCD on several contexts (each size leaves its masked streams in the pool: destroying a
CU-masked stream can hang the runtime, forward.hip).  usage (gpurun): python tools/probe/pooled_queues.py"""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights

print("fresh process:", pp.independent_only("fp32", 10), flush=True)
reads = pp.reads_of(6, 20000, 5)
for rep in range(2):
    for part in (1, 4, 8, 12, 16):
        be = Backend(0)
        be.load_weights(weights.synthetic_weights(seed=1234))
        be.set_decode_partition(part)
        be.pipe_set_lanes(3)
        t = [be.pipe_submit_raw("global", reads, 4, 1024, 512, 10, False) for _ in range(6)]
        be.pipe_flush()
        be.close()
    print(f"after {(rep + 1) * 5} contexts with partitions of 1 / 4 / 8 / 12 / 16 CUs:", pp.independent_only("fp32", 10), flush=True)
