#!/usr/bin/env python3
"""Does creating and destroying many contexts in one process slow it down?  (round 6: late in `pytest -m gpu`'s process blocking calls measured 2x
slower than in a fresh one.)  The independent figure of tools/policy_probe.py after 0 / 40 / 80 / 120 contexts that each load weights, basecall a few reads in
both decode types through the blocking calls and the pipeline, and close.  usage (gpurun): python tools/probe/context_churn.py"""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights

w = weights.synthetic_weights(seed=1234)
rng = np.random.default_rng(1)
print("fresh:", pp.independent_only("fp32", 10), flush=True)
n = 0
for block in range(3):
    for _ in range(40):
        be = Backend(0)
        be.load_weights(w)
        reads = pp.reads_of(int(rng.integers(2, 40)), int(rng.integers(500, 30000)), int(rng.integers(1 << 30)))
        be.basecall_raw_chunk(reads, 4, 1024, 512, 10)
        be.basecall_raw_global(reads, 4, 1024, 512, int(rng.choice([6, 10, 25, 100])), False)
        t = [be.pipe_submit_raw(m, reads, 4, 1024, 512, 10, False) if m == "global" else be.pipe_submit_raw(m, reads, 4, 1024, 512, 10) for m in ("global", "chunk", "global")]
        be.pipe_flush()
        be.close()
        n += 1
    free, total = Backend(0).mem_info()
    print(f"after {n} contexts:", pp.independent_only("fp32", 10), f"device memory in use {(total - free) / 2**30:.1f} GiB", flush=True)
