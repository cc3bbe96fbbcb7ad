#!/usr/bin/env python3
"""The beam-search workspace budget at real scale (round 6): a chunk-mode group of 8 x 4096 windows at W = 100 needs 32 768 x 102 401 trie nodes x 20 B
= 67 GB -- beyond the context's 24-GiB budget, so the group's launch is cut into runs that share the workspace.  Labels must equal the blocking
per-batch calls' (one run each); prints device memory in use and the rate.  usage (gpurun): python tools/probe/trie_budget_scale.py [W=100]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic

W = int(sys.argv[1]) if len(sys.argv) > 1 else 100
be = Backend(0)
w = weights.synthetic_weights(seed=1234).copy()
w[-645:-5] *= np.float32(0.05)
be.load_weights(w)
batches = [list(synthetic.synthetic_reads(512, 4096, seed=40 + i)) for i in range(8)]
free0, total = be.mem_info()
t0 = time.time()
ref = [be.basecall_raw_chunk(b, 4, 1024, 512, W) for b in batches]
t1 = time.time()
free1, _ = be.mem_info()
print(f"blocking: 8 batches x 4096 windows at W = {W}: {8 * 512 * 4096 / (t1 - t0) / 1e6:.2f} M samples/s; device memory in use {(total - free1) / 2**30:.1f} GiB", flush=True)
be.pipe_flush()
be.pipe_config(8)
t0 = time.time()
tickets = [be.pipe_submit_raw("chunk", b, 4, 1024, 512, W) for b in batches]
be.pipe_flush()
t1 = time.time()
free2, _ = be.mem_info()
bad = 0
for (lab, st), t in zip(ref, tickets):
    got, st2 = t.result()
    assert np.array_equal(st, st2)
    for a, b in zip(lab, got):
        for x, y in zip(a, b):
            bad += not np.array_equal(x, y)
print(f"pipeline, one group of 32 768 windows: {8 * 512 * 4096 / (t1 - t0) / 1e6:.2f} M samples/s; device memory in use {(total - free2) / 2**30:.1f} GiB "
      f"(the trie of the whole group would be {32768 * (1 + W * 1024) * 20 / 2**30:.0f} GiB); windows that differ from the blocking calls: {bad}")
be.close()
sys.exit(1 if bad else 0)
