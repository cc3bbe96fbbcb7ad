#!/usr/bin/env python3
"""Diagnostic (GPU box): for coarse-grid probabilities, find GPU-vs-oracle beam-search mismatches and show that each
starts at a time step where two of the oracle's beams around the pruning boundary have pr_total within a few ulp."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from radian_amd import Backend
from oracle import oracle as orc
import test_gpu_decode_stress as T

be = Backend(0)
rng = np.random.default_rng(123)
found = 0
for trial in range(20):
    mats, off, lens = T._mats(rng, 600, 300, "quant", np.float32)
    W = 2
    got = be.decode_batch(mats, off, lens, W)
    exp = orc.beam_search_batch(mats, off, lens, W)
    for i in range(len(lens)):
        if np.array_equal(got[i], exp[i]):
            continue
        m = mats[off[i]:off[i] + lens[i]]
        # first prefix length at which the two decoders disagree
        n = len(m)
        offs = np.arange(n, dtype=np.int64) * 0
        pl = np.arange(1, n + 1, dtype=np.int32)
        g = be.decode_batch(m, offs, pl, W)
        first = next(t for t in range(n) if not np.array_equal(g[t], orc.beam_search_labels(m[:t + 1], W)[0]))
        # oracle beam list after `first` steps (before the diverging step's pruning): look for near-ties
        _, fin = orc.beam_search_labels(m[:first + 1], W, max_final=12)
        tots = [f[1] for f in fin]
        gaps = [abs(tots[j] - tots[j + 1]) / max(1.0, abs(tots[j])) for j in range(len(tots) - 1)]
        print(f"seq {i}: T={n}, first divergence at prefix {first + 1}; oracle final pr_total (top 4): {tots[:4]}; min relative gap among top entries {min(gaps[:3]):.2e}")
        found += 1
        if found >= 6:
            raise SystemExit
print("mismatches", found)
