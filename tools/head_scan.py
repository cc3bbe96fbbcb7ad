#!/usr/bin/env python3
"""Which synthetic head gives dRNA-like rows?  The bench's He-normal head saturates (~5 bases per 1024-row window), the soft head (last Dense
kernel x 0.05) gives ~200; nanopore direct-RNA signal carries ~25 bases per 1024 samples.  Scan (kernel scale, blank bias) of the last Dense
layer and print bases per window at W = 10 on the bench batch.  usage (gpurun): python tools/head_scan.py"""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic

be = Backend(0)
reads = synthetic.synthetic_reads(64, 4096, seed=1000)
norm = np.stack([synthetic.mad_normalise(r, 4) for r in reads]).astype(np.float32)
d = be.dev_alloc(norm.nbytes)
be.h2d(d, norm)
off = np.arange(65, dtype=np.int64) * 4096
lab = np.zeros((512, 1024), dtype=np.uint8)
ln = np.zeros(512, dtype=np.int32)
base = weights.synthetic_weights(seed=1234)
for scale in (1.0, 0.2, 0.1, 0.05):
    for bias in (0.0, 1.0, 2.0, 3.0, 4.0, 6.0):
        w = base.copy()
        w[-645:-5] *= np.float32(scale)
        w[-1] += np.float32(bias)
        be.load_weights(w)
        be.basecall_reads_chunk_resident(d, off, 64, 1024, 512, 10, lab, ln)
        print(f"scale {scale:5.2f} blank bias {bias:4.1f}: {ln.mean():7.1f} bases per window (min {ln.min()}, max {ln.max()})", flush=True)
be.close()
