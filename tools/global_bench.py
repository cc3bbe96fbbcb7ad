#!/usr/bin/env python3
"""Single-GPU rate of BASELINE configs[3] geometry (global decode, step 512, beam 10, 12-mer LM = 4^11 x 4 table) and of the
W=25 stress of configs[4] (chunk decode), unpipelined reads-level calls; for DESIGN.md."""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.preprocess import mad_normalise

be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
n_reads = 512
reads = synthetic.synthetic_reads(n_reads, 4096, seed=9)
sigs = [mad_normalise(r, 4).astype(np.float32) for r in reads]
rng = np.random.default_rng(0)
table = rng.dirichlet([0.3] * 4, size=4 ** 11)
def rate(fn, reps=3):
    fn()
    t0 = time.time()
    for _ in range(reps):
        fn()
    return n_reads * 4096 * reps / (time.time() - t0) / 1e6
print("global step512 W10 no LM : %.2f M samples/s" % rate(lambda: be.basecall_reads_global(sigs, 1024, 512, 10, False)))
be.load_lm(table, 11)
print("global step512 W10 LM k11: %.2f M samples/s" % rate(lambda: be.basecall_reads_global(sigs, 1024, 512, 10, True, 0.5, 0.5)))
print("global step128 W6  LM k11: %.2f M samples/s" % rate(lambda: be.basecall_reads_global(sigs, 1024, 128, 6, True, 0.5, 0.5)))
be.load_lm(None, 0)
print("chunk  step512 W25       : %.2f M samples/s" % rate(lambda: be.basecall_reads_chunk(sigs, 1024, 512, 25)))
print("chunk  step512 W10       : %.2f M samples/s" % rate(lambda: be.basecall_reads_chunk(sigs, 1024, 512, 10)))
be.set_precision("f16x3")
print("global step512 W10 no LM f16x3: %.2f M samples/s" % rate(lambda: be.basecall_reads_global(sigs, 1024, 512, 10, False)))
