#!/usr/bin/env python3
"""Per-launch conv times (HIP events) for the chunk plan (streams + heads) vs streams only, 64 reads x 4096."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from radian_amd.backend import RD_TIMER_CONV
from radian_amd.preprocess import mad_normalise
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
if len(sys.argv) > 1:
    be.set_precision(sys.argv[1])          # usage: layer_times.py [precision] [n_reads]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reads = synthetic.synthetic_reads(n_reads, 4096, seed=1)
sigs = [mad_normalise(r, 4).astype(np.float32) for r in reads]
for name, fn in (("chunk (streams + heads)", lambda: be.basecall_reads_chunk(sigs, 1024, 512, 1)),
                 ("global (streams only)", lambda: be.basecall_reads_global(sigs, 1024, 512, 1, False))):
    fn(); fn()
    be.timer_enable(RD_TIMER_CONV, 11 * 4)
    for _ in range(4):
        fn()
    t = be.timer_read(RD_TIMER_CONV)
    be.timer_enable(RD_TIMER_CONV, 0)
    print(f"{n_reads} reads, {name}: conv avg {t['total_ms']/t['launches']:.4f} ms per launch, {t['total_ms']/4:.3f} ms per forward, {t['flops']/t['total_ms']/1e9:.1f} TFLOP/s")
