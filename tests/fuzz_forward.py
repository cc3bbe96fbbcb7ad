#!/usr/bin/env python3
"""Forward pass, GPU vs the C oracle, over random model depths / dilation sets / window shapes, both precisions:
max |softmax difference| per case.  usage: fuzz_forward.py [cases] [seed]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights
from oracle import oracle                      # checker (test infrastructure)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
be = Backend(0)
worst = 0.0
t0 = time.time()
for c in range(cases):
    nb = int(rng.integers(1, 8))
    dil = tuple(int(d) for d in rng.choice([1, 2, 3, 4, 5, 8, 16, 32, 64], size=nb))
    T = int(rng.choice([1, 2, 31, 32, 33, 100, 127, 128, 129, 255, 500, 1000, 1024, 1025]))
    nW = int(rng.integers(1, 9))
    w = weights.synthetic_weights(seed=int(rng.integers(1 << 30)), dilations=dil, head_gain=float(rng.choice([1.0, 3.0])))
    be.load_weights(w, dil)
    x = np.clip(rng.normal(size=(nW, T)), -4, 4).astype(np.float32)
    ref = oracle.tcn_forward(w, x, dilations=dil, acc64=True)
    line = f"case {c}: blocks {nb} dil {dil} nW {nW} T {T}:"
    for prec in ("fp32", "f16x3", "bf16x3"):
        be.set_precision(prec)
        got = be.forward(x)
        err = float(np.abs(got.astype(np.float64) - ref).max())
        worst = max(worst, err)
        line += f" {prec} {err:.2e}"
        assert np.allclose(got.sum(axis=2), 1.0, atol=1e-5)
    print(line, f"({time.time() - t0:.0f}s)", flush=True)
print(f"done: worst |dp| {worst:.2e}")
sys.exit(0 if worst <= 1e-4 else 1)
