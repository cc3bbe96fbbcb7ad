import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from radian_amd import Backend, weights
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
rng = np.random.default_rng(1)
def batch(lens):
    return [np.round(rng.normal(500.0, 80.0, size=int(n))).astype(np.int16) for n in lens]
uni = batch([4096] * 512)
for name, mk in (("uniform 512x4096", lambda: uni), ("ragged ~2.1M samples", lambda: batch(rng.integers(1500, 6700, size=512)))):
    for mode in ("chunk", "global"):
        ts = []
        for it in range(6):
            b = mk()
            tot = sum(len(x) for x in b)
            t0 = time.perf_counter()
            if mode == "chunk":
                be.basecall_raw_chunk(b, 4, 1024, 512, 10)
            else:
                be.basecall_raw_global(b, 4, 1024, 128, 6, False, 0.5, 0.5)
            ts.append((time.perf_counter() - t0, tot))
        t, tot = min(ts[1:])
        print(f"{name} {mode}: {t * 1e3:.1f} ms per call, {tot / t / 1e6:.2f} M samples/s")
