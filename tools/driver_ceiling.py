"""Where does the host driver (radian_amd.basecall.run) stop keeping up?  The same end-to-end leg in the faster matrix-product modes:
the device finishes a batch sooner, the host work per read stays.  usage: python tools/driver_ceiling.py [n_reads]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from radian_amd import weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
w = weights.synthetic_weights(seed=1234)
for mode in ("chunk", "global"):
    for prec in ("fp32", "bf16x3", "f16x3"):
        cli = ["--decode-type", mode, "--step-size", "512", "--beam-width", "10", "--rna-model", "None", "--precision", prec]
        r = bench.driver_leg(0, None, cli, np.full(n, 4096, dtype=np.int64), 1, w, desc="ceiling")
        print(f"{mode:6s} {prec:7s}: {r['value'] / 1e6:6.2f} M samples/s ({r['seconds']:.2f} s)", flush=True)
