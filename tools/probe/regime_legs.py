#!/usr/bin/env python3
"""bench.py's global-mode driver legs with the decode partition as the library chooses it and with it off (every CU; `--decode-partition 0`).
usage (gpurun): python tools/probe/regime_legs.py [partition,... = -1,0]"""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
import bench
from radian_amd import weights

parts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 0]
table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11)
soft = bench.soft_head_weights()
w0 = weights.synthetic_weights(seed=1234)
legs = (("reference_defaults (ragged, W 6, LM, soft head)", ["--rna-threshold", "0.5"], bench.ragged_lengths(16384, 72), soft, (table, 11)),
        ("long_reads (100 k samples, W 6, LM, soft head)", ["--rna-threshold", "0.5"], np.full(2048, 100000, dtype=np.int64), soft, (table, 11)),
        ("global W 10 LM, 4096-sample reads, He-normal head", ["--step-size", "512", "--beam-width", "10", "--rna-threshold", "0.5"], np.full(16384, 4096, dtype=np.int64), w0, (table, 11)),
        ("global W 10 no LM, ragged, soft head", ["--step-size", "512", "--beam-width", "10"], bench.ragged_lengths(16384, 73), soft, None),
        ("global W 25 LM, ragged, soft head", ["--step-size", "512", "--beam-width", "25", "--rna-threshold", "0.5"], bench.ragged_lengths(8192, 74), soft, (table, 11)))
for name, cli, lens, wf, lm in legs:
    row = []
    for part in parts:
        r = bench.driver_leg(0, None, cli + ["--decode-partition", str(part)], lens, 70002, wf, lm=lm, desc=name)
        row.append(f"partition {part:2d}: {r['value'] / 1e6:6.2f} M")
    print(f"{name:52s} " + "   ".join(row), flush=True)
