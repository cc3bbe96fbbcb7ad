import os, sys, json, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import policy_probe as pp
from radian_amd import Backend, weights
W = int(sys.argv[1]) if len(sys.argv) > 1 else 25
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
be.set_decode_math("glibc")
short = pp.reads_of(64, 4096, 1)
long_ = pp.reads_of(6, 40960, 2) + pp.reads_of(4, 4096, 3)
rag = pp.ragged_batches(12, 262144, 7)
pp.stream(be, [short, long_], W, 80)
for name, b in (("short", [short]), ("long", [long_]), ("ragged", rag), ("ragged", rag), ("ragged", rag)):
    s0 = be.pipe_stats()
    v = pp.stream(be, b, W, 360)
    s1 = be.pipe_stats()
    print(name, round(v / 1e6, 2), "stats delta", {k: s1[k] - s0[k] for k in s1}, {m: (round(be.pipe_policy(W, m)["us_per_step"], 2), be.pipe_policy(W, m)["rows_per_step"]) for m in (0, 1, 2, 3)}, round(be.pipe_policy(W, 1)["ns_per_row"], 1), flush=True)
print("ragged longest per batch", [max(len(r) for r in b) for b in rag], "reads", [len(b) for b in rag])
be.close()
