#!/usr/bin/env python3
"""One alternating-batches stream (tools/policy_probe.py) on its own, for a kernel trace:
   rocprofv3 --kernel-trace -d gpurun_out/alt -- python3 tools/policy_alt_run.py fp32 10 ; tools/pipe_trace_summary.py gpurun_out/alt
prints samples/s and the policy's figures."""
import json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 240
bg = None
if len(sys.argv) > 4 and sys.argv[4] == "load":
    import subprocess
    bg = subprocess.Popen([sys.executable, os.path.join(R, "tools", "policy_probe.py"), "--load-worker", "240"], stdout=subprocess.PIPE, text=True)
    assert "running" in bg.stdout.readline()
be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
be.set_precision(prec)
if os.environ.get("FORM"):
    be.set_decode_form(os.environ["FORM"])
short, long_ = pp.reads_of(64, 4096, 1), pp.reads_of(6, 40960, 2) + pp.reads_of(4, 4096, 3)
kind = os.environ.get("STREAM", "alternating")       # alternating | long | short
batches = {"alternating": [short, long_], "long": [long_], "short": [short]}[kind]
pp.stream(be, batches, W, 80)
r = [pp.stream(be, batches, W, n) for _ in range(3)]
print(json.dumps({"stream": kind, "alternating_samples_per_s": r, "stats": be.pipe_stats(), "policy": {m: be.pipe_policy(W, m) for m in (0, 1, 2, 3)}}))
be.close()
if bg is not None:
    bg.terminate()
    bg.wait()
