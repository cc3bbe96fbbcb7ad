"""Host time of every rd_pipe_submit_raw_global call of a stream of small long-read batches (6 x 40960 + 4 x 4096 samples, W = 10): which
submits take longer than 20 ms, when, and how many groups had been launched (DESIGN_LOG.md round 5: the staging-event wait of the
uncovered mode).  usage (GPU box): python tools/submit_times.py"""
import json, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import policy_probe as pp
from radian_amd import Backend, weights
be = Backend(0); be.load_weights(weights.synthetic_weights(seed=1234))
long_ = pp.reads_of(6, 40960, 2) + pp.reads_of(4, 4096, 3)
pp.stream(be, [long_], 10, 80)
be.pipe_flush()
slow = []; tickets = []
t_start = time.perf_counter()
for i in range(200):
    t0 = time.perf_counter()
    tickets.append(be.pipe_submit_raw("global", long_, 4, 1024, 512, 10, False))
    dt = time.perf_counter() - t0
    if dt > 0.02: slow.append((i, round((t0 - t_start) * 1e3), round(dt * 1e3, 1), be.pipe_stats()["launches"]))
for t in tickets: t.result()
print("slow submits (index, at ms, took ms, launches so far):", slow[:30])
print("total", round((time.perf_counter() - t_start) * 1e3), "ms;", be.pipe_stats())
be.close()
