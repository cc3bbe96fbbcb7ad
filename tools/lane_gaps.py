#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace directory of a pipelined run: the pauses between consecutive conv launches on one queue
(end of one -> start of the next), and what else started on the device inside the long ones.  usage: lane_gaps.py <dir>"""
import csv, glob, sys, collections
p = glob.glob(sys.argv[1].rstrip("/") + "/*/*_kernel_trace.csv")[0]
ev = []
for r in csv.DictReader(open(p)):
    n = r["Kernel_Name"]
    k = "head" if "tcn_gemm_kernel<2" in n else "conv" if "tcn_gemm_kernel" in n else "beam" if "beam_search" in n else "tcn_in" if "tcn_in" in n else n.split("(")[0][-28:]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Queue_Id"]))
ev.sort()
t0 = ev[len(ev) // 4][0]
ev = [e for e in ev if e[0] >= t0]          # skip the start-up
last = {}
gaps = []
for s, e, k, q in ev:
    if k in ("conv", "head", "tcn_in") and q in last and last[q][2] in ("conv", "tcn_in"):
        gaps.append((s - last[q][1], last[q][1], s, q, last[q][2], k))
    last[q] = (s, e, k)
g = sorted(x[0] for x in gaps)
print(f"{len(g)} consecutive forward launches on a queue; pause end -> start (us): p50 {g[len(g)//2]/1e3:.1f}, p90 {g[int(len(g)*.9)]/1e3:.1f}, p99 {g[int(len(g)*.99)]/1e3:.1f}, max {g[-1]/1e3:.1f}")
long_ = [x for x in gaps if x[0] > 50000]
print(f"{len(long_)} pauses > 50 us ({sum(x[0] for x in long_) / 1e6:.2f} ms of {(ev[-1][1] - t0) / 1e6:.1f} ms)")
for d, a, b, q, k0, k1 in long_[:12]:
    inside = collections.Counter(k for s, e, k, qq in ev if a <= s <= b and qq != q)
    running = collections.Counter(k for s, e, k, qq in ev if s < a and e > b)
    print(f"  {d / 1e3:7.1f} us on queue {q} ({k0} -> {k1}) at {(a - t0) / 1e6:8.3f} ms; started elsewhere meanwhile: {dict(inside)}; running throughout: {dict(running)}")
