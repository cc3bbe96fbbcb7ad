#!/usr/bin/env python3
"""Kernel timeline of a pipelined run from a rocprofv3 --kernel-trace directory: one line per dispatch of the last groups
(kernel, queue, start, end, duration in ms) + the beam-search durations.  usage: pipe_trace_summary.py <dir> [rows]
(profiles/r03a_global_pipe_trace.txt, r03f_global_pipe_trace.txt)"""
import csv, glob, sys

d = sys.argv[1].rstrip("/")
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
p = glob.glob(d + "/*/*_kernel_trace.csv")[0]


def short(n):
    for key, s in (("tcn_gemm_kernel<2", "head"), ("tcn_gemm_kernel", "conv"), ("beam_search2", "beam_search2"), ("beam_search", "beam_search"),
                   ("assemble_batch", "assemble_batch"), ("tcn_in", "tcn_in"), ("mad_normalise", "mad_normalise"), ("traceback", "traceback")):
        if key in n:
            return s
    return n.split("(")[0][-24:]


ev = []
for r in csv.DictReader(open(p)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
ev.sort()
t0 = ev[0][0]
q = {}
for e in ev:
    q.setdefault(e[3], f"q{len(q)}")
bs = [(e[1] - e[0]) / 1e6 for e in ev if e[2].startswith("beam_search")]
conv = [(e[1] - e[0]) / 1e6 for e in ev if e[2] == "conv"]
print(f"# {len(ev)} dispatches; beam search launches: {len(bs)}, durations ms min/median/max {min(bs):.2f} / {sorted(bs)[len(bs) // 2]:.2f} / {max(bs):.2f};"
      f" conv launches: {len(conv)}, median {sorted(conv)[len(conv) // 2]:.3f} ms")
print("# kernel  queue  start_ms  end_ms  dur_ms   (last dispatches)")
for s, e, n, qq in ev[-rows:]:
    print(f"{n:14s} {q[qq]:4s} {(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:8.3f}")
