#!/usr/bin/env python3
"""Start-up cost of the reference's default RNA model route (basecall.py:28,48-57) at the REAL size: writes a synthetic JSON with
all 4^k k-label contexts (k = 11: 4 194 304 keys, ~330 MB -- the shape of models/rnamodel_12mer_pc.json), loads it with
radian_amd.lm.load_json and reports seconds per stage and the process's peak RSS.  No GPU.   usage: lm_load_bench.py [k=11] [dir=/tmp]"""
import json, os, resource, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radian_amd import lm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
n = 4 ** k
path = os.path.join(d, f"rnamodel_{k + 1}mer_synth.json")
if not os.path.exists(path):
    rng = np.random.default_rng(0)
    vals = rng.dirichlet([0.3] * 4, size=n)
    letters = np.array(list("ACGT"))
    t0 = time.time()
    with open(path, "w") as f:
        f.write("{")
        B = 1 << 16
        for lo in range(0, n, B):
            idx = np.arange(lo, min(n, lo + B))
            digits = (idx[:, None] >> (2 * np.arange(k - 1, -1, -1))) & 3
            keys = ["".join(r) for r in letters[digits]]
            f.write(("," if lo else "") + ",".join(f'"{c}": [{v[0]!r}, {v[1]!r}, {v[2]!r}, {v[3]!r}]' for c, v in zip(keys, vals[idx].tolist())))
        f.write("}")
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.0f} MB in {time.time() - t0:.1f} s", flush=True)
# the library's one-pass reader first (peak RSS is a high-water mark: this leg must run before the object-tree leg)
rss00 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
t0 = time.time()
nat = lm._load_json_native(path)
t_nat = time.time() - t0
rss_nat = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
print(json.dumps({"native_reader_s": round(t_nat, 2), "peak_rss_MB": round(rss_nat), "rss_before_MB": round(rss00), "recognised": nat is not None}), flush=True)
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
t0 = time.time()
with open(path) as f:
    raw = json.load(f)
t1 = time.time()
table, kk = lm.table_from_dict(raw)
t2 = time.time()
del raw
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
print(json.dumps({"k": kk, "contexts": int(table.shape[0]), "file_MB": round(os.path.getsize(path) / 1e6), "json_load_s": round(t1 - t0, 2),
                  "table_from_dict_s": round(t2 - t1, 2), "total_s": round(t2 - t0, 2), "peak_rss_MB": round(rss), "rss_before_MB": round(rss0),
                  "missing": lm.n_missing(table), "native_equals_standard": bool(nat is not None and nat[1] == kk and nat[0].tobytes() == table.tobytes())}))
