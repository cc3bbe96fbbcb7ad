#!/usr/bin/env python3
"""One very long read (default 1 000 000 samples) through the raw-reads entry points on the GPU, both decode types, against the
oracle's window-level pipeline (normalise -> windows -> forward -> assemble / per-window -> beam search): index widths, workspace
growth, the streamed forward over ~2000 windows.  Run by hand on the GPU box (the oracle forward takes a minute).
usage: diag_long_read.py [samples] [step] [beam width]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights, synthetic
from oracle import oracle                      # checker (test infrastructure)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W = int(sys.argv[3]) if len(sys.argv) > 3 else 6
CHUNK = 1024
w = weights.synthetic_weights(seed=1234).copy()
w[-645:-5] *= np.float32(0.05)                 # soft head: long labelings
raw = synthetic.synthetic_reads(1, N, seed=9)[0]
be = Backend(0)
be.load_weights(w)
rng = np.random.default_rng(2)
table = rng.dirichlet([0.3] * 4, size=4 ** 5)
be.load_lm(table, 5)
t0 = time.time()
got_g, st = be.basecall_raw_global([raw], 4, CHUNK, STEP, W, True, 0.5, 0.5)
t1 = time.time()
got_c, st2 = be.basecall_raw_chunk([raw], 4, CHUNK, STEP, W)
t2 = time.time()
print(f"GPU: global {t1 - t0:.2f}s ({len(got_g[0])} labels), chunk {t2 - t1:.2f}s ({len(got_c[0])} fragments)", flush=True)
assert not st.any() and not st2.any()
norm = oracle.mad_normalise(raw, 4)
win, pad = oracle.get_windows(norm, CHUNK, STEP)
win = win.astype(np.float32)
t0 = time.time()
probs = be.forward(win)                        # window-level GPU forward (checked against the oracle forward on a sample below)
ref = oracle.tcn_forward(w, win[:: max(1, len(win) // 24)][:24])
err = float(np.abs(probs[:: max(1, len(win) // 24)][:24] - ref).max())
print(f"window-level forward vs oracle on 24 windows: {err:.2e}", flush=True)
assert err <= 1e-4
mat = oracle.assemble_matrices(probs, pad, STEP)
exp_g = oracle.beam_search_batch(mat, [0], [mat.shape[0]], W, table, 0.5, 0.5, 5)[0]
valid = np.full(len(win), CHUNK, dtype=np.int32)
valid[-1] = CHUNK - pad
off = np.arange(len(win), dtype=np.int64) * CHUNK
exp_c = oracle.beam_search_batch(probs.reshape(-1, 5), off, valid, W)
print(f"oracle decode {time.time() - t0:.1f}s", flush=True)
assert np.array_equal(got_g[0], exp_g), "global labels differ"
bad = [i for i in range(len(win)) if not np.array_equal(got_c[0][i], exp_c[i])]
assert not bad, (bad[:5], len(bad))
print(f"ok: {N} samples, {len(win)} windows: global labeling ({len(exp_g)} bases) and every chunk fragment identical to the oracle's")
