#!/usr/bin/env python3
"""Differential fuzz of the reads-level (streamed) paths against the window-level paths on the GPU: random geometries,
ragged read sets, both decode types, both precisions.  Labels must be identical.  usage: fuzz_reads.py [iterations] [seed]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, weights
from radian_amd.preprocess import get_windows



def windows(sig, chunk, step):
    w, pad = get_windows(sig, chunk, step)
    valid = np.full(w.shape[0], chunk, dtype=np.int32)
    valid[-1] = chunk - pad
    return w.astype(np.float32), valid, pad


def run(iters=200, seed=0, max_len=6000, log=print):
    """-> (geometries compared, geometries whose streamed labels differ from the window-level labels)"""
    rng = np.random.default_rng(seed)
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    k = 3
    be.load_lm(rng.dirichlet([0.3] * 4, size=4 ** k), k)
    t0 = time.time()
    n_fail = 0
    for it in range(iters):
        chunk = int(rng.choice([64, 128, 256, 300, 512, 1000, 1024, 2048]))
        step = int(rng.integers(1, chunk + 1)) if rng.random() < 0.5 else int(rng.choice([chunk, chunk // 2, max(1, chunk // 8), max(1, chunk - 252), max(1, chunk - 253)]))
        if step < chunk // 16:
            step = max(step, chunk // 16)          # bound the window count
        n_reads = int(rng.integers(1, 7))
        lengths = [int(rng.choice([1, 2, chunk - 1, chunk, chunk + 1, chunk + step, int(rng.integers(1, max_len))])) for _ in range(n_reads)]
        W = int(rng.choice([1, 4, 10, 25]))
        prec = "f16x3" if rng.random() < 0.3 else "fp32"
        be.set_precision(prec)
        sigs = [np.clip(rng.normal(size=n), -4, 4).astype(np.float32) for n in lengths]
        mode = "chunk" if rng.random() < 0.5 else "global"
        ok = True
        try:
            if mode == "chunk":
                got = be.basecall_reads_chunk(sigs, chunk, step, W)
                for r, sig in enumerate(sigs):
                    w, valid, _ = windows(sig, chunk, step)
                    exp = be.basecall_chunk(w, valid, W)
                    ok &= len(got[r]) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got[r], exp))
            else:
                use_lm = bool(rng.random() < 0.5)
                got = be.basecall_reads_global(sigs, chunk, step, W, use_lm, 0.5, 0.5)
                wins, offs, pads = [], [0], []
                for sig in sigs:
                    w, _, pad = windows(sig, chunk, step)
                    wins.append(w)
                    offs.append(offs[-1] + w.shape[0])
                    pads.append(pad)
                exp = be.basecall_global(np.concatenate(wins), np.array(offs, dtype=np.int32), np.array(pads, dtype=np.int32), step, W, use_lm, 0.5, 0.5)
                ok &= all(np.array_equal(a, b) for a, b in zip(got, exp))
        except Exception as e:   # report and go on
            ok = False
            log(f"EXC {type(e).__name__} {e}")
        if not ok:
            n_fail += 1
            log(f"MISMATCH it={it} mode={mode} prec={prec} chunk={chunk} step={step} W={W} lengths={lengths}")
        if it % 50 == 49:
            log(f"{it + 1} iterations, {n_fail} failures, {time.time() - t0:.0f}s")
    be.close()
    return iters, n_fail


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    iters, n_fail = run(*a, log=lambda m: print(m, flush=True))
    print(f"done: {iters} iterations, {n_fail} failures")
    sys.exit(1 if n_fail else 0)
