#!/usr/bin/env python3
"""Beam search, GPU vs the C oracle, over many random batches (the generator of tests/test_gpu_decode_stress.py): counts the
sequences whose labeling differs.  usage: fuzz_decode.py [rounds] [seed] [max rows per sequence = 400] [sequences per round = 800]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from radian_amd import Backend
from oracle import oracle                      # checker (test infrastructure)
from test_gpu_decode_stress import _mats



def run(rounds=20, seed=0, tmax=400, nseq=800, be=None, log=print):
    """-> (sequences compared, sequences whose labeling differs from the oracle's)"""
    rng = np.random.default_rng(seed)
    own = be is None
    if own:
        be = Backend(0)
    t0 = time.time()
    total = bad_total = 0
    try:
        for r in range(rounds):
            kind = ["flat", "peaky", "blocky"][r % 3]
            dtype = np.float32 if rng.random() < 0.5 else np.float64
            use_lm = dtype == np.float64 and rng.random() < 0.5
            k = int(rng.integers(1, 6))
            table = rng.dirichlet([0.2] * 4, size=4 ** k) if use_lm else None
            be.load_lm(table, k if use_lm else 0)
            mats, off, lens = _mats(rng, nseq, tmax, kind, dtype)
            W = int(rng.choice([1, 2, 5, 6, 7, 9, 10, 12, 13, 25, 26, 40, 51, 52, 64, 65, 90, 100, 127, 128, 129, 160, 200, 256, 257]))   # (65 ... 128: the five-wave shape of round 6; above: decode_wide.hip)
            s_thr, r_thr = float(rng.choice([0.0, 0.5, 0.8])), float(rng.choice([0.5, 0.9, 2.0]))
            form = str(rng.choice(["auto", "waves", "lanes", "two", "one", "queue"]))     # launch shape (rd_set_decode_form): waves / lanes for W > 12, two / one for W <= 12, queue: the work-queue kernel
            be.set_decode_form(form)
            math = str(rng.choice(["fast", "glibc"]))              # arithmetic of log / logaddexp (rd_set_decode_math)
            be.set_decode_math(math)
            if use_lm:
                got = be.decode_batch(mats, off, lens, W, use_lm=True, s_threshold=s_thr, r_threshold=r_thr)
                exp = oracle.beam_search_batch(mats, off, lens, W, table, s_thr, r_thr, k)
            else:
                got = be.decode_batch(mats, off, lens, W)
                exp = oracle.beam_search_batch(mats, off, lens, W)
            bad = [i for i in range(len(lens)) if not np.array_equal(got[i], exp[i])]
            total += len(lens)
            bad_total += len(bad)
            log(f"round {r}: {kind} {dtype.__name__} W={W} {form} {math} lm={'k=%d' % k if use_lm else 'no'}: {len(bad)} of {len(lens)} differ ({time.time() - t0:.0f}s)")
    finally:
        be.set_decode_form("auto")
        be.set_decode_math("glibc")
        if own:
            be.close()
    return total, bad_total


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    total, bad_total = run(*a, log=lambda m: print(m, flush=True))
    print(f"done: {bad_total} of {total} sequences differ")
    sys.exit(1 if bad_total else 0)
