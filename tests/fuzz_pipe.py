#!/usr/bin/env python3
"""Differential fuzz of the reads-level pipeline (rd_pipe_submit_raw_global / _chunk, rd_pipe_progress; csrc/pipe_reads.hip) against
the blocking entry points on the GPU: random sequences of submits on ONE context -- decode type, geometry, beam width, LM on / off,
thresholds, f16 logits, lanes, group size, decode partition and precision change between rounds; read sets are ragged (1 .. 9000
samples, constant reads, reads around the window length); progress is polled at random; blocking calls are interleaved.
Every delivered batch must equal the blocking call's labels and status.   usage: fuzz_pipe.py [rounds] [seed]"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import Backend, synthetic, weights



def same(mode, a, b, status):
    if mode == "global":
        return all(st != 0 or np.array_equal(x, y) for x, y, st in zip(a, b, status))
    return all(st != 0 or (len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y))) for x, y, st in zip(a, b, status))


def run(rounds=100, seed=0, max_len=9000, max_reads=30, log=print):
    """-> (batches delivered, mismatches against the blocking entry points)"""
    rng = np.random.default_rng(seed)
    be = Backend(0)
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)
    be.load_weights(w)
    be.load_lm(rng.dirichlet([0.3] * 4, size=4 ** 3), 3)

    def read_set(chunk, step):
        n = int(rng.integers(1, max_reads))
        out = []
        for _ in range(n):
            L = int(rng.choice([1, 2, chunk - 1, chunk, chunk + 1, chunk + step, 3 * step, int(rng.integers(1, max_len))]))
            if rng.random() < 0.05:
                out.append(np.full(max(L, 2), 7, dtype=np.int16))
            else:
                out.append(synthetic.synthetic_reads(1, L, seed=int(rng.integers(1 << 30)))[0])
        return out

    t0 = time.time()
    n_fail = n_batches = 0
    for rd in range(rounds):
        be.pipe_flush()
        be.pipe_config(int(rng.integers(1, 7)))
        be.pipe_set_lanes(int(rng.integers(1, 5)))
        be.set_decode_partition(int(rng.choice([-1, 0, 1, 4])))
        be.set_logits("f16" if rng.random() < 0.2 else "f32")
        be.set_precision("bf16x3" if rng.random() < 0.15 else "fp32")
        chunk = int(rng.choice([256, 512, 1024]))
        plan = []
        for _ in range(int(rng.integers(2, 9))):
            mode = "chunk" if rng.random() < 0.4 else "global"
            step = int(rng.choice([chunk, chunk // 2, chunk // 4, max(1, chunk - 252), max(1, chunk - 253), int(rng.integers(chunk // 8, chunk + 1))]))
            # (mostly the lane kernels' common widths; now and then their upper forms -- W 26..51, 52..64 -- and the general kernel above 64)
            W = int(rng.choice([1, 3, 6, 7, 10, 12, 13, 25])) if rng.random() < 0.9 else int(rng.choice([40, 52, 64, 65, 100, 128, 130, 256, 260]))
            lm = bool(mode == "global" and rng.random() < 0.5)
            thr = (float(rng.choice([0.0, 0.3, 0.6])), float(rng.choice([0.2, 0.9, 5.0])))
            plan.append((mode, step, W, lm, thr, read_set(chunk, step)))
        ref = []
        for mode, step, W, lm, thr, reads in plan:
            if mode == "global":
                ref.append(be.basecall_raw_global(reads, 4, chunk, step, W, lm, *thr))
            else:
                ref.append(be.basecall_raw_chunk(reads, 4, chunk, step, W))
        tickets = []
        for i, (mode, step, W, lm, thr, reads) in enumerate(plan):
            tickets.append(be.pipe_submit_raw(mode, reads, 4, chunk, step, W, lm, *thr))
            r = rng.random()
            if r < 0.2:
                be.pipe_progress(0)
            elif r < 0.3:
                tickets[int(rng.integers(0, len(tickets)))].wait()
            elif r < 0.4:      # a blocking call in between shares the context's workspaces
                j = int(rng.integers(0, len(plan)))
                m2, s2, W2, lm2, thr2, reads2 = plan[j]
                again = be.basecall_raw_global(reads2, 4, chunk, s2, W2, lm2, *thr2) if m2 == "global" else be.basecall_raw_chunk(reads2, 4, chunk, s2, W2)
                if not (np.array_equal(again[1], ref[j][1]) and same(m2, again[0], ref[j][0], ref[j][1])):
                    n_fail += 1
                    log(f"MISMATCH (interleaved blocking call) round={rd} batch={j}")
        if rng.random() < 0.5:
            be.pipe_flush()
        for i, t in enumerate(tickets):
            got, status = t.result()
            n_batches += 1
            if not (np.array_equal(status, ref[i][1]) and same(plan[i][0], got, ref[i][0], status)):
                n_fail += 1
                mode, step, W, lm, thr, reads = plan[i]
                log(f"MISMATCH round={rd} batch={i} mode={mode} chunk={chunk} step={step} W={W} lm={lm} thr={thr} lens={[len(r) for r in reads]}")
        if rd % 10 == 9:
            log(f"{rd + 1} rounds, {n_batches} batches, {n_fail} failures, {time.time() - t0:.0f}s")
    be.close()
    return n_batches, n_fail


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    n_batches, n_fail = run(*a, log=lambda m: print(m, flush=True))
    print(f"done: {n_batches} batches, {n_fail} failures")
    sys.exit(1 if n_fail else 0)
