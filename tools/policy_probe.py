#!/usr/bin/env python3
"""The global-mode group policy (csrc/pipe_reads.hip: Calib, chain_rows) against measurements it does not make itself -- the evidence
behind tests/test_gpu_policy.py ("a policy test that is not a bench leg").

For a precision mode / beam width, optionally with a second PROCESS keeping the GPU busy in the background:
  independent  forward pace = HIP-event kernel timers (rd_timer_*: conv + head + first conv, one lane, blocking calls) over the rows
               evaluated; chain pace = the beam-search timer of a blocking decode of the same reads / the longest read's rows
  policy       what rd_pipe_policy_read reports after a stream of rd_pipe_submit_raw_global batches (ns per forward row, all lanes
               together; us per time step of a group's longest chain on the decode partition under the next group's forwards)
  throughput   samples/s of uniform streams (all reads 4096 samples; all reads 40960), of a stream whose batches ALTERNATE
               between the two (the longest read jumps 10x from one batch to the next), same samples per batch, and (round 6) of a stream
               of RAGGED batches (seeded log-normal read lengths 1.5 k ... 60 k: every batch another longest read, read count and plan)
usage: policy_probe.py [fp32|bf16x3|f16x3] [W=10] [load=0|1] [partition CUs per XCD = -1: by width]      ->  one JSON line
       policy_probe.py --load-worker SECONDS                      (internal: the background process)"""
import json, os, subprocess, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
CHUNK, STEP = 1024, 512


class LoadWorkerFailed(RuntimeError):
    pass


def reads_of(n, length, seed):
    rng = np.random.default_rng(seed)
    return [np.round(rng.normal(500, 80, size=length)).astype(np.int16) for _ in range(n)]


def ragged_batches(n_batches, samples_per_batch, seed, lo=1500, hi=60000, median=9000.0, sigma=0.8):
    """batches of seeded log-normal read lengths (dRNA reads are heavy-tailed), each filled up to samples_per_batch: every batch has another
    longest read, another read count and another plan -- the third stream shape of the policy test (round 6)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_batches):
        lens, total = [], 0
        while total < samples_per_batch:
            n = int(np.clip(np.exp(rng.normal(np.log(median), sigma)), lo, hi))
            n = min(n, max(lo, samples_per_batch - total))
            lens.append(n)
            total += n
        out.append([np.round(rng.normal(500, 80, size=n)).astype(np.int16) for n in lens])
    return out


def load_worker(seconds):
    """a second process on the same GPU: forwards back to back for `seconds`"""
    from radian_amd import Backend, weights
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1))
    raws = reads_of(64, 4096, 99)
    t0 = time.time()
    print("load-worker running", flush=True)
    while time.time() - t0 < seconds:
        be.basecall_raw_chunk(raws, 4, CHUNK, STEP, 1)
    be.close()


def independent(be, raws, W):
    """HIP-event timers around blocking calls -> (ns per forward row, us per time step of the longest chain)"""
    from radian_amd.backend import RD_TIMER_CONV, RD_TIMER_DECODE, RD_TIMER_HEAD, RD_TIMER_IN
    be.pipe_flush()
    be.basecall_raw_global(raws, 4, CHUNK, STEP, W, False)          # warm
    reps = 3
    for t in (RD_TIMER_CONV, RD_TIMER_HEAD, RD_TIMER_IN, RD_TIMER_DECODE):
        be.timer_enable(t, 64 * reps)
    for _ in range(reps):
        be.basecall_raw_global(raws, 4, CHUNK, STEP, W, False)
    be.sync()
    tc, th, ti, td = (be.timer_read(t) for t in (RD_TIMER_CONV, RD_TIMER_HEAD, RD_TIMER_IN, RD_TIMER_DECODE))
    for t in (RD_TIMER_CONV, RD_TIMER_HEAD, RD_TIMER_IN, RD_TIMER_DECODE):
        be.timer_enable(t, 0)
    rows = sum(len(r) for r in raws) * reps          # the streamed forward evaluates every sample once
    fwd_ms = tc["total_ms"] + th["total_ms"] + ti["total_ms"]
    longest = max(len(r) for r in raws)
    return {"ns_per_row": fwd_ms * 1e6 / rows, "us_per_step": td["total_ms"] * 1e3 / max(1, td["launches"]) / longest,
            "decode_launches": td["launches"]}


def stream(be, batches, W, n_submits, window_rows=32 << 20):
    """n_submits batches through the pipeline, cycling over `batches` -> samples/s (wall clock, all delivered).  The submitter runs ahead by
    up to window_rows rows: waiting for a ticket whose batch sits in the OPEN group closes that group at once (rd_pipe_progress), so a driver
    that waits too early never lets a group of long reads gather the forward rows that cover its chain (the CLI keeps 24 batches of up to
    4096 x chunk_len rows in flight; this probe's batches are 16x smaller, hence a window in rows)."""
    be.pipe_flush()
    t0 = time.perf_counter()
    tickets, samples, ahead = [], 0, 0
    for i in range(n_submits):
        b = batches[i % len(batches)]
        rows = sum(len(r) for r in b)
        tickets.append((be.pipe_submit_raw("global", b, 4, CHUNK, STEP, W, False), rows))
        samples += rows
        ahead += rows
        while ahead > window_rows:
            t, r = tickets.pop(0)
            t.result()
            ahead -= r
    for t, _ in tickets:
        t.result()
    be.pipe_flush()
    return samples / (time.perf_counter() - t0)


def independent_only(prec="fp32", W=10):
    """the independent figures alone (a fresh context, the long batch): what tests/test_gpu_policy.py measures again when a probe's figures
    -- not its throughput ratios -- missed"""
    from radian_amd import Backend, weights
    be = Backend(0)
    try:
        be.load_weights(weights.synthetic_weights(seed=1234))
        be.set_precision(prec)
        be.set_decode_math("glibc")
        return independent(be, reads_of(6, 40960, 2) + reads_of(4, 4096, 3), W)
    finally:
        be.close()


def probe(prec="fp32", W=10, load=False, quick=False, part=-1, ragged=True):
    from radian_amd import Backend, weights
    bg = None
    if load:
        bg = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load-worker", "240"], stdout=subprocess.PIPE, text=True)
        if "running" not in bg.stdout.readline():
            bg.kill()
            bg.wait()
            raise LoadWorkerFailed("the background load process did not come up")
    be = Backend(0)
    try:
        be.load_weights(weights.synthetic_weights(seed=1234))
        be.set_precision(prec)
        be.set_decode_math("glibc")
        be.set_decode_partition(part)            # (-1: the library's choice by beam width)
        short = reads_of(64, 4096, 1)            # 262 144 samples per batch either way
        long_ = reads_of(6, 40960, 2) + reads_of(4, 4096, 3)
        out = {"precision": prec, "W": W, "background_load": bool(load)}
        out["independent_short"] = independent(be, short, W)
        out["independent_long"] = independent(be, long_, W)
        n = 120 if quick else 360
        stream(be, [short, long_], W, 80)        # warm-up: clocks, the slots' buffers grown to their working size, first measurements in
        out["samples_per_s_short"] = stream(be, [short], W, n)
        pol_s = {m: be.pipe_policy(W, m) for m in (0, 1, 2, 3)}
        out["samples_per_s_long"] = stream(be, [long_], W, n)
        pol_l = {m: be.pipe_policy(W, m) for m in (0, 1, 2, 3)}
        out["samples_per_s_alternating"] = stream(be, [short, long_], W, n)
        pol_a = {m: be.pipe_policy(W, m) for m in (0, 1, 2, 3)}
        if ragged:
            rag = ragged_batches(12, 262144, 7)
            out["samples_per_s_ragged"] = stream(be, rag, W, n) * 1.0
            out["ragged_reads_per_batch"] = [len(b) for b in rag]
        out["policy_after_short"], out["policy_after_long"], out["policy_after_alternating"] = pol_s, pol_l, pol_a
        hs = 2.0 / (1.0 / out["samples_per_s_short"] + 1.0 / out["samples_per_s_long"])     # equal samples per batch: harmonic mean
        out["alternating_over_steady"] = out["samples_per_s_alternating"] / hs
        if ragged:
            out["ragged_over_steady"] = out["samples_per_s_ragged"] / hs
        return out
    finally:
        be.close()
        if bg is not None:
            bg.terminate()      # (the exact child this call started)
            bg.wait()


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--load-worker":
        load_worker(float(a[1]))
    elif a and a[0] == "--independent":        # --independent prec W: the independent figures alone, one JSON line
        print(json.dumps(independent_only(a[1], int(a[2]))))
    elif len(a) > 4:                           # prec W load partition ragged: every argument given (tests/test_gpu_policy.py); exit code 3 = no load worker
        try:
            print(json.dumps(probe(a[0], int(a[1]), bool(int(a[2])), part=int(a[3]), ragged=bool(int(a[4])))))
        except LoadWorkerFailed as e:
            print(str(e), file=sys.stderr)
            sys.exit(3)
    else:
        print(json.dumps(probe(a[0] if a else "fp32", int(a[1]) if len(a) > 1 else 10, bool(int(a[2])) if len(a) > 2 else False, part=int(a[3]) if len(a) > 3 else -1)))
