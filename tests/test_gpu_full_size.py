"""BASELINE.json configs[1..3] at their FULL size on one GPU: 10 000 reads x 4096 samples (int16 = round(N(500, 80)),
seed 0; chunk 1024 / step 512 -> 80 000 windows, 40.96 M samples), through the CLI's own entry point (the reads-level pipeline fed
with raw int16 reads: Backend.pipe_submit_raw -> rd_pipe_submit_raw_chunk / rd_pipe_submit_raw_global).

The oracle needs minutes for this set, so the whole set is pinned by properties that do not depend on size, and a seeded sample of
it by the oracle:
  * batching / order invariance (reads are independent: basecall.py:70-76): the set in input order in batches of 64 against the set
    REVERSED in batches of 100 -- every read's labels identical, and one checksum over the per-read checksums;
  * the set's labels are not degenerate (every read decoded, status 0, the expected total of windows);
  * 48 reads drawn from all over the set: oracle normalise (preprocess.py:24-49) -> windows (preprocess.py:4-21) -> the GPU's own
    probabilities of those windows -> oracle beam search (decode.py:100-212) [global: oracle assembly (matrix_assembly.py:6-53) and
    the 12-mer LM gate (decode.py:42-96)] == what the full-size run delivered for exactly those reads.
configs[1] = the forward + W = 1 case, configs[2] = chunk W = 10, configs[3]'s geometry = global, W = 10, 4^11 x 4 LM table,
thresholds 0.5 / 0.5 (per rank; the reads shard across ranks without a data-path collective).  Soft-head weights in the global case:
with the bench's own weights the rows are saturated and the gate never opens (tests/test_gpu_baseline_configs.py).
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK, STEP, READ_LEN, N_READS = 1024, 512, 4096, 10000
WIN_PER_READ = 8
N_SAMPLE = 48


@pytest.fixture(scope="module")
def raws():
    rng = np.random.default_rng(0)
    return np.round(rng.normal(500.0, 80.0, size=(N_READS, READ_LEN))).astype(np.int16)


def soft_weights():
    from radian_amd import weights
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)          # dense_1 kernel [128, 5]: soft rows, ~1000 bases per read, the LM gate fires
    return w


def run_set(be, raws, order, per_batch, mode, W, use_lm=False):
    """the whole set through the pipeline -> {read index: labels (global) | list of per-window labels (chunk)}"""
    out = {}
    tickets = []

    def collect(upto):
        while len(tickets) > upto:
            idx, t = tickets.pop(0)
            res, status = t.result()
            assert not status.any()
            for r, lab in zip(idx, res):
                out[int(r)] = lab

    for b0 in range(0, len(order), per_batch):
        idx = order[b0:b0 + per_batch]
        t = be.pipe_submit_raw(mode, [raws[r] for r in idx], 4, CHUNK, STEP, W, use_lm, 0.5, 0.5)
        tickets.append((idx, t))
        collect(24)                # the driver's habit: results are taken while later batches are in flight
    be.pipe_flush()
    collect(0)
    return out


def digest(res, mode):
    h = hashlib.sha1()
    for r in range(N_READS):
        hr = hashlib.sha1()
        if mode == "global":
            hr.update(np.asarray(res[r], dtype=np.uint8).tobytes())
        else:
            for w in res[r]:
                hr.update(len(w).to_bytes(4, "little"))
                hr.update(np.asarray(w, dtype=np.uint8).tobytes())
        h.update(hr.digest())
    return h.hexdigest()


def sample_reads():
    return np.sort(np.random.default_rng(7).choice(N_READS, size=N_SAMPLE, replace=False))


@pytest.mark.parametrize("W", [1, 10])
def test_configs12_chunk_full_set(oracle, raws, W):
    from radian_amd import Backend, weights
    be = Backend(0)
    try:
        be.load_weights(weights.synthetic_weights(seed=1234))
        a = run_set(be, raws, np.arange(N_READS), 64, "chunk", W)
        b = run_set(be, raws, np.arange(N_READS)[::-1], 100, "chunk", W)
        assert len(a) == len(b) == N_READS
        assert sum(len(v) for v in a.values()) == N_READS * WIN_PER_READ
        bad = [r for r in range(N_READS) if len(a[r]) != len(b[r]) or any(not np.array_equal(x, y) for x, y in zip(a[r], b[r]))]
        assert not bad, (W, bad[:8], len(bad))
        assert digest(a, "chunk") == digest(b, "chunk")
        assert sum(len(w) for v in a.values() for w in v) > N_READS       # saturated rows: a few bases per window, not none
        # the sample against the oracle
        pick = sample_reads()
        norm = [oracle.mad_normalise(raws[r], 4).astype(np.float32) for r in pick]
        probs = be.forward_reads(norm, CHUNK, STEP)
        valid = np.full(WIN_PER_READ, CHUNK, dtype=np.int32)
        valid[-1] = CHUNK - STEP
        off = np.arange(WIN_PER_READ, dtype=np.int64) * CHUNK
        for r, p in zip(pick, probs):
            assert p.shape == (WIN_PER_READ, CHUNK, 5)
            exp = oracle.beam_search_batch(p.reshape(-1, 5), off, valid, W)
            got = a[int(r)]
            assert len(got) == WIN_PER_READ and all(np.array_equal(g, e) for g, e in zip(got, exp)), (W, int(r))
    finally:
        be.close()


def test_configs3_global_lm_full_set(oracle, raws):
    from radian_amd import Backend
    k = 11
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** k)
    be = Backend(0)
    try:
        be.load_weights(soft_weights())
        be.load_lm(table, k)
        a = run_set(be, raws, np.arange(N_READS), 64, "global", 10, True)
        b = run_set(be, raws, np.arange(N_READS)[::-1], 100, "global", 10, True)
        bad = [r for r in range(N_READS) if not np.array_equal(a[r], b[r])]
        assert not bad, (bad[:8], len(bad))
        assert digest(a, "global") == digest(b, "global")
        assert np.mean([len(a[r]) for r in range(N_READS)]) > 300
        pick = sample_reads()
        norm = [oracle.mad_normalise(raws[r], 4).astype(np.float32) for r in pick]
        probs = be.forward_reads(norm, CHUNK, STEP)
        mats = [oracle.assemble_matrices(p, STEP, STEP) for p in probs]
        lens = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
        assert (lens == READ_LEN).all() and all(m.dtype == np.float64 for m in mats)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        exp = oracle.beam_search_batch(np.concatenate(mats), off, lens, 10, table, 0.5, 0.5, k)
        nolm = oracle.beam_search_batch(np.concatenate(mats), off, lens, 10)
        assert all(np.array_equal(a[int(r)], e) for r, e in zip(pick, exp)), [int(r) for r, e in zip(pick, exp) if not np.array_equal(a[int(r)], e)]
        assert sum(not np.array_equal(x, y) for x, y in zip(exp, nolm)) >= N_SAMPLE // 2      # the gate fired on this workload
    finally:
        be.load_lm(None, 0)
        be.close()


def test_configs4_w25_ctx256_f16_logits_full_set(oracle, raws):
    """configs[4] on the whole set: global decode, W = 25, a context of the last 256 labels (hashed synthetic LM, table order 11), f16
    logits -- modes of this library without reference behaviour, pinned by the oracle's restatement of the same
    definitions (tests/test_gpu_cfg5.py): invariance over the whole set, the sample against oracle decode of float16(GPU rows)."""
    from radian_amd import Backend
    order, k = 11, 256
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** order)
    be = Backend(0)
    try:
        be.load_weights(soft_weights())
        be.load_lm_hashed(table, order, k)
        pick = sample_reads()
        norm = [oracle.mad_normalise(raws[r], 4).astype(np.float32) for r in pick]
        probs = be.forward_reads(norm, CHUNK, STEP)              # float32 rows; the f16 mode stores float16(these)
        be.set_logits("f16")
        a = run_set(be, raws, np.arange(N_READS), 64, "global", 25, True)
        b = run_set(be, raws, np.arange(N_READS)[::-1], 100, "global", 25, True)
        bad = [r for r in range(N_READS) if not np.array_equal(a[r], b[r])]
        assert not bad, (bad[:8], len(bad))
        assert digest(a, "global") == digest(b, "global")
        mats = [oracle.assemble_matrices(p.astype(np.float16).astype(np.float32), STEP, STEP) for p in probs]
        lens = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        exp = oracle.beam_search_batch(np.concatenate(mats), off, lens, 25, table, 0.5, 0.5, k, hash_order=order)
        assert all(np.array_equal(a[int(r)], e) for r, e in zip(pick, exp)), [int(r) for r, e in zip(pick, exp) if not np.array_equal(a[int(r)], e)]
        assert np.mean([len(e) for e in exp]) > 256               # labelings longer than the context window
    finally:
        be.set_logits("f32")
        be.close()
