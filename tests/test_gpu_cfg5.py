"""BASELINE configs[4]: beam width 25, --context-len 256, fp16 logits (no reference behaviour for a 256-label
context or for f16 rows -- these are modes of this library; parity is against the oracle's restatement of the same
definitions, bit-exact like everything else in the decoder).

* f16 logits (rd_set_logits): the head kernel rounds the softmax rows to float16, the decoder widens them exactly, so the
  labels must equal the oracle's beam search on float16(GPU float32 rows) for every window; agreement with the float32-row
  path and max |dp| are reported.
* long contexts (rd_load_lm_hashed): context of the last k <= 256 labels, table row = hash of the context, kept
  incrementally per beam with a 256-label ring; labels must equal the oracle's, which hashes the explicit labeling.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK, STEP, READ_LEN = 1024, 512, 4096


def _soft_weights():
    from radian_amd import weights
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)        # soft rows: long labelings, the gate fires (tests/test_gpu_baseline_configs.py)
    return w


def _mats(rng, n_seq, tmax, scale):
    lens = rng.integers(1, tmax + 1, size=n_seq)
    rows = []
    for n in lens:
        z = rng.normal(size=(n, 5)) * scale
        z -= z.max(axis=1, keepdims=True)
        e = np.exp(z)
        rows.append(e / e.sum(axis=1, keepdims=True))
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    return np.concatenate(rows), off, lens.astype(np.int32)


@pytest.mark.parametrize("k,order,W", [(256, 11, 25), (256, 5, 10), (17, 4, 6), (40, 3, 25), (3, 3, 10), (255, 2, 30)])
def test_long_context_hashed_lm_vs_oracle(oracle, k, order, W):
    """random sequences up to 700 rows (labelings grow past the 256-label window: the ring wraps), several widths"""
    from radian_amd import Backend
    rng = np.random.default_rng(k * 100 + W)
    table = rng.dirichlet([0.2] * 4, size=4 ** order)
    be = Backend(0)
    try:
        be.load_lm_hashed(table, order, k)
        for scale, tmax in ((0.4, 700), (1.0, 500)):
            mats, off, lens = _mats(rng, 120, tmax, scale)
            lens[0] = tmax
            mats, off, lens = _mats(np.random.default_rng(k + W + int(scale * 10)), 120, tmax, scale)
            for s_thr, r_thr in ((0.0, 5.0), (0.5, 0.5)):
                exp = oracle.beam_search_batch(mats, off, lens, W, table, s_thr, r_thr, k, hash_order=order)
                for form in (("waves", "lanes") if W > 12 else ("auto",)):     # both launch shapes of a wide beam (rd_set_decode_form)
                    be.set_decode_form(form)
                    got = be.decode_batch(mats, off, lens, W, use_lm=True, s_threshold=s_thr, r_threshold=r_thr)
                    bad = [i for i in range(len(lens)) if not np.array_equal(got[i], exp[i])]
                    assert not bad, (k, order, W, form, scale, s_thr, bad[:5], len(bad))
                if s_thr == 0.0:
                    nolm = oracle.beam_search_batch(mats, off, lens, W)
                    long_enough = [i for i in range(len(lens)) if len(exp[i]) > k + 5]
                    if k <= 40:
                        assert long_enough and any(not np.array_equal(exp[i], nolm[i]) for i in long_enough)
        if k == 256:   # the window really is exceeded
            assert max(len(e) for e in exp) > 256   # (270+ labels on the last set; 300+ on the flat one)
    finally:
        be.close()


def test_configs4_w25_ctx256_f16_logits_global_vs_oracle(oracle):
    """the stress config end to end on the bench reads: raw int16 -> global decode, step 512, W = 25, context 256 (hashed
    synthetic LM, table order 11), f16 logits; against oracle normalise -> windows -> f16(GPU rows) -> assemble -> decode"""
    from radian_amd import Backend, synthetic
    n_reads = 32
    raws = synthetic.synthetic_reads(n_reads, READ_LEN, seed=0)
    w = _soft_weights()
    order, k = 11, 256
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** order)
    be = Backend(0)
    try:
        be.load_weights(w)
        norm = np.stack([oracle.mad_normalise(r, 4) for r in raws]).astype(np.float32)
        wins, pads = [], []
        for r in range(n_reads):
            ww, pad = oracle.get_windows(norm[r], CHUNK, STEP)
            wins.append(ww)
            pads.append(pad)
        win = np.concatenate(wins).astype(np.float32)
        p32 = be.forward(win)                                      # float32 rows (window-level seam: always float32)
        p16 = p32.astype(np.float16).astype(np.float32)           # what the head kernel stores in f16-logits mode
        be.load_lm_hashed(table, order, k)
        be.set_logits("f16")
        got, status = be.basecall_raw_global(list(raws), 4, CHUNK, STEP, 25, True, 0.5, 0.5)
        assert not status.any()
        mats, lens = [], []
        for r in range(n_reads):
            m = oracle.assemble_matrices(p16[r * 8:(r + 1) * 8], int(pads[r]), STEP)
            mats.append(m)
            lens.append(m.shape[0])
        lens = np.asarray(lens, dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        exp = oracle.beam_search_batch(np.concatenate(mats), off, lens, 25, table, 0.5, 0.5, k, hash_order=order)
        bad = [r for r in range(n_reads) if not np.array_equal(got[r], exp[r])]
        assert not bad, (bad[:8], len(bad))
        assert np.mean([len(e) for e in exp]) > 256          # labelings longer than the context window
        nolm = oracle.beam_search_batch(np.concatenate(mats), off, lens, 25)
        assert any(not np.array_equal(a, c) for a, c in zip(exp, nolm))
    finally:
        be.close()


@pytest.mark.parametrize("W", [10, 25])
def test_f16_logits_chunk_vs_oracle_and_f32(oracle, W):
    """chunk mode on a bench batch with f16 rows: exact against the oracle on the rounded rows; >= 99.9 % of the windows
    carry the float32-row labels (bench weights: saturated rows)"""
    from radian_amd import Backend, weights, synthetic
    raws = synthetic.synthetic_reads(64, READ_LEN, seed=1)
    for name, w, min_agree in (("bench", weights.synthetic_weights(seed=1234), 0.999), ("soft", _soft_weights(), 0.0)):
        be = Backend(0)
        try:
            be.load_weights(w)
            norm = np.stack([oracle.mad_normalise(r, 4) for r in raws]).astype(np.float32)
            win = np.concatenate([oracle.get_windows(norm[r], CHUNK, STEP)[0] for r in range(64)]).astype(np.float32)
            valid = np.full(512, CHUNK, dtype=np.int32)
            valid[7::8] = CHUNK - STEP
            p32 = be.forward(win)
            p16 = p32.astype(np.float16)
            f32 = [f for fr in be.basecall_reads_chunk(list(norm), CHUNK, STEP, W) for f in fr]
            be.set_logits("f16")
            f16 = [f for fr in be.basecall_reads_chunk(list(norm), CHUNK, STEP, W) for f in fr]
            if W > 12:                      # the other launch shape of a wide beam reads the f16 rows the same way
                be.set_decode_form("lanes")
                f16_lanes = [f for fr in be.basecall_reads_chunk(list(norm), CHUNK, STEP, W) for f in fr]
                be.set_decode_form("auto")
                assert all(np.array_equal(a, b) for a, b in zip(f16, f16_lanes)), (name, W)
            be.set_logits("f32")
            off = np.arange(512, dtype=np.int64) * CHUNK
            exp = oracle.beam_search_batch(p16.astype(np.float32).reshape(-1, 5), off, valid, W)
            bad = [i for i in range(512) if not np.array_equal(f16[i], exp[i])]
            assert not bad, (name, W, bad[:8], len(bad))
            agree = np.mean([np.array_equal(a, b) for a, b in zip(f16, f32)])
            dp = float(np.abs(p16.astype(np.float32) - p32).max())
            print(f"{name} W={W}: f16-row labels == f32-row labels on {agree * 100:.2f} % of 512 windows, max |dp| = {dp:.2e}")
            assert agree >= min_agree, (name, agree)
            assert dp <= 2.0 ** -11 * 1.0001
            # the pipelined entry point takes the mode too
            be.set_logits("f16")
            d = be.dev_alloc(norm.nbytes)
            be.h2d(d, norm)
            lab = np.zeros((512, CHUNK), dtype=np.uint8)
            ln = np.full(512, -1, dtype=np.int32)
            be.pipe_submit_reads(d, np.arange(65, dtype=np.int64) * READ_LEN, 64, CHUNK, STEP, W, lab, ln)
            be.pipe_flush()
            be.dev_free(d)
            assert all(ln[i] == len(exp[i]) and np.array_equal(lab[i, : ln[i]], exp[i]) for i in range(512))
        finally:
            be.close()
