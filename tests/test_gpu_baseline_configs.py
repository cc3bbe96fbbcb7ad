"""Oracle parity at BASELINE.json's full sizes, on the exact batches bench.py times.

configs[2] (the headline): rank 0's four bench batches (seeds 1000*rank + b, 64 reads x 4096 samples, chunk 1024 / step
512 -> 512 windows per batch) go through the same entry point and pipeline configuration as the timed region
(rd_pipe_submit_reads, 2 forward lanes, decode groups of 8 batches); EVERY window's labels are compared with the oracle's
beam search (decode.py:100-212) of the GPU's own probabilities, for W in {1, 10, 25}, and a whole batch's probabilities
with the oracle forward (model.py:52-89) within 1e-4.
configs[3] geometry on one GPU: the same 64 reads, raw int16 in, --decode-type global, step 512, W = 10, a 12-mer LM
(k = 11 Dirichlet(0.3) table, thresholds 0.5 / 0.5) through rd_basecall_raw_global against oracle normalise -> window ->
assemble (matrix_assembly.py:6-53) -> LM beam search.
configs[1] (forward only, beam 1) is the probability comparison plus the W = 1 case.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK, STEP, READ_LEN, N_READS, N_BATCHES = 1024, 512, 4096, 64, 4
WIN_PER_READ = 8


@pytest.fixture(scope="module")
def be():
    from radian_amd import Backend, weights
    b = Backend(0)
    b.load_weights(weights.synthetic_weights(seed=1234))
    yield b
    b.close()


@pytest.fixture(scope="module")
def batches(oracle):
    """rank 0's bench batches, preprocessed by the ORACLE (preprocess.py:4-49), not by the product's host mirror"""
    from radian_amd import synthetic
    out = []
    for b in range(N_BATCHES):
        raws = synthetic.synthetic_reads(N_READS, READ_LEN, seed=1000 * 0 + b)
        norm = np.stack([oracle.mad_normalise(r, 4) for r in raws]).astype(np.float32)
        wins, pads = [], []
        for r in range(N_READS):
            w, pad = oracle.get_windows(norm[r], CHUNK, STEP)
            assert w.shape == (WIN_PER_READ, CHUNK) and pad == STEP
            wins.append(w)
            pads.append(pad)
        valid = np.full(N_READS * WIN_PER_READ, CHUNK, dtype=np.int32)
        valid[WIN_PER_READ - 1::WIN_PER_READ] = CHUNK - STEP
        out.append({"raw": raws, "norm": norm, "win": np.concatenate(wins).astype(np.float32), "valid": valid,
                    "pads": np.asarray(pads, dtype=np.int32)})
    return out


@pytest.fixture(scope="module")
def gpu_probs(be, batches):
    """the GPU's probabilities of every window of every batch (window-level forward; the streamed forward of the timed
    path is bit-identical to it: tests/test_gpu_reads.py)"""
    return [be.forward(b["win"]) for b in batches]


def test_configs1_forward_probabilities_vs_oracle(be, oracle, batches, gpu_probs):
    """a whole bench batch (512 windows x 1024 rows) against the oracle forward: |d softmax| <= 1e-4 (north_star's bound)"""
    from radian_amd import weights
    ref = oracle.tcn_forward(weights.synthetic_weights(seed=1234), batches[0]["win"])
    err = float(np.abs(gpu_probs[0] - ref).max())
    assert err <= 1e-4, err
    assert np.abs(gpu_probs[0].sum(axis=2) - 1.0).max() < 1e-5


@pytest.mark.parametrize("W", [1, 10, 25])
def test_configs2_chunk_bench_batches_vs_oracle(be, oracle, batches, gpu_probs, W):
    """the timed region's path and pipeline configuration; every window of every batch label for label"""
    nwin = N_READS * WIN_PER_READ
    read_off = np.arange(N_READS + 1, dtype=np.int64) * READ_LEN
    dptr = []
    for b in batches:
        d = be.dev_alloc(b["norm"].nbytes)
        be.h2d(d, b["norm"])
        dptr.append(d)
    be.pipe_flush()
    be.pipe_config(8)
    be.pipe_set_lanes(2)
    n_sub = 10   # more than one decode group: batches cycle like bench.py's steps
    outs = [(np.zeros((nwin, CHUNK), dtype=np.uint8), np.full(nwin, -1, dtype=np.int32)) for _ in range(n_sub)]
    try:
        for i in range(n_sub):
            be.pipe_submit_reads(dptr[i % N_BATCHES], read_off, N_READS, CHUNK, STEP, W, outs[i][0], outs[i][1])
        be.pipe_flush()
    finally:
        for d in dptr:
            be.dev_free(d)
    off = np.arange(nwin, dtype=np.int64) * CHUNK
    for b in range(N_BATCHES):
        exp = oracle.beam_search_batch(gpu_probs[b].reshape(-1, 5), off, batches[b]["valid"], W)
        for i in range(b, n_sub, N_BATCHES):
            lab, ln = outs[i]
            bad = [w for w in range(nwin) if ln[w] != len(exp[w]) or not np.array_equal(lab[w, : ln[w]], exp[w])]
            assert not bad, (W, b, i, bad[:8], len(bad))
    assert sum(len(e) for e in exp) > 0


@pytest.mark.parametrize("head_scale", [1.0, 0.05])
def test_configs3_global_lm_k11_vs_oracle(be, oracle, batches, gpu_probs, head_scale):
    """64 raw reads, global decode, step 512, W = 10, 12-mer LM (4^11 x 4 table), thresholds 0.5 / 0.5.
    head_scale 1.0 = the bench's weights: their softmax rows are saturated (random He-normal weights through 12 layers),
    so the signal-entropy side of the gate (decode.py:91) never opens and labelings are a few bases long.  head_scale 0.05
    multiplies the last Dense kernel: soft rows (mean entropy 0.85 nat), labelings of ~1000 bases, the gate fires and
    changes the result -- the LM path proper at this geometry."""
    from radian_amd import Backend, weights
    k = 11
    rng = np.random.default_rng(0)
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    b = batches[0]
    own = None
    if head_scale == 1.0:
        dev, probs = be, gpu_probs[0]
    else:
        w = weights.synthetic_weights(seed=1234).copy()
        w[-645:-5] *= np.float32(head_scale)          # dense_1 kernel [128,5]; the 5 biases stay
        dev = own = Backend(0)
        dev.load_weights(w)
        probs = dev.forward(b["win"])
        ref = oracle.tcn_forward(w, b["win"][:64])
        assert float(np.abs(probs[:64] - ref).max()) <= 1e-4
    dev.load_lm(table, k)
    try:
        got, status = dev.basecall_raw_global(list(b["raw"]), 4, CHUNK, STEP, 10, True, 0.5, 0.5)
        assert not status.any()
        mats, lens = [], []
        for r in range(N_READS):
            m = oracle.assemble_matrices(probs[r * WIN_PER_READ:(r + 1) * WIN_PER_READ], int(b["pads"][r]), STEP)
            assert m.dtype == np.float64 and m.shape == (READ_LEN, 5)
            mats.append(m)
            lens.append(m.shape[0])
        lens = np.asarray(lens, dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        exp = oracle.beam_search_batch(np.concatenate(mats), off, lens, 10, table, 0.5, 0.5, k)
        bad = [r for r in range(N_READS) if not np.array_equal(got[r], exp[r])]
        assert not bad, (head_scale, bad[:8], len(bad))
        if head_scale != 1.0:
            # the gate must actually fire on this workload, or the LM path was not exercised
            nolm = oracle.beam_search_batch(np.concatenate(mats), off, lens, 10)
            assert sum(not np.array_equal(a, c) for a, c in zip(exp, nolm)) >= N_READS // 2
            assert np.mean([len(e) for e in exp]) > 300
    finally:
        dev.load_lm(None, 0)
        if own is not None:
            own.close()


@pytest.mark.parametrize("W", [1, 10, 25])
def test_configs2_chunk_soft_head_vs_oracle(oracle, batches, W):
    """the headline geometry with the softened head (see above): long fragments, many live beams, merges and re-entries
    -- the decoder's hard case -- through the reads-level chunk path, every window against the oracle"""
    from radian_amd import Backend, weights
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)
    dev = Backend(0)
    try:
        dev.load_weights(w)
        b = batches[1]
        probs = dev.forward(b["win"])
        got = dev.basecall_reads_chunk(list(b["norm"]), CHUNK, STEP, W)
        nwin = N_READS * WIN_PER_READ
        off = np.arange(nwin, dtype=np.int64) * CHUNK
        exp = oracle.beam_search_batch(probs.reshape(-1, 5), off, b["valid"], W)
        flat = [f for frs in got for f in frs]
        bad = [i for i in range(nwin) if not np.array_equal(flat[i], exp[i])]
        assert not bad, (W, bad[:8], len(bad))
        if W >= 10:
            assert np.mean([len(e) for e in exp]) > 100
    finally:
        dev.close()
