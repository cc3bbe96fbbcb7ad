"""HIP TCN forward and assembly (through the C ABI) against the CPU oracle.

Forward tolerance: |softmax difference| <= 1e-4 (north_star); observed ~1e-6 (fp32 MFMA vs fp32 CPU, different
summation order).  Assembly: bit-exact (float64 gather + one division)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def be():
    from radian_amd import Backend, weights
    b = Backend(0)
    b.load_weights(weights.synthetic_weights(seed=1234))
    yield b
    b.close()


def test_forward_vs_oracle(be, oracle):
    from radian_amd import weights
    w = weights.synthetic_weights(seed=1234)
    rng = np.random.default_rng(0)
    x = np.clip(rng.normal(size=(3, 1024)), -4, 4).astype(np.float32)
    x[2, 700:] = 0.0  # zero-padded tail window
    got = be.forward(x)
    exp = oracle.tcn_forward(w, x)
    assert got.shape == (3, 1024, 5)
    assert np.all(np.isfinite(got))
    assert np.abs(got.sum(axis=2) - 1.0).max() < 1e-5
    err = np.abs(got - exp).max()
    assert err <= TOL, err


def test_forward_odd_chunk_len_and_weights(be, oracle):
    from radian_amd import Backend, weights
    w = weights.synthetic_weights(seed=77, head_gain=6.0, dilations=(1, 2, 4))
    b2 = Backend(0)
    b2.load_weights(w, dilations=(1, 2, 4))
    rng = np.random.default_rng(1)
    for T in (1, 100, 129, 300):
        x = rng.normal(size=(2, T)).astype(np.float32)
        got = b2.forward(x)
        exp = oracle.tcn_forward(w, x, dilations=(1, 2, 4))
        assert np.abs(got - exp).max() <= TOL, T
    b2.close()


def test_forward_batch_independence(be):
    """Results are batch-independent: a window alone == the same window inside a batch of 40."""
    rng = np.random.default_rng(2)
    x = rng.normal(size=(40, 1024)).astype(np.float32)
    full = be.forward(x)
    one = be.forward(x[17:18])
    assert np.array_equal(full[17], one[0])


def test_forward_causality(be):
    """Row t depends only on samples <= t (causal padding) and on at most 252 samples back (receptive field 253)."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 1024)).astype(np.float32)
    y = x.copy()
    y[0, 600:] += 1.0
    a, b = be.forward(x), be.forward(y)
    assert np.array_equal(a[0, :600], b[0, :600])
    z = x.copy()
    z[0, :300] = 0.0
    c = be.forward(z)
    assert np.array_equal(a[0, 300 + 252:], c[0, 300 + 252:])


def test_assemble_golden_and_oracle(be, oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "assemble_cases.json")))
    arr = np.load(os.path.join(golden_dir, "assemble.npz"))
    for c in g["cases"]:
        probs = arr["probs_" + c["tag"]]
        got = be.assemble(probs, c["pad"], c["step"])
        exp = arr["out_" + c["tag"]]
        assert str(got.dtype) == c["out_dtype"], c["tag"]
        assert np.array_equal(got, exp), c["tag"]
    # reference default geometry at full size: chunk 1024, step 128, N = 12833 (data/reads.fast5 read 0)
    rng = np.random.default_rng(5)
    N, chunk, step = 12833, 1024, 128
    nW = (N - chunk) // step + 2
    pad = (nW - 1) * step + chunk - N
    probs = rng.random(size=(nW, chunk, 5), dtype=np.float32) + np.float32(0.01)
    probs /= probs.sum(axis=2, keepdims=True)
    got = be.assemble(probs, pad, step)
    exp = oracle.assemble_matrices(probs, pad, step)
    assert got.shape == (N, 5) and np.array_equal(got, exp)


def test_fused_chunk_equals_separate(be, oracle):
    """rd_basecall_chunk == rd_forward then beam search of each window (oracle decoder on the GPU's probabilities)."""
    rng = np.random.default_rng(6)
    x = rng.normal(size=(6, 1024)).astype(np.float32)
    valid = np.array([1024, 1024, 512, 1024, 1, 1000], dtype=np.int32)
    probs = be.forward(x)
    for W in (1, 10):
        fused = be.basecall_chunk(x, valid, W)
        for i in range(6):
            exp, _ = oracle.beam_search_labels(probs[i, : valid[i]], W)
            assert np.array_equal(fused[i], exp), (W, i)


def test_fused_global_equals_separate(be, oracle):
    rng = np.random.default_rng(7)
    k = 3
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    be.load_lm(table, k)
    chunk, step = 1024, 512
    # read 0: 4096 samples -> 8 windows, pad 512; read 1: 700 samples -> 1 window, pad 324 (float32 matrix)
    read_win_off = np.array([0, 8, 9], dtype=np.int32)
    pads = np.array([512, 324], dtype=np.int32)
    x = rng.normal(size=(9, chunk)).astype(np.float32)
    x[7, 512:] = 0
    x[8, 700:] = 0
    probs = be.forward(x)
    got = be.basecall_global(x, read_win_off, pads, step, 6, True, 0.5, 0.5)
    m0 = oracle.assemble_matrices(probs[0:8], 512, step)
    m1 = oracle.assemble_matrices(probs[8:9], 324, step)
    assert m0.dtype == np.float64 and m1.dtype == np.float32
    e0, _ = oracle.beam_search_labels(m0, 6, table, 0.5, 0.5, k)
    e1, _ = oracle.beam_search_labels(m1, 6, table, 0.5, 0.5, k)
    assert np.array_equal(got[0], e0)
    assert np.array_equal(got[1], e1)
    be.load_lm(None, 0)


def test_pipeline_equals_unpipelined(be):
    """rd_pipe_submit/flush (two streams, batch i+1 forward overlapping batch i decode) == rd_basecall_chunk."""
    rng = np.random.default_rng(8)
    n, T = 24, 1024
    batches = []
    for b in range(5):
        x = rng.normal(size=(n, T)).astype(np.float32)
        valid = np.full(n, T, dtype=np.int32)
        valid[7::8] = 512
        valid[3] = 0
        batches.append((x, valid))
    ref = [be.basecall_chunk(x, v, 10) for x, v in batches]
    dptr = []
    for x, _ in batches:
        d = be.dev_alloc(x.nbytes)
        be.h2d(d, x)
        dptr.append(d)
    outs = [(np.zeros((n, T), dtype=np.uint8), np.full(n, -1, dtype=np.int32)) for _ in batches]
    for b, (x, v) in enumerate(batches):
        be.pipe_submit(dptr[b], n, T, v, 10, outs[b][0], outs[b][1])
    be.pipe_flush()
    for b in range(len(batches)):
        lab, ln = outs[b]
        for i in range(n):
            assert ln[i] == len(ref[b][i]), (b, i)
            assert np.array_equal(lab[i, : ln[i]], ref[b][i]), (b, i)
    for d in dptr:
        be.dev_free(d)


def test_forward_requires_weights():
    from radian_amd import Backend, RadianHipError
    b = Backend(0)
    with pytest.raises(RadianHipError):
        b.forward(np.zeros((1, 64), dtype=np.float32))
    b.close()


def test_forward_split_f16x3_accuracy(oracle):
    """precision mode 1 (split-f16 products hi*hi + hi*lo + lo*hi, fp32 accumulate): softmax within 1e-4 of the oracle,
    i.e. the same bound as the exact-fp32 mode, also with a peaky head and odd segment lengths."""
    from radian_amd import Backend, weights
    rng = np.random.default_rng(10)
    for seed, gain, dil in ((1234, 1.0, (1, 2, 4, 8, 16, 32)), (77, 6.0, (1, 2, 4, 8, 16, 32)), (5, 3.0, (1, 2, 4))):
        w = weights.synthetic_weights(seed=seed, head_gain=gain, dilations=dil)
        b = Backend(0)
        b.load_weights(w, dil)
        x = np.clip(rng.normal(size=(3, 1024)), -4, 4).astype(np.float32)
        exp = oracle.tcn_forward(w, x, dilations=dil, acc64=True)   # float64-accumulated yardstick
        p32 = b.forward(x)
        b.set_precision("f16x3")
        p16 = b.forward(x)
        assert np.all(np.isfinite(p16))
        e32, e16 = np.abs(p32 - exp).max(), np.abs(p16 - exp).max()
        print(f"seed {seed} gain {gain}: max|dp| vs f64-accumulated reference: fp32 MFMA {e32:.2e}, f16x3 {e16:.2e}")
        assert e16 <= TOL, (seed, e16)
        assert e32 <= TOL, (seed, e32)
        assert e16 <= 3 * max(e32, 5e-6), (seed, e16, e32)     # no worse than float32 summation-order noise
        for T in (1, 100, 300):
            xs = rng.normal(size=(2, T)).astype(np.float32)
            assert np.abs(b.forward(xs) - oracle.tcn_forward(w, xs, dilations=dil)).max() <= TOL, T
        b.set_precision("fp32")
        assert np.array_equal(b.forward(x), p32)
        b.close()


def test_split_mode_streamed_equals_windowed():
    from radian_amd import Backend, weights
    from radian_amd.preprocess import get_windows
    b = Backend(0)
    b.load_weights(weights.synthetic_weights(seed=1234))
    b.set_precision("f16x3")
    rng = np.random.default_rng(11)
    sig = np.clip(rng.normal(size=3000), -4, 4).astype(np.float32)
    stream = b.forward(sig[None, :])[0]
    w, pad = get_windows(sig, 1024, 512)
    probs = b.forward(w.astype(np.float32))
    for i in range(w.shape[0]):
        n = 1024 if i < w.shape[0] - 1 else 1024 - pad
        assert np.array_equal(probs[i, 252:n], stream[i * 512 + 252: i * 512 + n]), i
    got = b.basecall_reads_chunk([sig], 1024, 512, 10)[0]
    valid = np.full(w.shape[0], 1024, dtype=np.int32)
    valid[-1] = 1024 - pad
    exp = b.basecall_chunk(w.astype(np.float32), valid, 10)
    assert all(np.array_equal(a, e) for a, e in zip(got, exp))
    b.close()
