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


def _bf16_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def test_bf16x3_split_reconstructs_fp32_exactly():
    """precision mode 2 carries every fp32 operand as hi + mid + lo bf16.  The device-side split (rd_split3: the function
    the kernels' epilogues call) must give hi + mid + lo == v EXACTLY, term by term in float32 arithmetic, for every sampled
    finite fp32: random bit patterns over the whole normal range down to 2^-100, activations-like and weight-like values,
    powers of two, values one ulp around them, all-ones mantissas, +-0.  (Below 2^-110 the third term would fall under
    bf16's smallest subnormal; such values are sampled separately and must be exact to 2^-133 absolute.)"""
    from radian_amd import Backend
    rng = np.random.default_rng(0)
    n = 1 << 20
    bits = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    e = (bits >> 23) & 0xFF
    bits = bits[(e >= 27) & (e <= 254)]                       # finite, |v| >= 2^-100
    vals = [bits.view(np.float32),
            rng.normal(size=200000).astype(np.float32), (rng.normal(size=200000) * 0.05).astype(np.float32),
            np.abs(rng.normal(size=100000)).astype(np.float32) * np.float32(37.0),
            np.array([0.0, -0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 255.0 / 256, 257.0 / 256, 1e-30, -1e30, 3.4028235e38,
                      2.0 ** -100, 1.9999999 * 2.0 ** -100], dtype=np.float32),
            (2.0 ** rng.integers(-100, 127, size=5000)).astype(np.float32)]
    p2 = vals[-1]
    vals += [np.nextafter(p2, np.float32(0)), np.nextafter(p2, np.float32(np.inf))]
    v = np.concatenate(vals)
    b = Backend(0)
    try:
        t = b.split3(v)
        hi, mid, lo = (_bf16_to_f32(t[i]) for i in range(3))
        rec = hi + (mid + lo)                                  # float32 adds (the epilogue's order); exact when the split is exact
        bad = np.nonzero(rec.view(np.uint32) != v.view(np.uint32))[0]
        bad = [i for i in bad if not (v[i] == 0 and rec[i] == 0)]       # -0.0 -> +0.0 is fine
        assert not bad, (len(bad), v[bad[:5]], rec[bad[:5]])
        assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), v.astype(np.float64))
        # ordering of magnitudes: each term refines the previous one
        nz = v != 0
        assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -7) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -14)
        # tiny values: absolute defect below bf16's subnormal spacing
        tiny = (rng.random(20000).astype(np.float32) + np.float32(1.0)) * (2.0 ** rng.integers(-126, -101, size=20000)).astype(np.float32)
        tt = b.split3(tiny)
        rec = sum(_bf16_to_f32(tt[i]).astype(np.float64) for i in range(3))
        assert np.abs(rec - tiny.astype(np.float64)).max() <= 2.0 ** -133
    finally:
        b.close()


def test_forward_bf16x3_accuracy(oracle):
    """precision mode 2 (six bf16 MFMAs per product on exactly split operands): inside the 1e-4 bound against the float32
    oracle, and against the float64-ACCUMULATED oracle of the same size as the exact-fp32-MFMA mode's error.  Both errors are
    float32 accumulation-order noise: over the 500 random shapes of tests/fuzz_forward.py (seed 7) bf16x3's worst case
    (8.1e-5) and mean (7.9e-6) are below the fp32 mode's (1.11e-4, 8.4e-6), but shape by shape either mode is the larger
    one about half the time (235 / 500) -- so "never above the fp32 mode's error on any shape" does not hold, and this
    mode is not the headline."""
    from radian_amd import Backend, weights
    rng = np.random.default_rng(10)
    shapes = [(1234, 1.0, (1, 2, 4, 8, 16, 32), 3, 1024), (77, 6.0, (1, 2, 4, 8, 16, 32), 3, 1024), (5, 3.0, (1, 2, 4), 3, 1024),
              (9, 1.0, (1, 2, 4, 8, 16, 32), 2, 700), (11, 3.0, (32, 16, 8, 4, 2, 1), 4, 333), (13, 1.0, (1,), 5, 129),
              (21, 3.0, (1, 2, 4, 8, 16, 32, 64), 2, 1500)]
    for seed, gain, dil, nW, T in shapes:
        w = weights.synthetic_weights(seed=seed, head_gain=gain, dilations=dil)
        b = Backend(0)
        try:
            b.load_weights(w, dil)
            x = np.clip(rng.normal(size=(nW, T)), -4, 4).astype(np.float32)
            exp = oracle.tcn_forward(w, x, dilations=dil, acc64=True)
            p32 = b.forward(x)
            b.set_precision("bf16x3")
            p3 = b.forward(x)
            assert np.all(np.isfinite(p3)) and np.allclose(p3.sum(axis=2), 1.0, atol=1e-5)
            e32, e3 = float(np.abs(p32 - exp).max()), float(np.abs(p3 - exp).max())
            print(f"seed {seed} gain {gain} dil {dil} {nW}x{T}: max|dp| vs f64-accumulated reference: fp32 MFMA {e32:.2e}, bf16x3 {e3:.2e}")
            assert e3 <= TOL and e32 <= TOL, (seed, e3, e32)
            assert e3 <= 3 * max(e32, 5e-6), (seed, e3, e32)     # same order as float32 summation-order noise
            if gain == 1.0:   # (a 3-6x head gain amplifies the float32 summation-order noise of BOTH sides of this comparison)
                assert np.abs(p3 - oracle.tcn_forward(w, x, dilations=dil)).max() <= TOL
            for Ts in (1, 100, 300):
                xs = rng.normal(size=(2, Ts)).astype(np.float32)
                assert np.abs(b.forward(xs) - oracle.tcn_forward(w, xs, dilations=dil)).max() <= TOL, Ts
            b.set_precision("fp32")
            assert np.array_equal(b.forward(x), p32)
        finally:
            b.close()


def test_bf16x3_streamed_equals_windowed_and_bench_labels(oracle):
    """mode 2 on the reads-level (streamed) path: rows >= 252 of a window are bitwise the read's stream rows, fragments equal
    the windowed evaluation's, and on a bench batch the labels are those of the oracle's beam search on the mode's own
    probabilities (decode parity is independent of the forward's arithmetic)."""
    from radian_amd import Backend, weights, synthetic
    from radian_amd.preprocess import get_windows
    b = Backend(0)
    try:
        b.load_weights(weights.synthetic_weights(seed=1234))
        b.set_precision("bf16x3")
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
        reads = synthetic.synthetic_reads(16, 4096, seed=0)
        win, valid, _, _ = synthetic.reads_to_windows(reads, 1024, 512)
        pr = b.forward(win)
        ref = oracle.tcn_forward(weights.synthetic_weights(seed=1234), win[:32])
        assert float(np.abs(pr[:32] - ref).max()) <= TOL
        frags = b.basecall_chunk(win, valid, 10)
        off = np.arange(win.shape[0], dtype=np.int64) * 1024
        lab = oracle.beam_search_batch(pr.reshape(-1, 5), off, valid, 10)
        assert all(np.array_equal(a, e) for a, e in zip(frags, lab))
    finally:
        b.close()


def test_configs1_batch256_forward_shape_vs_oracle(oracle):
    """BASELINE configs[1] as it is worded: batch = 256 windows (32 reads x 4096, chunk 1024 / step 512), forward only (beam 1 =
    greedy is decode-side, tests/test_gpu_baseline_configs.py): the whole batch's probabilities against the oracle forward,
    |d softmax| <= 1e-4, through the window-level entry point and -- bit-identical to it -- the streamed reads-level forward."""
    from radian_amd import Backend, synthetic, weights
    w = weights.synthetic_weights(seed=1234)
    raws = synthetic.synthetic_reads(32, 4096, seed=2)
    norm = [np.asarray(oracle.mad_normalise(r, 4), dtype=np.float32) for r in raws]
    win = np.concatenate([np.asarray(oracle.get_windows(n, 1024, 512)[0], dtype=np.float32) for n in norm])
    assert win.shape == (256, 1024)
    be = Backend(0)
    try:
        be.load_weights(w)
        probs = be.forward(win)
        ref = oracle.tcn_forward(w, win)
        err = float(np.abs(probs - ref).max())
        assert err <= 1e-4, err
        # greedy (beam 1) labels of every window: the streamed chunk path equals the oracle's decode of these probabilities
        got = be.basecall_reads_chunk(norm, 1024, 512, 1)
        valid = np.full(256, 1024, dtype=np.int32)
        valid[7::8] = 512
        exp = oracle.beam_search_batch(probs.reshape(-1, 5), np.arange(256, dtype=np.int64) * 1024, valid, 1)
        flat = [f for fr in got for f in fr]
        assert all(np.array_equal(a, b) for a, b in zip(flat, exp))
    finally:
        be.close()


def test_conv_workgroup_shapes_are_bit_identical():
    """rd_set_conv_shape 1 (256 x 256 tiles, one 512-thread workgroup per CU; a measurement variant, profiles/r03b_conv_shape.txt)
    runs the same MFMA sequence per output element as the product's 128 x 256 tiles: probabilities equal bit for bit, for uniform
    windows and for ragged reads through the streamed forward (packed head tiles, empty sub-tiles)."""
    from radian_amd import Backend, synthetic, weights
    be = Backend(0)
    try:
        be.load_weights(weights.synthetic_weights(seed=3))
        win = synthetic.reads_to_windows(synthetic.synthetic_reads(5, 3000, seed=4), 1024, 300)[0]
        ragged = [synthetic.synthetic_reads(1, n, seed=n)[0] for n in (1, 31, 33, 700, 1024, 1500, 4097, 9000)]
        res = []
        for shape in (0, 1):
            be.set_conv_shape(shape)
            res.append((be.forward(win), be.basecall_raw_chunk(ragged, 4, 1024, 512, 10)[0], be.basecall_raw_global(ragged, 4, 1024, 128, 6, False)[0]))
        assert np.array_equal(res[0][0], res[1][0])
        assert all(np.array_equal(a, b) for x, y in zip(res[0][1], res[1][1]) for a, b in zip(x, y))
        assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    finally:
        be.close()


def test_first_conv_fused_is_bit_identical(oracle):
    """Round 4: block 0's first conv (one input channel) is computed inside the kernel of the block's second conv -- its A tiles are
    built in LDS from the raw samples instead of being written to HBM by tcn_in_kernel and loaded back (tcn_gemm_kernel<..., FIN>).
    Same fma chain per value, same MFMA sequence per output: probabilities equal bit for bit with the first conv as its own
    kernel (rd_set_conv_fuse 0) -- uniform windows, whole ragged reads through the streamed forward with per-layer head tiles (reads
    shorter than a tile, lengths around the 32-row sub-tile and 128-row tile edges), dilations 1 and 2 in block 0; models whose
    block-0 dilation is beyond what the variant's LDS regions hold (4, 8, 16) take the two kernels either way; and the fused path is within
    1e-4 of the oracle."""
    from radian_amd import Backend, synthetic, weights
    be = Backend(0)
    try:
        for dil in ((1, 2, 4, 8, 16, 32), (2, 4, 1), (4, 1), (8, 2), (16, 1)):
            w = weights.synthetic_weights(seed=11 + dil[0], dilations=dil)
            be.load_weights(w, dil)
            win = synthetic.reads_to_windows(synthetic.synthetic_reads(5, 3000, seed=4), 1024, 300)[0]
            sigs = [np.clip(np.random.default_rng(n).normal(size=n), -4, 4).astype(np.float32)
                    for n in (1, 2, 3, 31, 32, 33, 127, 128, 129, 700, 1024, 1500, 4097, 6000)]
            res = []
            for fuse in (0, 1):
                be.set_conv_fuse(fuse)
                res.append((be.forward(win), be.forward_reads(sigs, 1024, 512), be.forward_reads(sigs, 300, 77)))
            assert np.array_equal(res[0][0], res[1][0]), dil
            for k in (1, 2):
                assert all(np.array_equal(a, b) for a, b in zip(res[0][k], res[1][k])), (dil, k)
            ref = oracle.tcn_forward(w, win[:3], dilations=dil)
            assert float(np.abs(res[1][0][:3] - ref).max()) <= 1e-4, dil
    finally:
        be.close()


def test_forward_against_stock_pytorch_operators():
    """The HIP forward against a plain PyTorch reference of the same op, directly (not through the oracle): `F.conv1d(dilation=d)` after a
    causal left pad, ReLU, block 0's 1x1 matching conv, Dense / ReLU / Dense / softmax in float64 on the CPU (tests/test_forward_torch_cpu.py
    holds the statement) -- 12 windows x 1024 samples of MAD-normalised synthetic signal, the bench's He-normal weights and a soft-head set
    (rows far from saturation, where a softmax difference cannot hide behind a saturated class), all three matrix-product modes.
    Tolerance: north_star's 1e-4 on the softmax; observed ~1e-5.  PyTorch is test plumbing here (CPU), not the product."""
    torch = pytest.importorskip("torch")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_forward_torch_cpu import _torch_forward
    from radian_amd import Backend, weights, synthetic
    from radian_amd.preprocess import mad_normalise
    reads = synthetic.synthetic_reads(3, 4096, seed=11)
    win = np.stack([mad_normalise(r, 4).astype(np.float32)[i * 1024:(i + 1) * 1024] for r in reads for i in range(4)])
    win[5, :300] = 0.0
    be = Backend(0)
    try:
        for tag, flat in (("he_normal", weights.synthetic_weights(seed=1234)), ("soft_head", weights.synthetic_weights(seed=77, head_gain=0.3))):
            ref = _torch_forward(flat, win, weights.DEFAULT_DILATIONS)
            be.load_weights(flat)
            for prec, tol in (("fp32", 1e-4), ("bf16x3", 1e-4), ("f16x3", 1e-4)):
                be.set_precision(prec)
                got = be.forward(win)
                err = float(np.abs(got - ref).max())
                assert got.shape == ref.shape and err <= tol, (tag, prec, err)
    finally:
        be.set_precision("fp32")
        be.close()
