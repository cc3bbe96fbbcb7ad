"""Reads-level fused paths (streamed forward: every time step computed once) must be BIT-IDENTICAL to the windowed
computation the reference performs (radian/basecall.py:83-121)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from radian_amd import Backend, weights
    b = Backend(0)
    b.load_weights(weights.synthetic_weights(seed=1234))
    yield b
    b.close()


def _reads(rng, lengths):
    return [np.clip(rng.normal(size=n), -4, 4).astype(np.float32) for n in lengths]


def _windows(sig, chunk, step):
    from radian_amd.preprocess import get_windows
    w, pad = get_windows(sig, chunk, step)
    valid = np.full(w.shape[0], chunk, dtype=np.int32)
    valid[-1] = chunk - pad
    return w.astype(np.float32), valid, pad


def test_stream_rows_equal_window_rows_bitwise(be):
    """rows >= 252 of a window == rows of one forward pass over the whole read (receptive field 253)."""
    rng = np.random.default_rng(0)
    sig = _reads(rng, [3000])[0]
    stream = be.forward(sig[None, :])[0]          # one segment of 3000 rows
    w, valid, pad = _windows(sig, 1024, 512)
    probs = be.forward(w)
    halo = 252
    for i in range(w.shape[0]):
        n = valid[i]
        assert np.array_equal(probs[i, halo:n], stream[i * 512 + halo: i * 512 + n]), i
    assert np.array_equal(probs[0, :1024], stream[:1024])
    # and rows < 252 of a later window do differ (they see the window's zero left-padding)
    assert not np.array_equal(probs[1, :halo], stream[512: 512 + halo])


@pytest.mark.parametrize("chunk,step", [(1024, 512), (1024, 128), (1024, 1024), (1024, 900), (300, 100), (256, 256), (128, 64), (200, 7)])
def test_reads_chunk_equals_windowed(be, chunk, step):
    rng = np.random.default_rng(chunk * 7 + step)
    lengths = [4096, 700, chunk, chunk + step, chunk + 3 * step + 17, 1, 2500, chunk - 1, chunk + 1]
    if step < 50:
        lengths = [1500, 700, chunk, chunk + step, 1, chunk + 1]   # keep the window count of tiny steps moderate
    sigs = _reads(rng, lengths)
    W = 10
    got = be.basecall_reads_chunk(sigs, chunk, step, W)
    assert len(got) == len(sigs)
    for r, sig in enumerate(sigs):
        w, valid, _ = _windows(sig, chunk, step)
        exp = be.basecall_chunk(w, valid, W)
        assert len(got[r]) == len(exp) == be.count_windows(len(sig), chunk, step)
        for i in range(len(exp)):
            assert np.array_equal(got[r][i], exp[i]), (r, i, lengths[r])


@pytest.mark.parametrize("chunk,step", [(1024, 512), (1024, 128), (1024, 772), (1024, 773), (1024, 1024), (300, 40), (128, 64), (4096, 2048)])
def test_reads_global_equals_windowed(be, chunk, step):
    rng = np.random.default_rng(chunk + step)
    k = 3
    be.load_lm(rng.dirichlet([0.3] * 4, size=4 ** k), k)
    lengths = [4096, 700, chunk, chunk + step, 3000, 5, chunk + 1]
    sigs = _reads(rng, lengths)
    got = be.basecall_reads_global(sigs, chunk, step, 6, True, 0.5, 0.5)
    wins, offs, pads = [], [0], []
    for sig in sigs:
        w, _, pad = _windows(sig, chunk, step)
        wins.append(w)
        offs.append(offs[-1] + w.shape[0])
        pads.append(pad)
    exp = be.basecall_global(np.concatenate(wins), np.array(offs, dtype=np.int32), np.array(pads, dtype=np.int32), step, 6, True, 0.5, 0.5)
    for r in range(len(sigs)):
        assert np.array_equal(got[r], exp[r]), (r, lengths[r])
    be.load_lm(None, 0)


def test_reads_global_vs_oracle_end_to_end(be, oracle):
    """reads-level global path == oracle assembly + decode of the windowed GPU probabilities (default geometry)."""
    rng = np.random.default_rng(5)
    k = 3
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    be.load_lm(table, k)
    sig = _reads(rng, [5000])[0]
    got = be.basecall_reads_global([sig], 1024, 128, 6, True, 0.5, 0.5)[0]
    w, valid, pad = _windows(sig, 1024, 128)
    probs = be.forward(w)
    mat = oracle.assemble_matrices(probs, pad, 128)
    exp, _ = oracle.beam_search_labels(mat, 6, table, 0.5, 0.5, k)
    assert np.array_equal(got, exp)
    be.load_lm(None, 0)


def test_pipe_submit_reads_equals_unpipelined(be):
    rng = np.random.default_rng(9)
    n_reads, N, chunk, step, W = 6, 4096, 1024, 512, 10
    batches = [np.stack(_reads(rng, [N] * n_reads)) for _ in range(5)]
    read_off = np.arange(n_reads + 1, dtype=np.int64) * N
    nwin = n_reads * be.count_windows(N, chunk, step)
    ref = [be.basecall_reads_chunk(list(b), chunk, step, W) for b in batches]
    dptr = []
    for b in batches:
        d = be.dev_alloc(b.nbytes)
        be.h2d(d, b)
        dptr.append(d)
    be.pipe_config(2)
    for lanes in (1, 2, 4):   # forwards of consecutive batches on 1 / 2 / 4 independent streams: same labels
        be.pipe_set_lanes(lanes)
        outs = [(np.zeros((nwin, chunk), dtype=np.uint8), np.full(nwin, -1, dtype=np.int32)) for _ in batches]
        for b in range(len(batches)):
            be.pipe_submit_reads(dptr[b], read_off, n_reads, chunk, step, W, outs[b][0], outs[b][1])
        be.pipe_flush()
        for b in range(len(batches)):
            lab, ln = outs[b]
            w = 0
            for r in range(n_reads):
                for frag in ref[b][r]:
                    assert ln[w] == len(frag) and np.array_equal(lab[w, : ln[w]], frag), (lanes, b, r, w)
                    w += 1
    for d in dptr:
        be.dev_free(d)
    be.pipe_config(4)
    be.pipe_set_lanes(2)


def test_normalise_on_device_matches_golden_and_numpy(be, golden_dir):
    """rd_normalise_reads == float32(mad_normalise(...)) bit for bit, incl. the int64 quirk and the two error statuses."""
    import json
    import os
    from radian_amd.preprocess import mad_normalise
    g = json.load(open(os.path.join(golden_dir, "preprocess_cases.json")))
    arr = np.load(os.path.join(golden_dir, "preprocess.npz"))
    raws, exp, exp_status, clips = [], [], [], []
    for c in g["cases"]:
        raws.append(arr["sig_" + c["name"]])
        clips.append(c["clip"])
        if "error" in c:
            exp.append(None)
            exp_status.append(1 if "MAD" in c["error"] else 2)
        else:
            exp.append(arr["norm_" + c["name"]].astype(np.float32))
            exp_status.append(0)
    for clip in sorted(set(clips)):
        idx = [i for i, c in enumerate(clips) if c == clip]
        got, status = be.normalise_reads([raws[i] for i in idx], clip)
        for j, i in enumerate(idx):
            assert status[j] == exp_status[i], g["cases"][i]["name"]
            if exp[i] is not None:
                assert np.array_equal(got[j], exp[i]), g["cases"][i]["name"]
    # random reads of many lengths / parities against the NumPy implementation
    rng = np.random.default_rng(3)
    reads = [np.round(rng.normal(500, 80, size=n)).astype(np.int16) for n in (1, 2, 3, 4, 5, 100, 101, 4096, 12833, 70001)]
    reads.append(np.array([-32768, 32767, 0, 5, -7, 32767, -32768, 9], dtype=np.int16))
    reads.append((rng.integers(-32768, 32768, size=5000)).astype(np.int16))
    got, status = be.normalise_reads(reads, 4)
    for r, sig in enumerate(reads):
        try:
            e = mad_normalise(sig, 4).astype(np.float32)
            assert status[r] == 0 and np.array_equal(got[r], e), (r, len(sig))
        except ValueError:
            assert status[r] == 1


def test_raw_paths_equal_normalised_paths(be):
    from radian_amd.preprocess import mad_normalise
    rng = np.random.default_rng(4)
    reads = [np.round(rng.normal(500, 80, size=n)).astype(np.int16) for n in (4096, 700, 3000)]
    reads.insert(1, np.full(300, 512, dtype=np.int16))   # MAD == 0
    frags, status = be.basecall_raw_chunk(reads, 4, 1024, 512, 10)
    assert status.tolist() == [0, 1, 0, 0]
    good = [i for i in range(4) if status[i] == 0]
    exp = be.basecall_reads_chunk([mad_normalise(reads[i], 4).astype(np.float32) for i in good], 1024, 512, 10)
    for j, i in enumerate(good):
        assert len(frags[i]) == len(exp[j]) and all(np.array_equal(a, b) for a, b in zip(frags[i], exp[j]))
    labs, status = be.basecall_raw_global(reads, 4, 1024, 128, 6, False)
    exp = be.basecall_reads_global([mad_normalise(reads[i], 4).astype(np.float32) for i in good], 1024, 128, 6, False)
    for j, i in enumerate(good):
        assert np.array_equal(labs[i], exp[j])


def test_long_read_global_streamed(be, oracle):
    """a 60k-sample read through the reads-level global path (469 windows at step 128 in the reference; one stream here)
    against the oracle's assembly + decode of the windowed GPU probabilities"""
    rng = np.random.default_rng(21)
    sig = _reads(rng, [60000])[0]
    got = be.basecall_reads_global([sig], 1024, 128, 6, False)[0]
    w, valid, pad = _windows(sig, 1024, 128)
    probs = be.forward(w)
    mat = oracle.assemble_matrices(probs, pad, 128)
    exp, _ = oracle.beam_search_labels(mat, 6)
    assert mat.shape[0] == 60000 and np.array_equal(got, exp)


def test_clone_artifacts_gives_an_identical_context():
    """rd_clone_artifacts = the receiver's half of the multi-GPU broadcast (header -> reserve / bind -> images -> loaded) with a
    device copy as the transport: a context that never parsed anything must compute exactly what the source computes, in every
    matrix-product mode, with the LM, and keep working after the source is gone"""
    from radian_amd import Backend, weights, synthetic
    rng = np.random.default_rng(4)
    table = rng.dirichlet([0.3] * 4, size=4 ** 4)
    src = Backend(0)
    src.load_weights(weights.synthetic_weights(seed=99, dilations=(1, 2, 4, 8)), (1, 2, 4, 8))
    src.load_lm(table, 4)
    dst = Backend(0)
    with pytest.raises(Exception):
        Backend(0).clone_artifacts_from(dst)          # nothing loaded in the source: an error, not a crash
    dst.clone_artifacts_from(src)
    reads = [r.astype(np.float32) for r in synthetic.synthetic_reads(6, 3000, seed=2)]
    sigs = [(r - r.mean()) / r.std() for r in reads]
    try:
        for prec in ("fp32", "f16x3", "bf16x3"):
            src.set_precision(prec)
            dst.set_precision(prec)
            a = src.basecall_reads_global(sigs, 1024, 256, 6, True, 0.3, 1.0)
            b = dst.basecall_reads_global(sigs, 1024, 256, 6, True, 0.3, 1.0)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), prec
        src.set_precision("fp32")
        c_src = src.basecall_reads_chunk(sigs, 1024, 256, 10)
        src.close()
        src = None
        dst.set_precision("fp32")
        c = dst.basecall_reads_chunk(sigs, 1024, 256, 10)
        assert len(c) == 6 and all(len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(c, c_src))
    finally:
        if src is not None:
            src.close()
        dst.close()


@pytest.mark.parametrize("chunk,step", [(1024, 512), (1024, 128), (1024, 1024), (1024, 772), (1024, 773), (300, 100), (128, 64)])
def test_forward_reads_equals_predict_on_windows_bitwise(be, oracle, chunk, step):
    """rd_forward_reads (round 4; BASELINE configs[1] "forward only"): sig_model.predict over get_windows() of whole reads
    (basecall.py:83-93) through the streamed evaluation and a gather into window shape -- bit-identical to rd_forward on the
    windows themselves for every row the pad trim (basecall.py:96) keeps, zero beyond; and within 1e-4 of the oracle."""
    rng = np.random.default_rng(chunk + 13 * step)
    lens = [3000, chunk, chunk - 1, chunk + 1, 1, chunk + step, 5 * step + 7, 2500]
    sigs = _reads(rng, lens)
    got = be.forward_reads(sigs, chunk, step)
    assert len(got) == len(sigs)
    for sig, g in zip(sigs, got):
        w, valid, pad = _windows(sig, chunk, step)
        ref = be.forward(w)
        assert g.shape == ref.shape
        for i in range(w.shape[0]):
            assert np.array_equal(g[i, : valid[i]], ref[i, : valid[i]]), (len(sig), i)
            assert not g[i, valid[i]:].any()
    # the oracle on the first read's windows (forward parity: |dp| <= 1e-4, the tolerance north_star states)
    from radian_amd import weights
    w, valid, _ = _windows(sigs[0], chunk, step)
    o = oracle.tcn_forward(weights.synthetic_weights(seed=1234), w)
    for i in range(w.shape[0]):
        assert float(np.abs(got[0][i, : valid[i]] - o[i, : valid[i]]).max(initial=0.0)) <= 1e-4
