"""HIP beam search (through the C ABI) against the golden vectors of the reference and against the CPU oracle.

Bar: emitted labelings bit-exact; winner scores within a few ulp (ROCm's log/log1p/exp vs glibc)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BASES = "ACGT"


def fdec(h):
    return float.fromhex(h)


@pytest.fixture(scope="module")
def be():
    from radian_amd import Backend
    b = Backend(0)
    yield b
    b.close()


def s_of(labels):
    return "".join(BASES[c] for c in labels)


def softmax_rows(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def test_golden_nolm(be, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "beam_nolm_cases.json")))
    mats = np.load(os.path.join(golden_dir, "beam_nolm_mats.npz"))
    bad = []
    for c in g["cases"]:
        mat = mats[c["mat"]]
        (lab,), sc = be.decode_batch(mat.reshape(-1, 5), [0], [mat.shape[0]], c["W"], with_scores=True)
        if s_of(lab) != c["seq"]:
            bad.append((c["mat"], c["T"], c["kind"], c["W"]))
        elif "final" in c:
            exp = fdec(c["final"][0]["pr_total"])
            if np.isfinite(exp):
                assert abs(sc[0] - exp) <= 1e-12 * max(1.0, abs(exp)), (c["mat"], c["W"], sc[0], exp)
            else:
                assert sc[0] == exp
    assert not bad, bad


def test_golden_scores_bit_exact_in_glibc_mode(be, golden_dir):
    """rd_set_decode_math(1): log / exp / log1p evaluated as glibc 2.35's x86-64 FMA build does (csrc/glibc_math.h) -- the libm
    the golden vectors were generated on.  The winner's pr_total then equals the REFERENCE's own float64 bit for bit, on every
    no-LM golden case (136 matrices incl. exact 0 / 1 probabilities and duplicated rows), and every labeling is the reference's."""
    g = json.load(open(os.path.join(golden_dir, "beam_nolm_cases.json")))
    mats = np.load(os.path.join(golden_dir, "beam_nolm_mats.npz"))
    be.set_decode_math("glibc")
    try:
        n_scores = 0
        for c in g["cases"]:
            mat = mats[c["mat"]]
            (lab,), sc = be.decode_batch(mat.reshape(-1, 5), [0], [mat.shape[0]], c["W"], with_scores=True)
            assert s_of(lab) == c["seq"], (c["mat"], c["W"])
            if "final" in c:
                exp = fdec(c["final"][0]["pr_total"])
                assert sc[0] == exp or (np.isnan(exp) and np.isnan(sc[0])), (c["mat"], c["W"], float(sc[0]).hex(), c["final"][0]["pr_total"])
                n_scores += 1
        assert n_scores >= 50
        gl = json.load(open(os.path.join(golden_dir, "beam_lm_cases.json")))
        ml = np.load(os.path.join(golden_dir, "beam_lm_mats.npz"))
        cur = None
        for c in gl["cases"]:
            if cur != c["lm"]:
                be.load_lm(ml[c["lm"]], c["k"])
                cur = c["lm"]
            mat = ml[c["mat"]]
            (lab,), sc = be.decode_batch(mat.reshape(-1, 5), [0], [mat.shape[0]], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]),
                                         r_threshold=fdec(c["r_thr"]), with_scores=True)
            assert s_of(lab) == c["seq"], (c["mat"], c["k"], c["W"])
            if "final" in c and c["final"]:
                exp = fdec(c["final"][0]["pr_total"])
                assert sc[0] == exp or (np.isnan(exp) and np.isnan(sc[0])), (c["mat"], c["k"], c["W"], float(sc[0]).hex(), c["final"][0]["pr_total"])
    finally:
        be.load_lm(None, 0)
        be.set_decode_math("glibc")


def test_golden_lm(be, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "beam_lm_cases.json")))
    mats = np.load(os.path.join(golden_dir, "beam_lm_mats.npz"))
    bad = []
    cur = None
    for c in g["cases"]:
        if cur != c["lm"]:
            be.load_lm(mats[c["lm"]], c["k"])
            cur = c["lm"]
        mat = mats[c["mat"]]
        lab = be.decode(mat, c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]))
        if s_of(lab) != c["seq"]:
            bad.append((c["mat"], c["k"], c["W"], c["s_thr"], c["r_thr"]))
    be.load_lm(None, 0)
    assert not bad, bad


def test_batch_vs_oracle_chunk_shapes(be, oracle):
    """A batch like BASELINE config 3: many 1024-row windows + 512-row tails, float32, W in {1,10,25}."""
    rng = np.random.default_rng(11)
    lens = [1024, 1024, 512, 1024, 700, 1, 0, 1024, 333, 64, 65, 63]
    rows = []
    for i, n in enumerate(lens):
        z = rng.normal(size=(n, 5)) * (1.0 if i % 2 == 0 else 3.0)
        if i % 3 == 0:
            z[:, 4] += 1.5
        rows.append(softmax_rows(z).astype(np.float32))
    mats = np.concatenate(rows, axis=0)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    for W in (1, 2, 6, 10, 12, 13, 25, 30):
        got = be.decode_batch(mats, off, lens, W)
        exp = oracle.beam_search_batch(mats, off, lens, W)
        for i in range(len(lens)):
            assert np.array_equal(got[i], exp[i]), (W, i, lens[i])


def test_global_shape_f64_with_lm_vs_oracle(be, oracle):
    """BASELINE config 4 shape: [4096,5] float64 assembled matrices, W=10, k-mer LM, thresholds 0.5/0.5."""
    rng = np.random.default_rng(12)
    k = 5
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    be.load_lm(table, k)
    lens = [4096, 2000, 4096]
    rows = []
    for i, n in enumerate(lens):
        z = rng.normal(size=(n, 5)) * 2.5
        z[:, 4] += 1.0
        rows.append(softmax_rows(z))
    mats = np.concatenate(rows, axis=0)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    for W in (6, 10):
        got = be.decode_batch(mats, off, lens, W, use_lm=True, s_threshold=0.5, r_threshold=0.5)
        exp = oracle.beam_search_batch(mats, off, lens, W, table, 0.5, 0.5, k)
        for i in range(len(lens)):
            assert np.array_equal(got[i], exp[i]), (W, i)
    be.load_lm(None, 0)


def test_lm_k11_table(be, oracle):
    """The 12-mer LM geometry of the reference (context 11 -> 4^11 x 4 table, 128 MiB)."""
    rng = np.random.default_rng(13)
    k = 11
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    be.load_lm(table, k)
    z = rng.normal(size=(1500, 5)) * 2.5
    z[:, 4] += 1.0
    mat = softmax_rows(z)
    got = be.decode(mat, 10, use_lm=True, s_threshold=0.5, r_threshold=0.5)
    exp, _ = oracle.beam_search_labels(mat, 10, table, 0.5, 0.5, k)
    assert np.array_equal(got, exp)
    be.load_lm(None, 0)


def test_float32_with_lm_vs_oracle(be, oracle):
    """single-window read in global mode: the matrix stays float32 (numpy-1.19 semantics in the oracle)."""
    rng = np.random.default_rng(14)
    k = 3
    table = rng.dirichlet([0.3] * 4, size=4 ** k)
    be.load_lm(table, k)
    for kind in (1.0, 3.0):
        mat = softmax_rows(rng.normal(size=(600, 5)) * kind).astype(np.float32)
        for W in (6, 10):
            got = be.decode(mat, W, use_lm=True, s_threshold=0.5, r_threshold=0.9)
            exp, _ = oracle.beam_search_labels(mat, W, table, 0.5, 0.9, k)
            assert np.array_equal(got, exp)
    be.load_lm(None, 0)


def test_properties_at_full_size(be):
    """Size-independent properties at BASELINE batch size: 512 windows x 1024 rows, W=10.
    (a) permutation invariance of the batch, (b) a sequence decodes the same alone and in the batch,
    (c) one-hot rows decode to the CTC collapse of the argmax path."""
    rng = np.random.default_rng(15)
    n, T = 512, 1024
    z = rng.normal(size=(n * T, 5)).astype(np.float32) * 2.0
    mats = softmax_rows(z).astype(np.float32)
    off = np.arange(n, dtype=np.int64) * T
    lens = np.full(n, T, dtype=np.int32)
    a = be.decode_batch(mats, off, lens, 10)
    perm = rng.permutation(n)
    b = be.decode_batch(mats, off[perm], lens[perm], 10)
    for i in range(n):
        assert np.array_equal(a[perm[i]], b[i])
    for i in (0, 17, 511):
        alone = be.decode(mats[i * T:(i + 1) * T], 10)
        assert np.array_equal(alone, a[i])
    # one-hot: beam search == greedy collapse
    path = rng.integers(0, 5, size=2000)
    hot = np.zeros((2000, 5), dtype=np.float32)
    hot[np.arange(2000), path] = 1.0
    exp = [int(c) for i, c in enumerate(path) if c != 4 and (i == 0 or path[i - 1] != c)]
    for W in (1, 10, 25):
        got = be.decode(hot, W)
        assert got.tolist() == exp


def test_errors(be):
    from radian_amd import RadianHipError
    m = np.full((4, 5), 0.2, dtype=np.float32)
    with pytest.raises(RadianHipError):
        be.decode(m, 0)
    with pytest.raises(RadianHipError):
        be.decode(m, be.max_beam_width + 1)
    with pytest.raises(RadianHipError):
        be.decode(m, 6, use_lm=True)  # no LM loaded
    assert be.decode(np.zeros((0, 5), dtype=np.float32), 6).tolist() == []


def test_sequence_too_long_for_the_backpointer_packing_is_refused():
    """(parent id << 2) | label lives in 32 bits and a sequence has up to 1 + W * rows trie nodes: 1 + W * rows >= 2^29 is
    RD_ERR_ARG at the boundary (host-side check), not a silently wrong traceback.  W = 51: 10 527 375 rows is the limit."""
    from radian_amd import Backend, RadianHipError
    be = Backend(0)
    try:
        W = be.max_beam_width
        lim = ((1 << 29) - 2) // W
        probs = np.zeros((8, 5), dtype=np.float32)
        probs[:, 4] = 1.0
        with pytest.raises(RadianHipError, match="2\\^29"):
            be.decode_batch(probs, [0], [lim + 1], W)        # (refused before any row is read)
        assert be.decode_batch(probs, [0], [8], W)[0].size == 0
    finally:
        be.close()


def test_two_sequences_per_wave_form(be, golden_dir, oracle):
    """W <= 12 as two sequences per wave (beam_search2_kernel: one candidate per lane of the half for W <= 6, two for 7 <= W <= 12 --
    round 4; rd_set_decode_form 3): every golden case of those widths -- no-LM
    incl. the exact-0 / duplicated-row matrices, LM with k in {1, 3, 5} -- with bit-exact winner scores in glibc arithmetic; then
    batches against the oracle where the two halves of a wave carry sequences of different lengths (incl. empty ones and an odd
    sequence count), float32 / float64 / exact-tie ("quant") rows, both arithmetics, with and without an LM."""
    from test_gpu_decode_stress import _mats
    be.set_decode_form("two")
    try:
        g = json.load(open(os.path.join(golden_dir, "beam_nolm_cases.json")))
        mats = np.load(os.path.join(golden_dir, "beam_nolm_mats.npz"))
        cases = [c for c in g["cases"] if c["W"] <= 12]
        assert len(cases) >= 90 and any(c["W"] == 10 for c in cases)
        for math in ("glibc", "fast"):
            be.set_decode_math(math)
            # all cases of one width in ONE batch: neighbours share waves
            for W in sorted({c["W"] for c in cases}):
                cs = [c for c in cases if c["W"] == W]
                rows = np.concatenate([mats[c["mat"]].reshape(-1, 5) for c in cs])
                lens = np.array([mats[c["mat"]].shape[0] for c in cs], dtype=np.int32)
                off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
                got, sc = be.decode_batch(rows, off, lens, W, with_scores=True)
                for c, lab, s in zip(cs, got, sc):
                    assert s_of(lab) == c["seq"], (math, c["mat"], W)
                    if math == "glibc" and "final" in c:
                        exp = fdec(c["final"][0]["pr_total"])
                        assert s == exp or (np.isnan(exp) and np.isnan(s)), (c["mat"], W, float(s).hex(), c["final"][0]["pr_total"])
        gl = json.load(open(os.path.join(golden_dir, "beam_lm_cases.json")))
        ml = np.load(os.path.join(golden_dir, "beam_lm_mats.npz"))
        be.set_decode_math("glibc")
        cur, n_lm = None, 0
        for c in gl["cases"]:
            if c["W"] > 12:
                continue
            if cur != c["lm"]:
                be.load_lm(ml[c["lm"]], c["k"])
                cur = c["lm"]
            mat = ml[c["mat"]]
            # the same sequence in both halves of a wave and alone in a wave (odd count): three copies
            rows = np.concatenate([mat.reshape(-1, 5)] * 3)
            T = mat.shape[0]
            got = be.decode_batch(rows, [0, T, 2 * T], [T, T, T], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]))
            assert all(s_of(x) == c["seq"] for x in got), (c["mat"], c["k"], c["W"])
            n_lm += 1
        assert n_lm >= 200
        be.load_lm(None, 0)
        # random batches against the oracle: ragged lengths, so the halves of a wave end at different steps
        rng = np.random.default_rng(66)
        table = rng.dirichlet([0.3] * 4, size=4 ** 3)
        # (round 6, suite budget: the widths at the ends of both candidate layouts -- 1, 6 | 7, 12 -- and one inside; one row type per kind
        # except the exact-tie rows, which run in both)
        for kind in ("flat", "peaky", "blocky", "quant"):
            for dt in {"flat": (np.float32,), "peaky": (np.float64,), "blocky": (np.float32,), "quant": (np.float32, np.float64)}[kind]:
                for W in (1, 3, 6, 7, 12):
                    m, off, lens = _mats(rng, 301, 260, kind, dt)
                    lens[::7] = 0
                    lens[5] = 1
                    for math in ("glibc", "fast"):
                        if kind == "quant" and math == "fast":
                            continue            # (exact ties are the glibc arithmetic's promise)
                        be.set_decode_math(math)
                        for lm in (False, True):
                            if lm:
                                be.load_lm(table, 3)
                            exp = oracle.beam_search_batch(m, off, lens, W, table if lm else None, 0.4, 0.9, 3 if lm else 0)
                            got = be.decode_batch(m, off, lens, W, use_lm=lm, s_threshold=0.4, r_threshold=0.9)
                            bad = [i for i in range(len(lens)) if not np.array_equal(got[i], exp[i])]
                            assert not bad, (kind, dt.__name__, W, math, lm, bad[:6], len(bad))
                            if lm:
                                be.load_lm(None, 0)
    finally:
        be.load_lm(None, 0)
        be.set_decode_form("auto")
        be.set_decode_math("glibc")


def test_sparse_lm_reports_exactly_the_reads_the_reference_fails_on(be, golden_dir, oracle):
    """Round 4: an RNA model that lacks contexts.  204 cases generated from the imported reference: the labeling, or
    the KeyError of decode.py:83 -- here RD_LEN_MISSING_CONTEXT for that sequence (None from the binding).  Every case alone and in
    a batch beside sequences that do not reach the context; every launch form its width has; both arithmetics; float32 rows too
    (vs the oracle, itself pinned by the same goldens)."""
    g = json.load(open(os.path.join(golden_dir, "beam_lm_sparse_cases.json")))["cases"]
    mats = np.load(os.path.join(golden_dir, "beam_lm_sparse_mats.npz"))
    bad = []
    try:
        for math in ("glibc", "fast"):
            be.set_decode_math(math)
            for c in g:
                table = mats[c["lm"]].copy()
                table[c["missing"]] = np.nan
                be.load_lm(table, c["k"])
                mat = mats[c["mat"]]
                for form in (("auto", "one", "two") if c["W"] <= 12 else ("auto",)):
                    be.set_decode_form(form)
                    # the case twice in one launch (both halves of a two-sequence wave) + a one-row sequence that cannot reach any context
                    rows = np.concatenate([mat, mat, mat[:1]])
                    got = be.decode_batch(rows, [0, len(mat), 2 * len(mat)], [len(mat), len(mat), 1], c["W"], use_lm=True,
                                          s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]))
                    exp = None if "key_error" in c else c["seq"]
                    for lab in got[:2]:
                        if (None if lab is None else s_of(lab)) != exp:
                            bad.append((math, form, c["mat"], c["k"], c["W"]))
                    assert got[2] is not None and len(got[2]) <= 1
        assert not bad, bad[:10]
        # a batch of ragged random sequences against the oracle: float32 and float64 rows, W incl. the wide forms
        rng = np.random.default_rng(77)
        k = 3
        table = rng.dirichlet([0.3] * 4, size=4 ** k)
        table[rng.choice(4 ** k, size=6, replace=False)] = np.nan
        be.load_lm(table, k)
        be.set_decode_form("auto")
        for dtype in (np.float64, np.float32):
            lens = [int(x) for x in rng.integers(1, 90, size=64)]
            rows = np.concatenate([softmax_rows(rng.normal(size=(n, 5)) * rng.choice([0.7, 2.0])) for n in lens]).astype(dtype)
            off = np.concatenate([[0], np.cumsum(lens)[:-1]])
            for W in (1, 4, 6, 10, 13, 25, 30):
                for math in ("glibc", "fast"):
                    be.set_decode_math(math)
                    got = be.decode_batch(rows, off, lens, W, use_lm=True, s_threshold=0.2, r_threshold=1.5)
                    exp = oracle.beam_search_batch(rows, off, lens, W, table, 0.2, 1.5, k)
                    n_none = 0
                    for i in range(len(lens)):
                        assert (got[i] is None) == (exp[i] is None), (dtype.__name__, W, math, i)
                        n_none += got[i] is None
                        assert got[i] is None or np.array_equal(got[i], exp[i]), (dtype.__name__, W, math, i)
                    assert 0 < n_none < len(lens)
    finally:
        be.load_lm(None, 0)
        be.set_decode_math("glibc")
        be.set_decode_form("auto")


# ------------------------------------------------------------------------------ BASELINE.json's own geometries (round 5)
def _baseline(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "beam_baseline_cases.json")))
    return g, np.load(os.path.join(golden_dir, "beam_baseline_mats.npz"))


def test_baseline_global_k11_assembled_4096_vs_reference(be, golden_dir):
    """configs[3]'s decode against the imported reference directly (not through the oracle): the reference's window probabilities ->
    rd_assemble -> [4096,5] float64 (bytes == matrix_assembly.assemble_matrices') -> beam search with the reference's default
    context 11 (4^11-row table), W in {6, 10, 25}: labelings equal, the winner's pr_total bit-equal in glibc arithmetic."""
    from _golden_lm import checked_k11_table, table_sha256
    g, arr = _baseline(golden_dir)
    table = checked_k11_table(g["k11_table_sha256"])
    be.set_decode_math("glibc")
    be.load_lm(table, g["cases"][0]["k"])
    try:
        mats, n = {}, 0
        for c in g["cases"]:
            if c["group"] != "global_k11":
                continue
            if c["probs"] not in mats:
                m = be.assemble(arr[c["probs"]], c["pad"], c["step"])
                assert m.dtype == np.float64 and table_sha256(m) == c["mat_sha256"]
                mats[c["probs"]] = m
            m = mats[c["probs"]]
            (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]),
                                         with_scores=True)
            assert s_of(lab) == c["seq"], (c["probs"], c["W"], c["s_thr"])
            assert sc[0] == fdec(c["final"][0]["pr_total"]), (c["probs"], c["W"], float(sc[0]).hex(), c["final"][0]["pr_total"])
            n += 1
        assert n == 17
    finally:
        be.load_lm(None, 0)


def test_baseline_global_k11_sparse_vs_reference(be, golden_dir):
    """12-mer models lacking 7 ... 8 572 contexts: RD_LEN_MISSING_CONTEXT exactly for the reads on which decode.py:83 raises KeyError."""
    from _golden_lm import checked_k11_table, sparse_table
    g, arr = _baseline(golden_dir)
    table = checked_k11_table(g["k11_table_sha256"])
    try:
        n_err = 0
        for c in g["cases"]:
            if c["group"] != "global_k11_sparse":
                continue
            be.load_lm(sparse_table(table, c["missing"]), c["k"])
            m = be.assemble(arr[c["probs"]], c["pad"], c["step"])
            lab = be.decode(m, c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]))
            if "key_error" in c:
                assert lab is None, (c["probs"], c["W"], len(c["missing"]))
                n_err += 1
            else:
                assert lab is not None and s_of(lab) == c["seq"], (c["probs"], c["W"], len(c["missing"]))
        assert n_err == 8
    finally:
        be.load_lm(None, 0)


def test_baseline_wide_beams_vs_reference(be, golden_dir):
    """Beam widths 26 ... 100 against the imported reference: the four-wave launch form (26 ... 51) and the widths beyond it."""
    g, arr = _baseline(golden_dir)
    be.set_decode_math("glibc")
    try:
        n = 0
        for c in g["cases"]:
            if c["group"] not in ("wide_nolm", "wide_lm"):
                continue
            m = arr[c["mat"]]
            if "lm" in c:
                be.load_lm(arr[c["lm"]], c["k"])
                (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]),
                                             with_scores=True)
            else:
                be.load_lm(None, 0)
                (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], with_scores=True)
            assert s_of(lab) == c["seq"], (c["group"], c["mat"], c["W"])
            if "final" in c:
                exp = fdec(c["final"][0]["pr_total"])
                assert sc[0] == exp or (np.isnan(exp) and np.isnan(sc[0])), (c["group"], c["mat"], c["W"], float(sc[0]).hex(), c["final"][0]["pr_total"])
            n += 1
        assert n == 62
    finally:
        be.load_lm(None, 0)


@pytest.mark.parametrize("name,n_cases,widths", [("beam_wide128", 50, (65, 90, 127, 128)), ("beam_wide256", 40, (129, 200, 255, 256))])
def test_wide128_beams_vs_reference(be, golden_dir, name, n_cases, widths):
    """Round 6: the five-wave shape (65 ... 128 beams, the beam set in two halves) and the ten-wave shape (129 ... 256, four parts) against the imported
    reference's own beam_search at W = 65 / 90 / 100 / 127 / 128 and 129 / 200 / 255 / 256: every labeling, and the winner's pr_total bit for bit in
    glibc arithmetic; one sequence per launch and all cases of a width in one launch (neighbouring workgroups)."""
    g = json.load(open(os.path.join(golden_dir, name + "_cases.json")))
    arr = np.load(os.path.join(golden_dir, name + "_mats.npz"))
    be.set_decode_math("glibc")
    try:
        n = 0
        for c in g["cases"]:
            m = arr[c["mat"]]
            if "lm" in c:
                be.load_lm(arr[c["lm"]], c["k"])
                (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]), with_scores=True)
            else:
                be.load_lm(None, 0)
                (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], with_scores=True)
            assert s_of(lab) == c["seq"], (c["group"], c["mat"], c["W"])
            exp = fdec(c["final"][0]["pr_total"])
            assert sc[0] == exp or (np.isnan(exp) and np.isnan(sc[0])), (c["group"], c["mat"], c["W"], float(sc[0]).hex(), c["final"][0]["pr_total"])
            n += 1
        assert n == n_cases
        be.load_lm(None, 0)
        for W in widths:
            cs = [c for c in g["cases"] if c["W"] == W and "lm" not in c]
            rows = np.concatenate([arr[c["mat"]].reshape(-1, 5) for c in cs])
            lens = np.array([arr[c["mat"]].shape[0] for c in cs], dtype=np.int32)
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            got = be.decode_batch(rows, off, lens, W)
            assert [s_of(x) for x in got] == [c["seq"] for c in cs], W
    finally:
        be.load_lm(None, 0)


def test_baseline_single_window_float32_lm_vs_reference(be, golden_dir):
    """Reads shorter than one chunk in global mode (float32 matrix + LM), k = 3 and k = 11; cases are numpy-version-independent
    (tests/golden/make_golden.py gen_beam_baseline, group single_window_f32_lm)."""
    from _golden_lm import checked_k11_table
    g, arr = _baseline(golden_dir)
    table = checked_k11_table(g["k11_table_sha256"])
    be.set_decode_math("glibc")
    try:
        cur, n = None, 0
        for c in g["cases"]:
            if c["group"] != "single_window_f32_lm":
                continue
            if cur != c["lm"]:
                be.load_lm(table if c["lm"] == "k11" else arr[c["lm"]], c["k"])
                cur = c["lm"]
            m = arr[c["mat"]]
            (lab,), sc = be.decode_batch(m, [0], [m.shape[0]], c["W"], use_lm=True, s_threshold=fdec(c["s_thr"]), r_threshold=fdec(c["r_thr"]),
                                         with_scores=True)
            assert s_of(lab) == c["seq"], (c["mat"], c["lm"], c["W"])
            assert sc[0] == fdec(c["winner_pr_total"])
            n += 1
        assert n == 36
    finally:
        be.load_lm(None, 0)


def test_wide_beams_vs_oracle(be, oracle):
    """Widths above the wave-per-sequence kernels' 51 (decode_wide.hip; the reference slices with any --beam-width, decode.py:145):
    batches with empty / one-row / ragged sequences, float32 and float64 rows, exact-0 probabilities (ties), with a 3-mer LM (dense
    and sparse), both arithmetics, against the oracle; the width where the two kernels meet (51 | 52) included."""
    assert be._L.rd_decode_lane_width() == 256 and be.max_beam_width >= 1024
    rng = np.random.default_rng(2026)
    lens = [300, 0, 1, 64, 65, 129, 512, 7]
    rows = []
    for i, n in enumerate(lens):
        z = rng.normal(size=(n, 5)) * (1.0 if i % 2 == 0 else 3.5)
        m = softmax_rows(z)
        if i % 3 == 2 and n:
            m[rng.integers(0, n, size=max(1, n // 5)), rng.integers(0, 5)] = 0.0     # exact zeros: -inf scores and ties
        rows.append(m)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    table = rng.dirichlet([0.3] * 4, size=4 ** 3)
    sparse = table.copy()
    sparse[[5, 17, 40]] = np.nan
    for math in ("glibc", "fast"):
        be.set_decode_math(math)
        for dtype in (np.float32, np.float64):
            mats = np.concatenate(rows, axis=0).astype(dtype)
            for W in (51, 52, 63, 64, 65, 100, 127, 128, 129, 200, 256, 257, 1024):       # (52 ... 64: four waves, two candidates per lane; 65 ... 128: five waves, the beam set in two halves -- round 6; above: decode_wide.hip)
                if math == "fast" and W not in (52, 64, 100, 128, 256):
                    continue
                be.load_lm(None, 0)
                got = be.decode_batch(mats, off, lens, W)
                exp = oracle.beam_search_batch(mats, off, lens, W)
                for i in range(len(lens)):
                    assert np.array_equal(got[i], exp[i]), (math, dtype, W, i, "no LM")
                if W in (52, 64, 100, 128, 129, 256, 257):
                    for tb in (table, sparse):
                        be.load_lm(tb, 3)
                        got = be.decode_batch(mats, off, lens, W, use_lm=True, s_threshold=0.5, r_threshold=0.9)
                        exp = oracle.beam_search_batch(mats, off, lens, W, tb, 0.5, 0.9, 3)
                        for i in range(len(lens)):
                            if exp[i] is None:
                                assert got[i] is None, (math, dtype, W, i, "sparse")
                            else:
                                assert got[i] is not None and np.array_equal(got[i], exp[i]), (math, dtype, W, i, "LM")
    be.load_lm(None, 0)
    be.set_decode_math("glibc")


def test_work_queue_form_matches_the_reference(be, golden_dir, oracle):
    """beam_search_queue_kernel (rd_set_decode_form 5: 16 waves' worth of resident workgroups take the sequences of a launch from a counter,
    one after the other -- what the reads pipeline launches for a group larger than its decode partition): every golden case of the
    wave-per-sequence widths, with and without an LM, bit-equal winner scores; then batches of hundreds of ragged sequences (each workgroup
    runs dozens of sequences back to back: the per-sequence LDS state must start clean every time) against the oracle, dense and sparse LM."""
    be.set_decode_form("queue")
    be.set_decode_math("glibc")
    try:
        g = json.load(open(os.path.join(golden_dir, "beam_nolm_cases.json")))
        mats = np.load(os.path.join(golden_dir, "beam_nolm_mats.npz"))
        for c in g["cases"]:
            m = mats[c["mat"]]
            (lab,), sc = be.decode_batch(m.reshape(-1, 5), [0], [m.shape[0]], c["W"], with_scores=True)
            assert s_of(lab) == c["seq"], (c["mat"], c["W"])
            if "final" in c:
                exp = fdec(c["final"][0]["pr_total"])
                assert sc[0] == exp or (np.isnan(exp) and np.isnan(sc[0]))
        gb, arr = _baseline(golden_dir)
        for c in gb["cases"]:
            if c["group"] == "wide_nolm" and c["W"] <= 51:
                (lab,), _ = be.decode_batch(arr[c["mat"]], [0], [arr[c["mat"]].shape[0]], c["W"], with_scores=True)
                assert s_of(lab) == c["seq"], (c["mat"], c["W"])
        rng = np.random.default_rng(77)
        n = 700
        lens = rng.integers(0, 90, size=n)
        lens[::50] = 400
        rows = [softmax_rows(rng.normal(size=(int(k), 5)) * (1.0 if i % 2 else 3.0)) for i, k in enumerate(lens)]
        mat64 = np.concatenate(rows, axis=0)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]])
        table = rng.dirichlet([0.3] * 4, size=4 ** 3)
        sparse = table.copy()
        sparse[[3, 30]] = np.nan
        for W in (1, 6, 10, 25, 40):
            for dtype in (np.float32, np.float64):
                mats_b = mat64.astype(dtype)
                be.load_lm(None, 0)
                got = be.decode_batch(mats_b, off, lens, W)
                exp = oracle.beam_search_batch(mats_b, off, lens, W)
                assert all(np.array_equal(a, b) for a, b in zip(got, exp)), (W, dtype)
            for tb in (table, sparse):
                be.load_lm(tb, 3)
                got = be.decode_batch(mat64, off, lens, W, use_lm=True, s_threshold=0.4, r_threshold=0.9)
                exp = oracle.beam_search_batch(mat64, off, lens, W, tb, 0.4, 0.9, 3)
                for a, b in zip(got, exp):
                    assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), W
    finally:
        be.load_lm(None, 0)
        be.set_decode_form("auto")


def test_baseline_chunk_route_vs_reference(be, golden_dir):
    """configs[2]'s chunk route at its own geometry against the imported reference directly: the reference's window probabilities -> the GPU's
    per-window beam search (W = 10 / 6 / 1, valid rows = the pad trim) -> the native stitch (rd_stitch_chunk): fragments and consensus equal."""
    from radian_amd.sequence_assembly import consensus_batch
    g = json.load(open(os.path.join(golden_dir, "pipeline_baseline_cases.json")))
    arr = np.load(os.path.join(golden_dir, "pipeline_baseline.npz"))
    be.load_lm(None, 0)
    for c in g["cases"]:
        probs = arr[c["probs"]]
        nW, T, _ = probs.shape
        lens = np.full(nW, T, dtype=np.int32)
        lens[-1] = T - c["pad"]
        off = np.arange(nW, dtype=np.int64) * T
        labs = be.decode_batch(probs.reshape(-1, 5), off, lens, c["W"])
        assert [s_of(l) for l in labs] == c["chunk_fragments"], (c["step"], c["kind"], c["W"])
        lab2d = np.zeros((nW, T), dtype=np.uint8)
        ll = np.zeros(nW, dtype=np.int32)
        for i, l in enumerate(labs):
            lab2d[i, : len(l)] = l
            ll[i] = len(l)
        assert consensus_batch(lab2d, ll, [nW], threads=2) == [c["chunk_seq"]]
