"""Host-side product logic without a GPU: preprocess / stitch / LM loader against the golden vectors, fast5 and
Keras-h5 readers, the CLI contract and the driver loop (with the oracle-backed test double)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_product_preprocess_matches_golden(golden_dir):
    from radian_amd import preprocess as P
    g = json.load(open(os.path.join(golden_dir, "preprocess_cases.json")))
    arr = np.load(os.path.join(golden_dir, "preprocess.npz"))
    for c in g["cases"]:
        sig = arr["sig_" + c["name"]]
        if "error" in c:
            with pytest.raises(ValueError) as ei:
                P.mad_normalise(sig, c["clip"])
            assert ei.value.args[0] == c["error"]
            continue
        n = P.mad_normalise(sig, c["clip"])
        assert str(n.dtype) == c["norm_dtype"] and np.array_equal(n, arr["norm_" + c["name"]]), c["name"]
        w, pad = P.get_windows(n, c["chunk"], c["step"])
        assert pad == c["pad"] and str(w.dtype) == c["win_dtype"] and np.array_equal(w, arr["win_" + c["name"]]), c["name"]
    for e in g["window_errors"]:
        with pytest.raises(ValueError) as ei:
            P.get_windows(np.zeros(100), e["chunk"], e["step"])
        assert ei.value.args[0] == e["error"]


def test_product_stitch_matches_golden(golden_dir):
    from radian_amd import sequence_assembly as S
    g = json.load(open(os.path.join(golden_dir, "seq_assembly_cases.json")))
    for c in g["cases"]:
        cons = S.simple_assembly(c["fragments"])
        assert list(cons.shape) == c["consensus_shape"]
        assert np.array_equal(cons.astype(np.int64), np.array(c["consensus"], dtype=np.int64).reshape(cons.shape))
        assert S.consensus_sequence(c["fragments"]) == c["seq"]


def test_product_stitch_matches_reference_random_cases(golden_dir):
    """160 random fragment lists generated through the reference's own simple_assembly (make_golden.py
    gen_seq_assembly_random): vote matrix by SHA-256, consensus string, and the lists on which the reference raises"""
    import hashlib
    from radian_amd import sequence_assembly as S
    g = json.load(open(os.path.join(golden_dir, "seq_assembly_random_cases.json")))
    assert len(g["cases"]) == 160 and sum("error" in c for c in g["cases"]) >= 8
    for i, c in enumerate(g["cases"]):
        if "error" in c:
            with pytest.raises(IndexError):
                S.simple_assembly(c["fragments"])
            continue
        cons = S.simple_assembly(c["fragments"])
        assert list(cons.shape) == c["consensus_shape"], i
        assert hashlib.sha256(np.ascontiguousarray(cons, dtype=np.int64).tobytes()).hexdigest() == c["consensus_sha256"], i
        assert S.consensus_sequence(c["fragments"]) == c["seq"], i


def test_product_stitch_matches_oracle_hypothesis(oracle):
    """random fragment lists (overlapping, unrelated, empty, lower case, >= 200 characters = difflib autojunk, more than
    1000 columns = the reference's matrix growth) against the oracle's statement-level restatement of
    sequence_assembly.py:19-48; the IndexError of the reference's fixed growth step is kept too"""
    from hypothesis import given, settings, strategies as st
    from radian_amd import sequence_assembly as S

    @st.composite
    def fragment_lists(draw):
        n = draw(st.integers(0, 7))
        base = draw(st.text("ACGT", min_size=0, max_size=700))
        frags = []
        pos = 0
        for _ in range(n):
            kind = draw(st.integers(0, 4))
            if kind == 0 or not base:
                f = draw(st.text("ACGTacgt", min_size=0, max_size=draw(st.sampled_from([5, 40, 260, 450]))))
            else:   # a noisy slice of a common sequence, advancing like consecutive windows do
                pos = min(len(base), pos + draw(st.integers(0, 120)))
                f = base[pos: pos + draw(st.integers(0, 420))]
                if f and draw(st.booleans()):
                    i = draw(st.integers(0, len(f) - 1))
                    f = f[:i] + draw(st.sampled_from("ACGT")) + f[i + 1:]
            frags.append(f)
        return frags

    @settings(max_examples=300, deadline=None)
    @given(fragment_lists())
    def check(frags):
        try:
            exp = oracle.simple_assembly(frags)
        except IndexError:
            with pytest.raises(IndexError):
                S.simple_assembly(frags)
            return
        got = S.simple_assembly(frags)
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp)
        assert S.consensus_sequence(frags) == oracle.chunk_consensus(frags)

    check()
    # deterministic corner cases: one fragment (L = 0), growth by exactly one step, overflow of the fixed step, bad letters
    long = "ACGT" * 260                                   # 1040 characters
    assert S.simple_assembly(["ACGT"]).shape == (4, 0) and S.consensus_sequence(["ACGTT"]) == ""
    assert np.array_equal(S.simple_assembly(["A" * 900, "C" * 950]), oracle.simple_assembly(["A" * 900, "C" * 950]))
    for bad in ([long], ["A" * 990, "C", "C" + "G" * 1012]):
        with pytest.raises(IndexError):
            oracle.simple_assembly(bad)
        with pytest.raises(IndexError):
            S.simple_assembly(bad)
    with pytest.raises(KeyError):
        S.simple_assembly(["ACGT", "ACNT"])
    v = np.zeros((4, 6))
    S.add_count(v, -2, "ACGTA")
    assert v[:, :3].tolist() == [[0, 0, 1], [0, 0, 0], [1, 0, 0], [0, 1, 0]] and v.sum() == 3
    assert S.index2base([0, 3, 2, 1]) == "ATGC"


def test_lm_json_loader(tmp_path):
    from radian_amd import lm
    rng = np.random.default_rng(0)
    k = 3
    raw = {}
    for i in range(4 ** k):
        ctx = "".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k))
        raw[ctx] = [float(x) for x in rng.dirichlet([0.3] * 4)]
    p = tmp_path / "lm.json"
    p.write_text(json.dumps(raw))
    table, kk = lm.load_json(str(p))
    assert kk == k and table.shape == (64, 4)
    # index = base-4 number of the context, first char most significant (tuple keys of basecall.py:56)
    assert np.array_equal(table[lm.context_index("GTA")], raw["GTA"])
    assert lm.context_index("GTA") == 2 * 16 + 3 * 4 + 0 == lm.context_index((2, 3, 0))
    assert lm.n_missing(table) == 0
    # tuple keys (the reference's re-keyed dict, basecall.py:51-57) give the same table
    t2, k2 = lm.table_from_dict({tuple("ACGT".index(b) for b in c): d for c, d in raw.items()})
    assert k2 == k and np.array_equal(t2, table)
    # a sparse model loads (the reference's dict simply lacks the key; decode.py:83 fails only when it is looked up):
    # absent contexts are rows of NaN
    del raw["AAA"], raw["GTC"]
    p.write_text(json.dumps(raw))
    sparse, _ = lm.load_json(str(p))
    assert lm.n_missing(sparse) == 2 and np.isnan(sparse[0]).all() and np.isnan(sparse[lm.context_index("GTC")]).all()
    keep = ~np.isnan(sparse[:, 0])
    assert np.array_equal(sparse[keep], table[keep])
    for bad in ({"AAX": [0.25] * 4, "AAA": [0.25] * 4}, {"AA": [0.25] * 4, "AAA": [0.25] * 4}, {"AAA": [0.5, 0.5]}, {}):
        with pytest.raises(ValueError):
            lm.table_from_dict(bad)


def test_lm_loader_vectorised_at_scale():
    """4^9 = 262 144 contexts in shuffled order, 1000 of them absent: the key -> row conversion is one vectorised pass.  (The real
    model has 4^11 = 4 194 304 twelve-mer contexts, models/rnamodel_12mer_pc.json, basecall.py:28: tools/lm_load_bench.py writes
    a JSON of that size, times the whole route and reports the peak RSS -- DESIGN.md section 4.13.)"""
    import time
    from radian_amd import lm
    k = 9
    n = 4 ** k
    rng = np.random.default_rng(1)
    vals = rng.random((n, 4))
    letters = np.array(list("ACGT"))
    digits = (np.arange(n)[:, None] >> (2 * np.arange(k - 1, -1, -1))) & 3
    keys = ["".join(r) for r in letters[digits]]
    perm = rng.permutation(n)
    model = {keys[i]: vals[i].tolist() for i in perm[: n - 1000]}       # (shuffled order, 1000 contexts absent)
    t0 = time.time()
    table, kk = lm.table_from_dict(model)
    dt = time.time() - t0
    assert kk == k and table.shape == (n, 4) and lm.n_missing(table) == 1000
    held = np.sort(perm[: n - 1000])
    assert np.array_equal(table[held], vals[held])
    assert dt < 10, f"table_from_dict took {dt:.1f} s for 4^9 contexts"


def _have_hdf5():
    try:
        from radian_amd import h5
        h5.lib()
        return True
    except Exception:
        return False


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_fast5_reader_roundtrip_and_reference_file(tmp_path, golden_dir):
    from radian_amd import fast5
    ids = json.load(open(os.path.join(golden_dir, "reads_fast5_ids.json")))["read_ids"]
    sig = np.load(os.path.join(golden_dir, "reads_fast5_signals.npz"))
    sub = tmp_path / "a" / "b"
    sub.mkdir(parents=True)
    fast5.write_multi_fast5(str(sub / "x.fast5"), {r: sig[r] for r in ids})
    got = [(r.read_id, r.get_raw_data()) for r in fast5.iter_directory(str(tmp_path))]  # recursive like rglob
    assert [g[0] for g in got] == ids
    assert all(a.dtype == np.int16 and np.array_equal(a, sig[r]) for r, a in got)
    ref = "/root/reference/radian/data/reads.fast5"   # build container only
    if os.path.exists(ref):
        got = [(r.read_id, r.get_raw_data()) for r in fast5.iter_reads(ref)]
        assert [g[0] for g in got] == ids
        assert all(np.array_equal(a, sig[r]) for r, a in got)


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_keras_h5_converter_roundtrip(tmp_path):
    from radian_amd import h5weights, weights
    w = weights.synthetic_weights(seed=3)
    p = str(tmp_path / "sig2seq.h5")
    h5weights.write_keras_weights(p, w)
    assert np.array_equal(h5weights.read_keras_weights(p), w)
    with pytest.raises(ValueError):
        h5weights.read_keras_weights(p, dilations=(1, 2, 4))
    # the forms a Keras / h5py-written file can take (written here through libhdf5 in exactly those HDF5 encodings):
    # NumPy-'S' arrays (NUL-padded fixed strings, Keras 2.4 + h5py 2.10 -- the reference's pin), variable-length
    # strings (h5py 3), NUL-terminated; the tree at the root (save_weights) or under /model_weights (model.save);
    # `weight_names` split into weight_names0.. chunks
    for i, (kind, root, chunk) in enumerate([("nullpad", "/", 0), ("vlen", "/", 0), ("nullterm", "/model_weights", 0),
                                             ("vlen", "/model_weights", 7), ("nullpad", "/", 5)]):
        q = str(tmp_path / f"form{i}.h5")
        h5weights.write_keras_weights(q, w, attr_kind=kind, root=root, chunk_names=chunk)
        assert np.array_equal(h5weights.read_keras_weights(q), w), (kind, root, chunk)
    from radian_amd import h5
    with h5.File(str(tmp_path / "form0.h5"), "r") as f:     # exact-fit NUL-padded strings come back whole
        assert f.attr("/", "layer_names") == ["inputs", "tcn", "dense", "activation", "dense_1", "activation_1"]
        assert f.attr("/tcn", "weight_names")[0] == "tcn/residual_block_0/conv1D_0/kernel:0"
    (tmp_path / "junk.h5").write_bytes(b"not hdf5")
    with pytest.raises(Exception):
        h5weights.read_keras_weights(str(tmp_path / "junk.h5"))


def test_cli_flags_match_reference():
    """radian/basecall.py:19-35: names, defaults and types of the 13 flags + 2 positionals."""
    from radian_amd import basecall
    a = basecall.build_parser().parse_args(["in", "out"])
    assert (a.fast5_dir, a.fasta_dir) == ("in", "out")
    assert a.local is False and a.chunk_len == 1024 and a.step_size == 128 and a.batch_size == 32
    assert a.outlier_clip == 4 and isinstance(a.outlier_clip, int)
    assert a.rna_model == "models/rnamodel_12mer_pc.json" and a.sig_model == "models/sig2seq.h5"
    assert a.sig_config == "models/sig2seq.yaml" and a.beam_width == 6 and a.decode_type == "global"
    assert a.sig_threshold == 0.5 and a.rna_threshold == 0.5 and a.context_len == 11
    with pytest.raises(SystemExit):
        basecall.build_parser().parse_args(["in", "out", "--decode-type", "greedy"])


def test_fasta_writer_rotation(tmp_path):
    from radian_amd.basecall import FastaWriter
    w = FastaWriter(str(tmp_path))
    for i in range(2000):
        w.write(f"r{i}", "ACGT")
    w.close()
    files = sorted(os.listdir(tmp_path))
    assert files == ["reads-0.fasta", "reads-1.fasta", "reads-2.fasta"]  # empty trailing file, like the reference
    assert open(tmp_path / "reads-0.fasta").read().startswith(">r0\nTGCA\n")  # reversed to 5'->3'
    assert sum(1 for _ in open(tmp_path / "reads-1.fasta")) == 2000
    assert os.path.getsize(tmp_path / "reads-2.fasta") == 0


@pytest.mark.parametrize("mode", ["chunk", "global"])
def test_driver_loop_with_test_double(mode, oracle, capsys):
    """basecall.run: cross-read batching gives the same result as one read per batch; bad reads are skipped with the
    reference's messages (basecall.py:77-82)."""
    from radian_amd import basecall, weights
    from _oracle_backend import OracleBackend
    from _reads import golden_reads
    be = OracleBackend()
    be.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2)), (1, 2))
    be.load_lm(np.random.default_rng(9).dirichlet([0.3] * 4, size=16), 2)
    base = ["a", "b", "--chunk-len", "128", "--step-size", "64", "--beam-width", "3", "--decode-type", mode, "--context-len", "2"]
    outs = []
    for gbw in ("1", "40"):
        args = basecall.build_parser().parse_args(base + ["--gpu-batch-windows", gbw])
        args._lm_loaded = True
        outs.append(basecall.run(args, be, reads=golden_reads(700, extra_bad=True), writer=None))
    assert outs[0] == outs[1]
    assert len(outs[0]) == 5 and [i for i, _, _ in outs[0]] == [0, 1, 3, 5, 6]
    text = capsys.readouterr().out
    assert "flat-signal signal issue, skipping this read." in text
    assert "('MAD is zero, issue with signal.',)" in text
    assert "('Signal must not be empty to normalise',)" in text
    assert text.count("Basecalled read ") == 10


def test_driver_loop_stitch_workers(oracle):
    """chunk mode with the fragment stitch in spawned worker processes == stitch on the driver thread, same order"""
    from radian_amd import basecall, weights
    from _oracle_backend import OracleBackend
    from _reads import golden_reads
    be = OracleBackend()
    be.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2)), (1, 2))
    args = basecall.build_parser().parse_args(["a", "b", "--chunk-len", "128", "--step-size", "64", "--beam-width", "3",
                                                "--decode-type", "chunk", "--gpu-batch-windows", "40"])
    args._lm_loaded = False
    base = basecall.run(args, be, reads=golden_reads(700, extra_bad=True), writer=None)
    pool = basecall.make_stitch_pool(2)
    assert pool is not None
    try:
        got = basecall.run(args, be, reads=golden_reads(700, extra_bad=True), writer=None, stitch_pool=pool)
    finally:
        pool.shutdown()
    assert got == base and len(got) == 5


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 needed to write the fixtures")
def test_pure_python_hdf5_reader(tmp_path, golden_dir, monkeypatch):
    """radian_amd.h5pure (no libhdf5) reads the same fast5 / Keras files: groups in name order, int16 chunked signals,
    float32 2-D weights, fixed-string attributes; through fast5.iter_reads with RADIAN_HDF5_PURE=1 too."""
    from radian_amd import fast5, h5pure, h5weights, weights
    ids = json.load(open(os.path.join(golden_dir, "reads_fast5_ids.json")))["read_ids"]
    sig = np.load(os.path.join(golden_dir, "reads_fast5_signals.npz"))
    p = str(tmp_path / "x.fast5")
    fast5.write_multi_fast5(p, {r: sig[r] for r in ids})
    f = h5pure.PureFile(p)
    assert f.keys("/") == sorted("read_" + r for r in ids)
    for r in ids:
        a = f.read(f"/read_{r}/Raw/Signal")
        assert a.dtype == np.int16 and np.array_equal(a, sig[r])
        assert f.attr(f"/read_{r}/Raw", "read_id") == r
    assert f.exists(f"/read_{ids[0]}/Raw/Signal") and not f.exists("/nope/Raw")
    monkeypatch.setenv("RADIAN_HDF5_PURE", "1")
    got = [(r.read_id, r.get_raw_data()) for r in fast5.iter_reads(p)]
    assert [g[0] for g in got] == ids and all(np.array_equal(a, sig[r]) for r, a in got)
    w = weights.synthetic_weights(seed=3)
    q = str(tmp_path / "sig2seq.h5")
    monkeypatch.delenv("RADIAN_HDF5_PURE")
    h5weights.write_keras_weights(q, w)
    monkeypatch.setenv("RADIAN_HDF5_PURE", "1")
    assert np.array_equal(h5weights.read_keras_weights(q), w)
    ref = "/root/reference/radian/data/reads.fast5"   # build container only: compact link-message groups inside
    if os.path.exists(ref):
        got = [(r.read_id, r.get_raw_data()) for r in fast5.iter_reads(ref)]
        assert [g[0] for g in got] == ids and all(np.array_equal(a, sig[r]) for r, a in got)


def _label_matrix(frag_lists, chunk_len):
    nw = [len(c) for c in frag_lists]
    lab = np.zeros((sum(nw), chunk_len), dtype=np.uint8)
    lens = np.zeros(sum(nw), dtype=np.int32)
    w = 0
    for c in frag_lists:
        for f in c:
            lab[w, : len(f)] = f
            lens[w] = len(f)
            w += 1
    return lab, lens, nw


def test_native_stitch_matches_reference_goldens(golden_dir):
    """rd_stitch_chunk (csrc/stitch.hip: simple_assembly + difflib restated in C++, the driver's chunk-mode host stage) on the
    reference-generated cases whose fragments are plain upper-case ACGT (the device emits labels, which have no case): same
    consensus strings; the cases on which the reference raises IndexError raise here too."""
    from radian_amd import sequence_assembly as S
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    n_ok = n_err = 0
    for name in ("seq_assembly_cases.json", "seq_assembly_random_cases.json"):
        for c in json.load(open(os.path.join(golden_dir, name)))["cases"]:
            if not all(set(f) <= set("ACGT") for f in c["fragments"]):
                continue
            frags = [np.array([code[ch] for ch in f], dtype=np.uint8) for f in c["fragments"]]
            lab, lens, nw = _label_matrix([frags], max([1] + [len(f) for f in frags]))
            if "error" in c:
                with pytest.raises(IndexError):
                    S.consensus_batch(lab, lens, nw, threads=1)
                n_err += 1
            else:
                assert S.consensus_batch(lab, lens, nw, threads=1) == [c["seq"]]
                n_ok += 1
    assert n_ok >= 15, (n_ok, n_err)


def test_native_stitch_equals_difflib_randomised():
    """against the Python mirror (which calls difflib itself) over noisy overlapping fragments of a common sequence and
    unrelated random ones: fragment lengths around difflib's autojunk threshold (199 / 200 / 201), skewed base
    compositions (elements that stay below the "popular" count), empty fragments, single fragments, fragments past the
    reference's 1000-column capacity rule (IndexError on both sides); several threads."""
    from radian_amd import sequence_assembly as S
    rng = np.random.default_rng(7)

    def noisy_read(nfrag, L, overlap, err, p):
        truth = rng.choice(4, size=int(L * (1 + (nfrag - 1) * (1 - overlap))) + 10, p=p)
        out, pos = [], 0
        for _ in range(nfrag):
            n = min(1024, max(0, int(L + rng.integers(-L // 5 - 1, L // 5 + 1))))
            f = truth[pos: pos + n].copy()
            m = rng.random(f.size) < err
            f[m] = rng.integers(0, 4, size=int(m.sum()))
            out.append(f[rng.random(f.size) > err / 2].astype(np.uint8))
            pos += int(L * (1 - overlap))
        return out
    reads = []
    for L in (0, 1, 3, 10, 50, 150, 199, 200, 201, 260, 400, 700, 1020):
        for nfrag in (1, 2, 3, 8):
            for p in ([0.25] * 4, [0.7, 0.2, 0.08, 0.02], [0.97, 0.01, 0.01, 0.01]):
                for _ in range(3):
                    reads.append(noisy_read(nfrag, L, float(rng.choice([0.0, 0.5, 0.9])), float(rng.choice([0.0, 0.05, 0.3])), p))
    for _ in range(200):
        reads.append([rng.integers(0, 4, size=int(rng.integers(0, 300))).astype(np.uint8) for _ in range(int(rng.integers(1, 6)))])
    exp = []
    for fr in reads:
        try:
            exp.append(S.consensus_sequence([S.labels_to_str(f) for f in fr]))
        except IndexError:
            exp.append(IndexError)
    assert exp.count(IndexError) >= 5 and sum(isinstance(e, str) and len(e) > 200 for e in exp) >= 50
    ok = [i for i, e in enumerate(exp) if e is not IndexError]
    lab, lens, nw = _label_matrix([reads[i] for i in ok], 1024)
    got = S.consensus_batch(lab, lens, nw, threads=4)
    bad = [i for i, g in zip(ok, got) if g != exp[i]]
    assert not bad, (bad[:5], [len(f) for f in reads[bad[0]]])
    for i, e in enumerate(exp):
        if e is IndexError:
            lab, lens, nw = _label_matrix([reads[i]], 1024)
            with pytest.raises(IndexError):
                S.consensus_batch(lab, lens, nw, threads=1)


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_default_artifact_route_host_half(tmp_path, monkeypatch):
    """basecall.py:28-30,48-62 without a GPU: the parser's DEFAULT paths (models/sig2seq.h5, models/sig2seq.yaml,
    models/rnamodel_12mer_pc.json, relative to the working directory) through load_dilations (utilities.py:16-18 +
    sig2seq.yaml:34-49), the Keras-h5 converter (model.py:42-45) and the LM loader -- including nb_stacks > 1 and the
    configurations the backend refuses."""
    import json
    import yaml
    from radian_amd import basecall, h5weights, weights
    models = tmp_path / "models"
    models.mkdir()
    monkeypatch.chdir(tmp_path)

    def write_cfg(**over):
        tcn = {"nb_filters": 256, "kernel_size": 3, "nb_stacks": 1, "dilations": [1, 2, 4, 8, 16, 32], "padding": "causal",
               "use_skip_connections": False, "dropout_rate": 0.0, "return_sequences": True, "activation": "relu",
               "kernel_initializer": "he_normal", "use_batch_norm": False}
        model = {"relu_units": 128, "softmax_units": 5, "timesteps": 1024}
        for k, v in over.items():
            (model if k in model else tcn)[k] = v
        model["tcn"] = tcn
        (models / "sig2seq.yaml").write_text(yaml.safe_dump({"train": {"batch_size": 32}, "model": model}))

    args = basecall.build_parser().parse_args(["in", "out", "--context-len", "2"])
    assert (args.sig_model, args.sig_config, args.rna_model) == ("models/sig2seq.h5", "models/sig2seq.yaml", "models/rnamodel_12mer_pc.json")
    write_cfg()
    assert basecall.load_dilations(args.sig_config) == (1, 2, 4, 8, 16, 32)
    write_cfg(nb_stacks=2, dilations=[1, 2, 4])
    assert basecall.load_dilations(args.sig_config) == (1, 2, 4, 1, 2, 4)
    write_cfg(nb_stacks=3, dilations=[1, 8])
    assert basecall.load_dilations(args.sig_config) == (1, 8, 1, 8, 1, 8)
    for over, msg in (({"use_batch_norm": True}, "does not implement"), ({"use_skip_connections": True}, "does not implement"),
                      ({"dropout_rate": 0.05}, "does not implement"), ({"padding": "same"}, "does not implement"),
                      ({"activation": "tanh"}, "does not implement"), ({"nb_filters": 64}, "geometry"), ({"kernel_size": 5}, "geometry"),
                      ({"relu_units": 64}, "geometry"), ({"softmax_units": 4}, "geometry")):
        write_cfg(**over)
        with pytest.raises(ValueError, match=msg):
            basecall.load_dilations(args.sig_config)
    assert basecall.load_dilations("none") == weights.DEFAULT_DILATIONS     # no file: the constants of sig2seq.yaml

    # the whole host half at the default paths: yaml (two stacks) + matching .h5 + LM JSON
    write_cfg(nb_stacks=2, dilations=[1, 2, 4])
    dil = (1, 2, 4, 1, 2, 4)
    w = weights.synthetic_weights(seed=9, dilations=dil)
    h5weights.write_keras_weights(str(models / "sig2seq.h5"), w, dilations=dil)
    rng = np.random.default_rng(0)
    lm = {a + b: [float(x) for x in rng.dirichlet([1.0] * 4)] for a in "ACGT" for b in "ACGT"}
    (models / "rnamodel_12mer_pc.json").write_text(json.dumps(lm))
    art = basecall.load_artifacts(args)
    assert art["dilations"] == dil and np.array_equal(art["weights"], w) and art["lm_k"] == 2
    assert np.array_equal(art["lm_table"][4 * 1 + 2], lm["CG"])
    # (a weights file does not record dilations: the same six-block file loads under the default config too)
    write_cfg()
    assert np.array_equal(basecall.load_artifacts(args)["weights"], w)
    # the config says four blocks, the file holds six: refused by the tensor count before any GPU is touched
    write_cfg(dilations=[1, 2, 4, 8])
    with pytest.raises(ValueError, match="weight tensors"):
        basecall.load_artifacts(args)
    # the reference's default --context-len (11) against this 2-label model: decode.py:83 raises KeyError lazily, on the first read
    # whose search keeps a labeling of 11 labels -- the artefacts ask for the model in which every 11-label context is absent
    write_cfg(nb_stacks=2, dilations=[1, 2, 4])
    art = basecall.load_artifacts(basecall.build_parser().parse_args(["in", "out"]))
    assert art["lm_absent"] == 11 and art["lm_k"] == 11 and art["lm_table"] is None
    basecall.save_artifacts(art, str(tmp_path))
    back = basecall.load_artifacts(None, cache_dir=str(tmp_path))
    assert back["lm_absent"] == 11 and back["lm_table"] is None and np.array_equal(back["weights"], w)
    with pytest.raises(KeyError):      # no dense image beyond 13 labels: there the KeyError comes at load
        basecall.load_artifacts(basecall.build_parser().parse_args(["in", "out", "--context-len", "14"]))
    # start-up failures like the reference's: a --sig-config that does not exist (utilities.py:16-18 opens it), an --rna-model that
    # does not exist in EITHER decode type (basecall.py:48-50 opens it before the decode type matters)
    with pytest.raises(FileNotFoundError):
        basecall.load_artifacts(basecall.build_parser().parse_args(["in", "out", "--context-len", "2", "--sig-config", "models/nope.yaml"]))
    for mode in ("global", "chunk"):
        with pytest.raises(FileNotFoundError) as ei:
            basecall.load_artifacts(basecall.build_parser().parse_args(["in", "out", "--decode-type", mode, "--rna-model", "models/nope.json"]))
        assert ei.value.filename == "models/nope.json"
    # ... and the explicit ways of running without either file stay: --sig-config none, --rna-model None
    art = basecall.load_artifacts(basecall.build_parser().parse_args(["in", "out", "--decode-type", "chunk", "--rna-model", "None", "--sig-config", "none",
                                                                       "--sig-model", "synthetic:3"]))
    assert art["lm_table"] is None and art["dilations"] == weights.DEFAULT_DILATIONS


def test_driver_raises_reference_keyerror_at_the_read_that_reaches_a_missing_context(oracle, capsys):
    """A sparse RNA model (basecall.py:48-57 builds a dict of whatever the JSON holds) works in the reference until a read's beam
    search looks an absent context up (decode.py:83: KeyError, uncaught in basecall.py:70-141, so the run dies there with the
    earlier reads written).  Same here, whatever the device batch size: reads before the failing one come out, then KeyError."""
    from radian_amd import basecall, weights
    from _oracle_backend import OracleBackend
    from _reads import golden_reads
    be = OracleBackend()
    be.load_weights(weights.synthetic_weights(seed=5, dilations=(1, 2)), (1, 2))
    dense = np.random.default_rng(9).dirichlet([0.3] * 4, size=16)
    base = ["a", "b", "--chunk-len", "128", "--step-size", "64", "--beam-width", "3", "--decode-type", "global", "--context-len", "2",
            "--sig-threshold", "0.0", "--rna-threshold", "9.0"]
    args = basecall.build_parser().parse_args(base + ["--gpu-batch-windows", "40"])
    args._lm_loaded = True
    be.load_lm(dense, 2)
    full = basecall.run(args, be, reads=golden_reads(700), writer=None)
    assert len(full) >= 4
    found = None
    for c in range(16):
        t = dense.copy()
        t[c] = np.nan
        be.load_lm(t, 2)
        labs, _ = be.basecall_raw_global([r.get_raw_data() for r in golden_reads(700)], 4, 128, 64, 3, True, 0.0, 9.0)
        bad = [i for i, l in enumerate(labs) if l is None]
        if bad and 1 <= bad[0] and len(bad) < len(labs):
            found = (c, bad)
            break
    assert found is not None, "no single absent context splits the five reads"
    c, bad = found
    for gbw in ("1", "40"):
        a2 = basecall.build_parser().parse_args(base + ["--gpu-batch-windows", gbw])
        a2._lm_loaded = True
        seen = []
        with pytest.raises(KeyError, match="decode.py:83"):
            basecall.run(a2, be, reads=golden_reads(700), writer=None, on_result=lambda idx, rid, seq: seen.append((idx, rid, seq)))
        # (reads that do not reach the context decode exactly as with the dense model: the absent row is never read for them)
        assert seen == full[: bad[0]]
    capsys.readouterr()


def test_read_ahead_preserves_order_bounds_memory_and_propagates_errors():
    """basecall._read_ahead (round 4): the reader thread hands reads over in order, never runs more than max_reads / max_samples
    ahead of the consumer, re-raises the producer's exception at the position where it happened, and stops (closing the source
    generator on its own thread) when the consumer leaves early."""
    import threading
    import time
    from radian_amd import basecall

    class R:
        def __init__(self, i, n, boom=False):
            self.read_id, self.n, self.boom = f"r{i}", n, boom

        def get_raw_data(self):
            if self.boom:
                raise OSError("disk on fire")
            return np.full(self.n, 7, dtype=np.int16)

    state = {"made": 0, "closed_on": None}

    def source(n, boom_at=None):
        try:
            for i in range(n):
                state["made"] = i + 1
                yield i, R(i, 10 + i % 5, boom=(i == boom_at))
        finally:
            state["closed_on"] = threading.current_thread().name

    got = list(basecall._read_ahead(source(1000), max_reads=64, max_samples=1 << 20, block=16))
    assert [k for k, _, _ in got] == list(range(1000)) and [rid for _, rid, _ in got] == [f"r{i}" for i in range(1000)]
    assert all(raw.shape[0] == 10 + i % 5 for i, (_, _, raw) in enumerate(got))
    assert state["closed_on"] == "radian-read-ahead"
    # bounded run-ahead: a slow consumer never finds the producer more than max_reads (+ one block in the making) ahead
    state["made"] = 0
    it = basecall._read_ahead(source(2000), max_reads=64, max_samples=1 << 20, block=16)
    for n_taken, _ in enumerate(it, 1):
        if n_taken % 100 == 0:
            time.sleep(0.02)
            assert state["made"] - n_taken <= 64 + 16 + 16, (state["made"], n_taken)
        if n_taken == 900:
            break
    it.close()                                   # the consumer leaves early: the producer stops and closes its generator
    assert state["closed_on"] == "radian-read-ahead" and state["made"] < 2000
    # the producer's error arrives after the reads before it
    seen = []
    with pytest.raises(OSError, match="disk on fire"):
        for k, _, _ in basecall._read_ahead(source(300, boom_at=137), block=16):
            seen.append(k)
    assert seen == list(range(137))   # (every read before the failing one, the partly filled block included: basecall.py:70-141)


def test_artifact_cache_roundtrip_for_multi_gpu_ranks(tmp_path):
    """basecall.save_artifacts / load_artifacts(cache_dir): what the multi-GPU launcher parsed once is what the broadcasting rank
    loads -- weights, dilations, a SPARSE RNA table (rows of NaN survive), the hashed-context order."""
    from radian_amd import basecall, weights
    w = weights.synthetic_weights(seed=4, dilations=(1, 2, 4))
    table = np.random.default_rng(2).dirichlet([0.5] * 4, size=64)
    table[[3, 40]] = np.nan
    for art in ({"dilations": (1, 2, 4), "weights": w, "lm_table": table, "lm_k": 3},
                {"dilations": (1, 2, 4), "weights": w, "lm_table": None, "lm_k": 0},
                {"dilations": (1, 2, 4), "weights": w, "lm_table": table, "lm_k": 40, "lm_hashed_order": 3}):
        d = tmp_path / f"c{art['lm_k']}"
        d.mkdir()
        basecall.save_artifacts(art, str(d))
        got = basecall.load_artifacts(None, cache_dir=str(d))
        assert got["dilations"] == (1, 2, 4) and np.array_equal(got["weights"], w) and got["lm_k"] == art["lm_k"]
        assert got.get("lm_hashed_order") == art.get("lm_hashed_order")
        if art["lm_table"] is None:
            assert got["lm_table"] is None
        else:
            assert np.array_equal(got["lm_table"], table, equal_nan=True) and np.isnan(got["lm_table"][3]).all()


def test_lm_json_native_reader_equals_the_standard_parser(tmp_path):
    """lm.load_json's first route (rd_lm_json_probe / rd_lm_json_fill: the model file scanned straight into the table) gives the
    table json.load + table_from_dict gives, bit for bit -- dense and sparse models, any whitespace, integer / exponent / subnormal
    numbers, a repeated key (last value wins, as in a dict) -- and declines (None -> the standard parser decides and raises the
    reference's errors) on everything that is not an object of equal-length ACGT keys with four finite JSON numbers each."""
    import json
    from radian_amd import lm
    rng = np.random.default_rng(5)
    for k, frac, indent, seps in [(1, 1.0, None, None), (3, 1.0, None, None), (5, 0.6, 2, None), (4, 1.0, None, (",", ":")), (6, 0.9, 1, None)]:
        keys = ["".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k)) for i in range(4 ** k)]
        rng.shuffle(keys)
        keys = keys[: max(1, int(len(keys) * frac))]
        d = {c: [float(x) for x in rng.dirichlet([0.3] * 4)] for c in keys}
        d[keys[0]] = [0, 1, 1e-310, 2.5E-7]
        p = tmp_path / f"m{k}.json"
        p.write_text(json.dumps(d, indent=indent, separators=seps))
        nat, std = lm._load_json_native(str(p)), lm.load_json(str(p), native=False)
        assert nat is not None and nat[1] == std[1] == k and nat[0].tobytes() == std[0].tobytes(), (k, frac)
        got = lm.load_json(str(p))
        assert got[1] == k and got[0].tobytes() == std[0].tobytes()
        assert lm.n_missing(got[0]) == 4 ** k - len(keys)
    p = tmp_path / "dup.json"
    p.write_text('{"A":[1,0,0,0],"C":[0,1,0,0],"A":[0.5,0.5,0,-0.0],"G":[0,0,1,0],"T":[0,0,0,1]}')
    nat, std = lm._load_json_native(str(p)), lm.load_json(str(p), native=False)
    assert nat[0].tobytes() == std[0].tobytes() and nat[0][0, 0] == 0.5
    declined = ['{"ACGU":[0.1,0.2,0.3,0.4]}', '{"AC":[0.1,0.2,0.3,0.4],"A":[1,0,0,0]}', '{"AC":[0.1,0.2,0.3]}', '{"AC":[0.1,0.2,0.3,NaN]}',
                '{"AC":[0.1,0.2,0.3,0.4]} x', '[]', '{}', '', '{"AC":[0.1,0.2,0.3,0.4,0.5]}', '{"A\\u0043":[0.1,0.2,0.3,0.4]}', '{"AC":[1e999,0,0,0]}',
                '{"AC":[.5,0,0,0]}', '{"AC":[01,0,0,0]}', '{"AC":[1,0,0,0],}', '{"AC":{"x":1}}', '{"AC":[1,0,0,0]', '{"ac":[1,0,0,0]}']
    for text in declined:
        p = tmp_path / "d.json"
        p.write_text(text)
        assert lm._load_json_native(str(p)) is None, text
    # ... and the standard route then speaks: the reference's ValueError for a non-ACGT key (bases.index, basecall.py:56)
    p.write_text('{"ACGU":[0.1,0.2,0.3,0.4]}')
    with pytest.raises(ValueError, match="is not in list"):
        lm.load_json(str(p))


def test_lm_json_reader_under_address_sanitizer(tmp_path):
    """csrc/lmjson.hip is host code that walks untrusted text: compiled for the CPU with -fsanitize=address,undefined and fed 60 000
    valid, mutated, truncated and random texts in exact-size heap buffers (no terminator), it must never read past an end."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "asan_lmjson"
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-x", "c++", os.path.join(ROOT, "radian_amd", "csrc", "lmjson.hip"),
                        os.path.join(ROOT, "tests", "asan_lmjson.cpp"), "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if r.returncode != 0 and b"sanitize" in r.stderr and b"cannot find" in r.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    r = subprocess.run([str(exe), "60000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"no sanitizer report" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-3000:])
    accepted = int(r.stdout.split()[0])
    assert 5000 < accepted < 40000          # the harness feeds both kinds


def test_native_stitch_under_address_sanitizer(tmp_path):
    """csrc/stitch.hip (simple_assembly + difflib restated, host threads) compiled for the CPU with -fsanitize=address,undefined: 20 000
    random batches -- empty / repeated / shifted fragments, fragments of 200+ labels (autojunk), exact-size buffers, 1-3 threads."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "asan_stitch"
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-x", "c++", os.path.join(ROOT, "radian_amd", "csrc", "stitch.hip"),
                        os.path.join(ROOT, "tests", "asan_stitch.cpp"), "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if r.returncode != 0 and b"sanitize" in r.stderr and b"cannot find" in r.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    r = subprocess.run([str(exe), "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"no sanitizer report" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-3000:])


def test_tile_planner_properties_under_address_sanitizer(tmp_path):
    """csrc/plan.hip -- the host code whose tile descriptors every forward kernel indexes HBM with -- compiled for the CPU with
    -fsanitize=address,undefined and checked on 2500 random (model, read set, chunk, step) geometries: every sub-tile inside its segment,
    every write inside the activation tensor and no row written twice within a layer, every sample / stream row a sub-tile reads existing,
    every decoded sequence's rows inside the tensor (tests/asan_plan.cpp).  A violated property would be an out-of-bounds access on the GPU.
    Round 6: the same harness checks rd_plan_trie_runs (common.h) on 10 000 random launches -- the runs tile the launch in order, node offsets
    restart per run, every run of more than one sequence fits the workspace budget, no run ends early."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "asan_plan"
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-x", "c++", os.path.join(ROOT, "radian_amd", "csrc", "plan.hip"),
                        os.path.join(ROOT, "tests", "asan_plan.cpp"), "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if r.returncode != 0 and b"sanitize" in r.stderr and b"cannot find" in r.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    r = subprocess.run([str(exe), "2500"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"every property holds" in r.stdout, (r.stdout.decode()[-800:], r.stderr.decode()[-3000:])


def test_host_budget_placement_rule(tmp_path):
    """radian_amd/hostbudget.py: every local rank gets a disjoint, contiguous share of the usable cores -- the cores of
    its GPU's NUMA node when the (fake) sysfs tells -- and sizes its threads from the share: eight chunk-mode ranks on 64 cores start 8 x 6
    stitch threads, not 8 x 16."""
    from radian_amd import hostbudget as hb
    assert hb.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and hb.parse_cpulist("") == []
    # a fake two-socket node: 64 cores (0-31 | 32-63), eight GPUs, four per socket; one CPU-only KFD node in front
    sysfs = tmp_path / "sys"
    for n, cl in ((0, "0-31"), (1, "32-63")):
        d = sysfs / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    top = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (top / "0").mkdir(parents=True)
    (top / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for g in range(8):
        (top / str(g + 1)).mkdir()
        bus = 0x10 + g * 0x10
        (top / str(g + 1) / "properties").write_text(f"simd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        pd = sysfs / "bus" / "pci" / "devices" / f"0000:{bus:02x}:00.0"
        pd.mkdir(parents=True)
        (pd / "numa_node").write_text(f"{0 if g < 4 else 1}\n")
    assert hb.gpu_numa_nodes(str(sysfs)) == [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [hb.plan(r, 8, usable=range(64), gpu_nodes=hb.gpu_numa_nodes(str(sysfs)), sysfs=str(sysfs)) for r in range(8)]
    assert all(p["how"] == "numa" and len(p["cpus"]) == 8 for p in plans)
    assert [p["cpus"][0] for p in plans] == [0, 8, 16, 24, 32, 40, 48, 56] and [p["numa_node"] for p in plans] == [0] * 4 + [1] * 4
    assert sorted(c for p in plans for c in p["cpus"]) == list(range(64))                      # disjoint, nothing left over
    assert hb.threads_for(8, "chunk")["stitch_threads"] == 6 and hb.threads_for(8, "global")["stitch_threads"] == 0
    assert sum(hb.threads_for(len(p["cpus"]), "chunk")["total"] for p in plans) <= 2 * 64      # <= 10 threads per rank, most of them waiting
    # a cgroup that only allows 20 cores of socket 0 and 4 of socket 1: still NUMA-local where every rank gets a core
    usable = list(range(4, 24)) + [40, 41, 42, 43]
    plans = [hb.plan(r, 8, usable=usable, gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], sysfs=str(sysfs)) for r in range(8)]
    assert all(p["how"] == "numa" for p in plans) and [len(p["cpus"]) for p in plans] == [5, 5, 5, 5, 1, 1, 1, 1]
    assert sorted(c for p in plans for c in p["cpus"]) == usable
    # fewer usable cores on a node than ranks on it: the even split for EVERY rank (mixed rules could overlap)
    plans = [hb.plan(r, 8, usable=list(range(0, 32)) + [40, 41], gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], sysfs=str(sysfs)) for r in range(8)]
    assert all(p["how"] == "even" for p in plans) and sorted(c for p in plans for c in p["cpus"]) == list(range(32)) + [40, 41]
    # no topology: even split; one rank: everything; the container this test runs in, whatever it is
    plans = [hb.plan(r, 3, usable=range(8), gpu_nodes=None) for r in range(3)]
    assert [p["cpus"] for p in plans] == [[0, 1], [2, 3, 4], [5, 6, 7]] and plans[0]["how"] == "even"
    assert hb.plan(0, 1, usable=range(8))["how"] == "all"
    here = [hb.plan(r, 2) for r in range(2)]
    assert not set(here[0]["cpus"]) & set(here[1]["cpus"]) and here[0]["cpus"] and here[1]["cpus"]
    # a launcher that narrows every process to its own device (HIP_VISIBLE_DEVICES=<r> per rank) leaves ONE visible GPU and
    # eight local ranks -- nobody can see the peers' GPUs: the even split, not 1/8 of this GPU's node for everyone
    plans = [hb.plan(r, 8, usable=range(64), gpu_nodes=[0 if r < 4 else 1], sysfs=str(sysfs)) for r in range(8)]
    assert all(p["how"] == "even" and len(p["cpus"]) == 8 for p in plans) and sorted(c for p in plans for c in p["cpus"]) == list(range(64))
    # ... while several ranks put on ONE device by an explicit list (RD_CLI_DEVICE / RD_BENCH_DEVICE rehearsals) share that GPU's node
    plans = [hb.plan(r, 4, usable=range(64), gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], devices=[5, 5, 5, 5], sysfs=str(sysfs)) for r in range(4)]
    assert all(p["how"] == "numa" and p["numa_node"] == 1 and len(p["cpus"]) == 8 for p in plans)
    assert sorted(c for p in plans for c in p["cpus"]) == list(range(32, 64))
    # HIP_VISIBLE_DEVICES re-orders the devices a rank sees
    import os
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    os.environ["HIP_VISIBLE_DEVICES"] = "4,5"
    try:
        assert hb._visible_order(8) == [4, 5]
    finally:
        if old is None:
            del os.environ["HIP_VISIBLE_DEVICES"]
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = old


def test_apply_binds_a_child_rank_to_its_slice():
    """hostbudget.apply in a fresh process (the worker's first act): rank 1 of 2 ends up bound to the upper half of what it could use, and a
    thread-count default that reads sched_getaffinity (sequence_assembly.consensus_batch) follows the slice."""
    import subprocess
    import sys
    code = ("import os, json; from radian_amd import hostbudget as hb; before = sorted(os.sched_getaffinity(0)); p = hb.apply(1, 2); "
            "print(json.dumps({'before': before, 'after': sorted(os.sched_getaffinity(0)), 'plan': p['cpus'], 'bound': p['bound']}))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    import json
    d = json.loads(out.stdout.strip().splitlines()[-1])
    if len(d["before"]) < 2:
        pytest.skip("one usable core")
    assert d["bound"] and d["after"] == d["plan"] == d["before"][len(d["before"]) // 2:]
    code = code.replace("hb.apply(1, 2)", "hb.apply(1, 2, 'none')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert not d["bound"] and d["after"] == d["before"]
    # threads that exist BEFORE apply (a BLAS / OpenMP pool started at import) are bound too, not only the calling thread
    code = ("import os, json, threading; from radian_amd import hostbudget as hb; ev = threading.Event(); "
            "ts = [threading.Thread(target=ev.wait) for _ in range(3)]; [t.start() for t in ts]; p = hb.apply(0, 2); "
            "masks = [sorted(os.sched_getaffinity(int(t))) for t in os.listdir('/proc/self/task')]; ev.set(); [t.join() for t in ts]; "
            "print(json.dumps({'plan': p['cpus'], 'masks': masks, 'blas': os.environ.get('OPENBLAS_NUM_THREADS')}))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(d["masks"]) >= 4 and all(m == d["plan"] for m in d["masks"]) and d["blas"] == "1"


def _write_single_read_fast5(path, read_id, sig, group="Read_17", id_kind="nullterm", filters=()):
    from radian_amd import h5
    with h5.File(path, "w") as f:
        f.create_group("/Raw")
        f.create_group("/Raw/Reads")
        f.create_group(f"/Raw/Reads/{group}")
        f.write(f"/Raw/Reads/{group}/Signal", np.ascontiguousarray(sig, dtype=np.int16), chunks=(max(1, min(len(sig), 1000)),), filters=filters)
        if read_id is not None:
            f.set_attr_str(f"/Raw/Reads/{group}", "read_id", read_id, kind=id_kind)


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_native_fast5_reader_matches_libhdf5(tmp_path, golden_dir, monkeypatch):
    """csrc/fast5.hip (rd_fast5_*: the classic HDF5 layout walked over a mapping, reads copied in batches) against the ctypes -> libhdf5
    reader: the five signals of the reference's data/reads.fast5 (fixture: tests/golden/reads_fast5_signals.npz, re-written through
    libhdf5), a 3 000-read multi-read file with ragged lengths incl. empty and one-sample reads, single-read files with and without a
    read_id attribute, claims [lo, hi) as the work queue makes them; and the hand-over to libhdf5 for what it has no verdict on (an int32
    signal, a file that is no HDF5 at all raises what libhdf5 raises)."""
    import json
    from radian_amd import fast5, h5
    ids = json.load(open(os.path.join(golden_dir, "reads_fast5_ids.json")))["read_ids"]
    sig = np.load(os.path.join(golden_dir, "reads_fast5_signals.npz"))
    p1 = str(tmp_path / "golden.fast5")
    fast5.write_multi_fast5(p1, {r: sig[r] for r in ids})
    rng = np.random.default_rng(8)
    many = {}
    for i in range(3000):
        n = int(rng.choice([0, 1, 2, 63, 4095, 4096, 4097, 9000])) if i % 11 == 0 else int(rng.integers(200, 6000))
        many[f"{rng.integers(0, 1 << 62):016x}-{i}"] = np.round(rng.normal(500, 80, size=n)).astype(np.int16)
    p2 = str(tmp_path / "many.fast5")
    fast5.write_multi_fast5(p2, many)
    p3, p4 = str(tmp_path / "single.fast5"), str(tmp_path / "single_noid.fast5")
    _write_single_read_fast5(p3, "4f1b2c3d-aaaa-bbbb-cccc-0123456789ab", sig[ids[1]])
    _write_single_read_fast5(p4, None, sig[ids[2]][:777], group="Read_5")

    def both(path):
        monkeypatch.setenv("RADIAN_FAST5_NATIVE", "0")
        a = [(r.read_id, np.asarray(r.get_raw_data())) for r in fast5.iter_reads(path)]
        monkeypatch.setenv("RADIAN_FAST5_NATIVE", "1")
        assert fast5._open_native(path) is not None, "the native reader should recognise " + path
        b = [(r.read_id, np.asarray(r.get_raw_data())) for r in fast5.iter_reads(path)]
        assert len(a) == len(b)
        for (ia, sa), (ib, sb) in zip(a, b):
            assert ia == ib and sb.dtype == np.int16 and sa.dtype == sb.dtype and np.array_equal(sa, sb)
        return b

    got = both(p1)
    assert [g[0] for g in got] == sorted(ids) and all(np.array_equal(g[1], sig[g[0]]) for g in got)
    got = both(p2)
    assert len(got) == 3000 and [g[0] for g in got] == sorted(many)
    assert both(p3)[0][0] == "4f1b2c3d-aaaa-bbbb-cccc-0123456789ab"
    assert both(p4)[0][0] == "Read_5"
    # claims as the multi-GPU work queue makes them, across the native reader's block boundary
    src = fast5.Fast5Source(p2)
    assert src.n_reads() == 3000
    names = sorted(many)
    for lo, hi in ((0, 1), (250, 530), (2990, 3100), (17, 17)):
        out = list(src.reads(lo, hi))
        assert [i for i, _ in out] == list(range(lo, min(hi, 3000)))
        assert all(r.read_id == names[i] and np.array_equal(r.get_raw_data(), many[names[i]]) for i, r in out)
    src.close()
    assert src.n_reads() == 3000       # (re-opened after a close, as the queue does)
    src.close()
    # no verdict -> libhdf5: a signal stored as int32 (the native reader only takes int16)
    p5 = str(tmp_path / "int32.fast5")
    with h5.File(p5, "w") as f:
        f.create_group("/read_x/Raw")
        f.write("/read_x/Raw/Signal", np.arange(50, dtype=np.int32))
    assert fast5._open_native(p5) is not None          # the index is readable ...
    r = [(x.read_id, x.get_raw_data()) for x in fast5.iter_reads(p5)]     # ... the read is not: libhdf5 takes over from that read on
    assert len(r) == 1 and r[0][0] == "x" and np.array_equal(r[0][1], np.arange(50))
    p6 = str(tmp_path / "junk.fast5")
    open(p6, "wb").write(b"this is not an HDF5 file" * 100)
    assert fast5._open_native(p6) is None
    with pytest.raises(h5.H5Error):
        list(fast5.iter_reads(p6))


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_native_fast5_reader_single_read_files_as_h5py_writes_them(tmp_path):
    """Single-read files as ont_fast5_api's multi_to_single_fast5 leaves them: h5py stores a Python str attribute as a VARIABLE-LENGTH string
    (its bytes live in the file's global heap), and the signal gzip-compressed.  csrc/fast5.hip alone: the id and the samples libhdf5 reads."""
    from radian_amd import fast5
    rng = np.random.default_rng(12)
    for i, (rid, kind, filters) in enumerate([("0a1b2c3d-0000-1111-2222-333344445555", "vlen", (("deflate", 1),)),
                                              ("f" * 36, "vlen", ()), ("7e57-id", "nullpad", ("shuffle", ("deflate", 3))), ("", "vlen", ())]):
        sig = np.round(rng.normal(500, 80, size=int(rng.integers(1, 5000)))).astype(np.int16)
        p = str(tmp_path / f"s{i}.fast5")
        _write_single_read_fast5(p, rid, sig, group=f"Read_{40 + i}", id_kind=kind, filters=filters)
        nf = fast5.NativeFile(p)
        names, samples, off = nf.batch(0, nf.n)
        nf.close()
        ref = _iter_libhdf5(p)
        assert names == [ref[0][0]] and np.array_equal(samples, ref[0][1]) and np.array_equal(samples, sig), (i, names, ref[0][0])
        assert names == [rid], (i, names)


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_native_fast5_reader_gives_no_verdict_on_what_it_does_not_model(tmp_path):
    """ADVICE r5 (csrc/fast5.hip): cases the native reader used to answer silently now end without a verdict and libhdf5 reads the file --
    a Signal with a fill value other than zero (libhdf5 returns that value for storage that was never written: contiguous and chunked),
    a space-padded fixed-length read_id.  A zero fill value and NUL-padded ids stay native.  A failed resolve leaves no state behind:
    asking the same handle again gives the same answer."""
    from radian_amd import fast5, h5, _lib
    import ctypes
    L = _lib.load()

    def native(path):
        h = ctypes.c_void_p()
        assert L.rd_fast5_open(path.encode(), ctypes.byref(h)) == 0
        n = ctypes.c_int64()
        assert L.rd_fast5_count(h, ctypes.byref(n)) == 0
        lens = np.zeros(n.value, dtype=np.int64)
        rcs = [L.rd_fast5_lengths(h, 0, n.value, lens.ctypes.data_as(ctypes.c_void_p)) for _ in range(2)]   # (twice: a retry on the same entry)
        msg = L.rd_last_error().decode()
        out = None
        if rcs[0] == 0:
            samples = np.zeros(int(lens.sum()) + 1, dtype=np.int16)
            off = np.zeros(n.value + 1, dtype=np.int64)
            ids = ctypes.create_string_buffer(64 * n.value)
            rc = L.rd_fast5_read_batch(h, 0, n.value, samples.ctypes.data_as(ctypes.c_void_p), samples.size, off.ctypes.data_as(ctypes.c_void_p), ids, 64)
            msg = L.rd_last_error().decode()
            out = (rc, samples[: int(lens.sum())], [ids.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode() for i in range(n.value)])
        L.rd_fast5_close(h)
        return rcs, msg, out

    sig = np.arange(1, 2501, dtype=np.int16)
    cases = []
    for name, kw in (("fill7_contig_unwritten", dict(fill=7, store=False)), ("fill7_chunked_unwritten", dict(fill=7, store=False, chunks=(1000,))),
                     ("fill7_chunked_written", dict(fill=7, chunks=(1000,))), ("fill0_contig_unwritten", dict(fill=0, store=False)),
                     ("fill0_chunked_written", dict(fill=0, chunks=(1000,), filters=(("deflate", 1),)))):
        p = str(tmp_path / (name + ".fast5"))
        with h5.File(p, "w") as f:
            f.create_group("/read_a/Raw")
            f.write("/read_a/Raw/Signal", sig, **kw)
        cases.append((name, p, kw))
    for name, p, kw in cases:
        rcs, msg, out = native(p)
        ref = _iter_libhdf5(p)
        exp = np.full(2500, kw["fill"], dtype=np.int16) if not kw.get("store", True) else sig
        assert ref[0][0] == "a" and np.array_equal(ref[0][1], exp), name
        if kw["fill"] == 7:
            assert rcs == [-6, -6] and "fill value" in msg, (name, rcs, msg)            # RD_ERR_FORMAT, both times
        else:
            assert rcs == [0, 0] and out[0] == 0 and np.array_equal(out[1], exp), (name, rcs, msg)
        got = [(r.read_id, np.asarray(r.get_raw_data())) for r in fast5.iter_reads(p)]        # the product's route: native first, libhdf5 takes over
        assert got[0][0] == "a" and np.array_equal(got[0][1], exp), name
    # single-read files: a space-padded read_id has no native verdict; libhdf5 (like h5py) hands the id without its blanks
    for kind, native_ok in (("spacepad", False), ("nullpad", True), ("nullterm", True)):
        p = str(tmp_path / f"id_{kind}.fast5")
        _write_single_read_fast5(p, "0a1b-id", sig, id_kind=kind)
        rcs, msg, out = native(p)
        if native_ok:
            assert rcs == [0, 0] and out[2] == ["0a1b-id"] and np.array_equal(out[1], sig), (kind, rcs, msg)
        else:
            assert rcs == [-6, -6] and "space-padded" in msg, (kind, rcs, msg)
        got = [(r.read_id, np.asarray(r.get_raw_data())) for r in fast5.iter_reads(p)]
        assert got == [] or (got[0][0] == "0a1b-id" and np.array_equal(got[0][1], sig)), (kind, got[0][0])
        assert len(got) == 1


def _iter_libhdf5(path):
    from radian_amd import fast5
    old = os.environ.get("RADIAN_FAST5_NATIVE")
    os.environ["RADIAN_FAST5_NATIVE"] = "0"
    try:
        return [(r.read_id, np.asarray(r.get_raw_data())) for r in fast5.iter_reads(path)]      # (read while the file is open)
    finally:
        if old is None:
            del os.environ["RADIAN_FAST5_NATIVE"]
        else:
            os.environ["RADIAN_FAST5_NATIVE"] = old


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_native_fast5_reader_undoes_hdf5_builtin_filters(tmp_path):
    """Signals stored through HDF5's built-in filters -- deflate (the gzip level-1 signals of pre-VBZ MinKNOW files), byte shuffle, Fletcher-32,
    in any order, with chunks shorter and longer than the reads -- written by libhdf5 itself and read back by csrc/fast5.hip ALONE
    (NativeFile.batch: no fall-back behind it): same ids, same samples as libhdf5 reads.  A damaged stream or checksum is no verdict
    (libhdf5 then reports it), and so is a filter the reader does not know."""
    from radian_amd import fast5, h5
    rng = np.random.default_rng(8)
    reads = {f"{i:06d}-{rng.integers(0, 1 << 40):x}": np.round(rng.normal(500, 80, size=int(rng.choice([0, 1, 2, 699, 700, 701, int(rng.integers(1, 12000))])))).astype(np.int16)
             for i in range(60)}
    ids = sorted(reads)
    combos = [((("deflate", 1),), 4096), (("shuffle", ("deflate", 4)), 700), (("fletcher32",), 4096), (("shuffle", ("deflate", 9), "fletcher32"), 777),
              (("fletcher32", ("deflate", 1)), 4096), ((("deflate", 6), "shuffle"), 100), (("shuffle",), 333)]
    for filters, chunk in combos:
        p = str(tmp_path / "f.fast5")
        fast5.write_multi_fast5(p, reads, filters=filters, chunk=chunk)
        nf = fast5.NativeFile(p)
        names, samples, off = nf.batch(0, nf.n)
        nf.close()
        assert names == ids, filters
        assert all(np.array_equal(samples[off[i]:off[i + 1]], reads[r]) for i, r in enumerate(ids)), filters
        with h5.File(p) as f:         # (and libhdf5 agrees that the file holds what was written)
            assert all(np.array_equal(f.read(f"/read_{r}/Raw/Signal"), reads[r]) for r in ids[:5])
    # damage inside the compressed, checksummed chunks: never wrong samples -- no verdict, and through iter_reads libhdf5's error
    p = str(tmp_path / "g.fast5")
    fast5.write_multi_fast5(p, reads, filters=(("deflate", 1), "fletcher32"), chunk=4096)
    img = bytearray(open(p, "rb").read())
    good = bytes(img)
    outcomes = set()
    for trial in range(40):
        img[:] = good
        at = int(rng.integers(len(img) // 3, len(img)))
        img[at] ^= 1 << int(rng.integers(8))
        q = str(tmp_path / "bad.fast5")
        open(q, "wb").write(img)
        try:
            nf = fast5.NativeFile(q)
            names, samples, off = nf.batch(0, nf.n)
            nf.close()
        except fast5.NativeFile.Unrecognised as e:
            outcomes.add("no verdict")
            assert "inflate" in str(e) or "checksum" in str(e) or "chunk" in str(e) or "outside" in str(e) or "header" in str(e) or "B-tree" in str(e), str(e)
            continue
        # the flipped bit lay outside every chunk (padding, free space, an attribute): whatever decoded is what was written
        outcomes.add("intact")
        assert all(np.array_equal(samples[off[i]:off[i + 1]], reads[r]) for i, r in enumerate(names) if r in reads and off[i + 1] - off[i] == len(reads[r]))
    assert "no verdict" in outcomes
    # an unknown filter id in the pipeline message (VBZ's 32020 in place of deflate's 1): no verdict
    img[:] = good
    k = good.find(b"deflate\0")
    assert k > 0
    # (version-1 filter description: id u16 | name length u16 | flags u16 | n values u16 | name)
    assert img[k - 8] == 1 and img[k - 7] == 0
    img[k - 8:k - 6] = (32020).to_bytes(2, "little")
    q = str(tmp_path / "vbz.fast5")
    open(q, "wb").write(img)
    nf = fast5.NativeFile(q)          # (the index is still readable: the first read's signal is not)
    with pytest.raises(fast5.NativeFile.Unrecognised, match="VBZ"):
        nf.batch(0, 1)
    nf.close()


@pytest.mark.skipif(not _have_hdf5(), reason="libhdf5 not available")
def test_native_fast5_reader_under_address_sanitizer(tmp_path, golden_dir):
    """csrc/fast5.hip compiled for the CPU with -fsanitize=address,undefined: valid multi- and single-read files parsed from exact-size heap
    copies, then thousands of mutated / truncated copies (bit flips, 0xff runs, wild 8-byte fields, half of them in the first 16 KiB): any
    answer is fine, a read outside the image or the output block is not."""
    import shutil
    import subprocess
    from radian_amd import fast5
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    rng = np.random.default_rng(3)
    reads = {f"{i:04d}-{rng.integers(0, 1 << 40):x}": np.round(rng.normal(500, 80, size=int(rng.integers(0, 2500)))).astype(np.int16) for i in range(40)}
    p1 = str(tmp_path / "multi.fast5")
    fast5.write_multi_fast5(p1, reads)
    p2 = str(tmp_path / "single.fast5")
    _write_single_read_fast5(p2, "0a1b2c3d-0000-1111-2222-333344445555", np.arange(2300) % 700)
    p4 = str(tmp_path / "single_vlen.fast5")    # (id in the global heap)
    _write_single_read_fast5(p4, "0a1b2c3d-9999-1111-2222-333344445555", np.arange(1200) % 650, id_kind="vlen", filters=(("deflate", 1),))
    p3 = str(tmp_path / "filtered.fast5")     # every chunk through shuffle + deflate + Fletcher-32: the mutations land in compressed streams and checksums too
    fast5.write_multi_fast5(p3, dict(list(reads.items())[:12]), filters=("shuffle", ("deflate", 1), "fletcher32"), chunk=700)
    exe = tmp_path / "asan_fast5"
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-x", "c++", os.path.join(ROOT, "radian_amd", "csrc", "fast5.hip"),
                        os.path.join(ROOT, "tests", "asan_fast5.cpp"), "-lz", "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if r.returncode != 0 and b"sanitize" in r.stderr and b"cannot find" in r.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    r = subprocess.run([str(exe), "2500", p1, p2, p3, p4], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0 and b"no sanitizer report" in r.stdout, (r.stdout.decode()[-800:], r.stderr.decode()[-3000:])
    lines = r.stdout.decode().splitlines()
    assert lines[0].split(": ")[1].startswith("40 reads") and lines[1].split(": ")[1].startswith("1 reads") and lines[2].split(": ")[1].startswith("12 reads")
    assert lines[3].split(": ")[1].startswith("1 reads")
    opened, refused = int(lines[-1].split()[0]), int(lines[-1].split()[2])
    assert opened > 2000 and refused > 300          # the mutations reach both outcomes


def test_driver_loop_forgets_consumed_records(tmp_path):
    """basecall.run with a writer (the single-process command line's route, basecall.main): the reference writes a record and forgets it
    (basecall.py:129); a run over 200 000 tiny reads returns an empty list and the process's resident memory stays flat over the last
    150 000 reads (kept records would add > 60 MB: tuple + 400-char sequence + read id each).  The list is still what comes back when
    neither a writer nor on_result takes the records."""
    import contextlib
    import gc
    import os
    from radian_amd import basecall

    page = os.sysconf("SC_PAGE_SIZE")

    def rss():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * page

    class Stub:
        """Backend stand-in with the blocking per-batch entry point only (--no-pipeline): 400 labels per read, no arithmetic"""
        lab = (np.arange(400) % 4).astype(np.uint8)

        def basecall_raw_global(self, raws, *a):
            return [self.lab] * len(raws), [0] * len(raws)

    class R:
        sig = np.zeros(8, dtype=np.int16)

        def __init__(self, i):
            self.read_id = f"{i:08d}-0000-4000-8000-{i:012d}"

        def get_raw_data(self):
            return self.sig

    class Writer(basecall.FastaWriter):
        marks = {}

        def write(self, rid, seq):
            super().write(rid, seq)
            n = self.n * 1000 + self.i
            if n in (50000, 200000):
                gc.collect()
                self.marks[n] = rss()

    args = basecall.build_parser().parse_args(["-", str(tmp_path), "--no-pipeline", "--gpu-batch-windows", "512"])
    args._lm_loaded = False
    w = Writer(str(tmp_path))
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        try:
            res = basecall.run(args, Stub(), reads=(R(i) for i in range(200000)), writer=w)
        finally:
            w.close()
    assert res == []
    assert set(w.marks) == {50000, 200000}
    grown = (w.marks[200000] - w.marks[50000]) / 1e6
    assert grown < 25.0, f"resident memory grew by {grown:.0f} MB over 150 000 written reads"
    n_rec = sum(1 for f in os.listdir(tmp_path) for line in open(tmp_path / f) if line.startswith(">"))
    assert n_rec == 200000 and len(os.listdir(tmp_path)) == 201      # 200 full files + the empty trailing one (basecall.py:133-138)
    # on_result alone: consumed as well; neither: the list is the way out
    seen = []
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        assert basecall.run(args, Stub(), reads=(R(i) for i in range(10)), writer=None, on_result=lambda *r: seen.append(r)) == []
        kept = basecall.run(args, Stub(), reads=(R(i) for i in range(10)), writer=None)
    assert len(seen) == 10 and [tuple(k) for k in kept] == seen


def test_literal_entry_point_exists_and_parses_the_reference_command_line(tmp_path):
    """`python basecall.py --help` at the repository root (the reference's own entry, basecall.py:143-144) from a foreign working directory:
    the 13 reference flags with their defaults are there and nothing touches a GPU for it."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "basecall.py"), "--help"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    for flag in ("--local", "--chunk-len", "--step-size", "--batch-size", "--outlier-clip", "--rna-model", "--sig-model", "--sig-config",
                 "--beam-width", "--decode-type", "--sig-threshold", "--rna-threshold", "--context-len"):
        assert flag in p.stdout
    assert "fast5_dir" in p.stdout and "fasta_dir" in p.stdout


def test_keras_h5_names_are_checked_not_only_shapes(tmp_path, capsys):
    """Round 6: conv1D_0 and conv1D_1 of blocks 1..5 all have shape [3,256,256]; a file whose order differs from
    load_weights order (model.py:44) must not load silently wrong.  Two same-shaped datasets swapped in the name list -> ValueError naming
    both; the same file converts by name (order='by_name', tools/verify_h5.py --by-name) to exactly the original parameters; a file whose
    names follow no known scheme loads positionally with a warning; session-numbered layer names (tcn_1, dense_2/dense_3) load."""
    import importlib.util
    from radian_amd import h5, h5weights, weights
    dil = (1, 2, 4)
    w = weights.synthetic_weights(seed=3, dilations=dil)
    shapes = weights.tensor_shapes(dil)

    def write(path, order, rename=lambda n: n):
        off, arrs = 0, []
        for name, shape in shapes:
            n = int(np.prod(shape))
            arrs.append((name, w[off:off + n].reshape(shape)))
            off += n
        arrs = [arrs[i] for i in order]
        by_layer = {}
        for name, a in arrs:
            by_layer.setdefault(name.split("/")[0], []).append((rename(name) + ":0", a))
        with h5.File(path, "w") as f:
            for layer in ("tcn", "dense", "dense_1"):
                f.create_group("/" + layer)
                for wn, a in by_layer[layer]:
                    f.write(f"/{layer}/{wn}", a)
                f.set_attr_str("/" + layer, "weight_names", [wn for wn, _ in by_layer[layer]], kind="nullpad")
            f.set_attr_str("/", "layer_names", ["tcn", "dense", "dense_1"], kind="nullpad")

    n = len(shapes)
    good = str(tmp_path / "good.h5")
    write(good, list(range(n)))
    assert np.array_equal(h5weights.read_keras_weights(good, dil), w)
    # block 1's conv1D_0 kernel <-> conv1D_1 kernel: same shape, different place
    names = [s[0] for s in shapes]
    i, j = names.index("tcn/residual_block_1/conv1D_0/kernel"), names.index("tcn/residual_block_1/conv1D_1/kernel")
    order = list(range(n))
    order[i], order[j] = order[j], order[i]
    swapped = str(tmp_path / "swapped.h5")
    write(swapped, order)
    with pytest.raises(ValueError, match="residual_block_1/conv1D_1/kernel.*position.*residual_block_1/conv1D_0/kernel"):
        h5weights.read_keras_weights(swapped, dil)
    assert np.array_equal(h5weights.read_keras_weights(swapped, dil, order="by_name"), w)
    # blocks 1 and 2 swapped wholesale (every shape still fits)
    b1 = [k for k, nm in enumerate(names) if "residual_block_1/" in nm]
    b2 = [k for k, nm in enumerate(names) if "residual_block_2/" in nm]
    order = list(range(n))
    for a, b in zip(b1, b2):
        order[a], order[b] = b, a
    write(str(tmp_path / "blocks.h5"), order)
    with pytest.raises(ValueError, match="order differs"):
        h5weights.read_keras_weights(str(tmp_path / "blocks.h5"), dil)
    # unknown naming scheme: positional, with a warning that the order is unverifiable
    write(str(tmp_path / "renamed.h5"), list(range(n)), rename=lambda nm: nm.replace("residual_block_", "rb").replace("conv1D_", "c"))
    with pytest.warns(UserWarning, match="taken on trust"):
        assert np.array_equal(h5weights.read_keras_weights(str(tmp_path / "renamed.h5"), dil), w)
    with pytest.raises(ValueError, match="cannot be placed by name"):
        h5weights.read_keras_weights(str(tmp_path / "renamed.h5"), dil, order="by_name")
    # session-numbered names (a second model built in the same Keras session): tcn_1/..., dense_2, dense_3
    write(str(tmp_path / "numbered.h5"), list(range(n)),
          rename=lambda nm: nm.replace("tcn/", "tcn_1/").replace("dense_1/", "dense_3/").replace("dense/", "dense_2/"))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert np.array_equal(h5weights.read_keras_weights(str(tmp_path / "numbered.h5"), dil), w)
    # the tool: table + exit code; --by-name writes a blob --sig-model accepts
    spec = importlib.util.spec_from_file_location("verify_h5", os.path.join(ROOT, "tools", "verify_h5.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    cfg = tmp_path / "cfg.yaml"
    import yaml
    cfg.write_text(yaml.safe_dump({"model": {"relu_units": 128, "softmax_units": 5, "tcn": {
        "nb_filters": 256, "kernel_size": 3, "nb_stacks": 1, "dilations": list(dil), "padding": "causal", "use_skip_connections": False,
        "dropout_rate": 0.0, "activation": "relu", "use_batch_norm": False}}}))
    assert tool.main([good, "--sig-config", str(cfg)]) == 0
    assert "OK: loads as it is" in capsys.readouterr().out
    assert tool.main([swapped, "--sig-config", str(cfg)]) == 1
    text = capsys.readouterr().out
    assert "PROBLEM:" in text and "name says 1/conv1D_1/kernel" in text
    blob = str(tmp_path / "fixed.rdnw")
    assert tool.main([swapped, "--sig-config", str(cfg), "--by-name", blob]) == 0
    flat, d2 = weights.unpack_blob(open(blob, "rb").read())
    assert tuple(d2) == dil and np.array_equal(flat, w)
