"""The drop-in CLI on the GPU: fast5 in, FASTA out (BASELINE configs[0] geometry: the 5 reads of data/reads.fast5,
chunk decode, beam 1) -- compared with the oracle's decode/assembly/stitch applied to the GPU's own probabilities
(forward parity is checked separately within 1e-4; decode is discontinuous in its input)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _make_inputs(tmp_path, golden_dir, k=3):
    from radian_amd import fast5
    ids = json.load(open(os.path.join(golden_dir, "reads_fast5_ids.json")))["read_ids"]
    sig = np.load(os.path.join(golden_dir, "reads_fast5_signals.npz"))
    in_dir = tmp_path / "fast5"
    in_dir.mkdir()
    fast5.write_multi_fast5(str(in_dir / "reads.fast5"), {r: sig[r] for r in ids})
    rng = np.random.default_rng(21)
    raw = {}
    for i in range(4 ** k):
        ctx = "".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k))
        raw[ctx] = [float(x) for x in rng.dirichlet([0.3] * 4)]
    lm_path = tmp_path / "lm.json"
    lm_path.write_text(json.dumps(raw))
    return ids, sig, str(in_dir), str(lm_path)


def _read_fasta(d):
    recs = []
    for fn in sorted(os.listdir(d)):
        lines = open(os.path.join(d, fn)).read().split("\n")
        for i in range(0, len(lines) - 1, 2):
            recs.append((lines[i][1:], lines[i + 1]))
    return recs


def _expected(be, oracle, ids, sig, chunk, step, W, mode, table=None, k=0):
    out = []
    for r in ids:
        win, pad = oracle.get_windows(oracle.mad_normalise(sig[r], 4), chunk, step)   # the expected side is oracle-only
        probs = be.forward(np.asarray(win, dtype=np.float32))
        if mode == "chunk":
            frags = []
            for i in range(probs.shape[0]):
                m = probs[i] if i < probs.shape[0] - 1 else probs[i][: chunk - pad]
                frags.append(oracle.beam_search(m, "ACGT", W))
            seq = oracle.chunk_consensus(frags)
        else:
            mat = oracle.assemble_matrices(probs, pad, step)
            seq = oracle.beam_search(mat, "ACGT", W, table, 0.5, 0.5, k)
        out.append((r, seq[::-1]))
    return out


def test_cli_chunk_beam1_config0(tmp_path, golden_dir, oracle, capsys):
    from radian_amd import Backend, basecall, weights
    ids, sig, in_dir, _ = _make_inputs(tmp_path, golden_dir)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    basecall.main([in_dir, str(out_dir), "--decode-type", "chunk", "--beam-width", "1", "--step-size", "512",
                   "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", "None"])
    got = _read_fasta(str(out_dir))
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = _expected(be, oracle, ids, sig, 1024, 512, 1, "chunk")
    be.close()
    assert got == exp
    assert capsys.readouterr().out.count("Basecalled read ") == 5


def test_cli_global_default_geometry_with_lm(tmp_path, golden_dir, oracle):
    """reference defaults: chunk 1024, step 128, beam 6, global decode with the k-mer LM (k=3 synthetic table)."""
    from radian_amd import Backend, basecall, weights, lm
    ids, sig, in_dir, lm_path = _make_inputs(tmp_path, golden_dir)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    basecall.main([in_dir, str(out_dir), "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", lm_path,
                   "--context-len", "3", "--gpu-batch-windows", "200"])
    got = _read_fasta(str(out_dir))
    table, k = lm.load_json(lm_path)
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = _expected(be, oracle, ids, sig, 1024, 128, 6, "global", table, k)
    be.close()
    assert got == exp
    with pytest.raises(KeyError):
        basecall.main([in_dir, str(out_dir), "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", lm_path])


def test_cli_worker_path_with_rccl_single_rank(tmp_path, golden_dir):
    """The multi-GPU worker route (RCCL id rendezvous, communicator init, artefact broadcast, shard, merge) with
    one rank: must give the same FASTA as the in-process run."""
    from radian_amd import basecall, launch
    ids, sig, in_dir, _ = _make_inputs(tmp_path, golden_dir)
    a_dir, b_dir = tmp_path / "a", tmp_path / "b"
    a_dir.mkdir()
    b_dir.mkdir()
    argv = [in_dir, str(a_dir), "--decode-type", "chunk", "--beam-width", "3", "--step-size", "512", "--sig-model", "synthetic:7",
            "--sig-config", "none", "--rna-model", "None"]
    basecall.main(argv)
    argv_b = [in_dir, str(b_dir)] + argv[2:] + ["--device-contexts", "2"]
    args = basecall.build_parser().parse_args(argv_b)
    args.gpus = 1
    report = launch.run_multi_gpu(args, argv_b)
    assert _read_fasta(str(a_dir)) == _read_fasta(str(b_dir))
    assert len(_read_fasta(str(b_dir))) == 5
    # the worker honours --device-contexts like the single-GPU CLI: the further contexts are filled by rd_clone_artifacts
    # from the one that received the broadcast; the one-rank communicator is RCCL's
    assert [{k: r[k] for k in ("device", "contexts", "transport")} for r in report["ranks"]] == [{"device": 0, "contexts": 2, "transport": "rccl"}], report
    assert report["ranks"][0]["cpu_split"] == "all" and not report["ranks"][0]["cpu_bound"]      # one rank: the whole node is its budget
    assert report["records"] == 5


def test_cli_cfg5_flags_f16_logits_and_hashed_long_context(tmp_path, golden_dir, oracle):
    """the configs[4] flags end to end through the CLI: --logits f16 --lm-hashed-context --context-len 40 --beam-width 25 (global),
    against the oracle's long-context decode of the f16-rounded GPU rows; and both precision flags reach the device."""
    from radian_amd import Backend, basecall, weights, lm
    mad_normalise, get_windows = oracle.mad_normalise, oracle.get_windows
    ids, sig, in_dir, lm_path = _make_inputs(tmp_path, golden_dir)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    basecall.main([in_dir, str(out_dir), "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", lm_path, "--step-size", "512",
                   "--context-len", "40", "--lm-hashed-context", "--logits", "f16", "--beam-width", "25", "--sig-threshold", "0.0",
                   "--rna-threshold", "5.0"])
    got = _read_fasta(str(out_dir))
    table, k = lm.load_json(lm_path)
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = []
    for r in ids:
        win, pad = get_windows(mad_normalise(sig[r], 4), 1024, 512)
        probs = be.forward(np.asarray(win, dtype=np.float32)).astype(np.float16).astype(np.float32)
        mat = oracle.assemble_matrices(probs, pad, 512)
        lab = oracle.beam_search_batch(mat, [0], [mat.shape[0]], 25, table, 0.0, 5.0, 40, hash_order=k)[0]
        exp.append((r, "".join("ACGT"[c] for c in lab)[::-1]))
    be.close()
    assert got == exp
    with pytest.raises(KeyError):   # without the flag a context length that is not the model's stays the reference's KeyError
        basecall.main([in_dir, str(out_dir), "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", lm_path, "--context-len", "40"])
    out2 = tmp_path / "out2"
    out2.mkdir()
    basecall.main([in_dir, str(out2), "--decode-type", "chunk", "--beam-width", "3", "--step-size", "512", "--sig-model", "synthetic:7",
                   "--sig-config", "none", "--rna-model", "None", "--precision", "bf16x3"])
    assert len(_read_fasta(str(out2))) == 5
    # --decode-math glibc reaches the device; on softmax rows it gives the labelings the default arithmetic gives
    outs = []
    for math in ("glibc", "fast"):
        o = tmp_path / ("out_" + math)
        o.mkdir()
        basecall.main([in_dir, str(o), "--decode-type", "chunk", "--beam-width", "6", "--step-size", "512", "--sig-model", "synthetic:1234",
                       "--sig-config", "none", "--rna-model", "None", "--decode-math", math])
        outs.append(_read_fasta(str(o)))
    assert len(outs[0]) == 5 and outs[0] == outs[1]


def _write_default_artifacts(cwd, seed, k, dilations=(1, 2, 4, 8, 16, 32), nb_stacks=1, tcn_overrides=None, lm_name="rnamodel_12mer_pc.json"):
    """The three artefact files at the reference's DEFAULT relative paths (basecall.py:28-30) under `cwd`: models/sig2seq.h5
    (Keras-2.4 weights-only layout, NUL-padded name attributes), models/sig2seq.yaml (the keys of sig2seq.yaml:34-49) and the
    RNA model JSON."""
    import yaml
    from radian_amd import h5weights, weights
    models = cwd / "models"
    models.mkdir(exist_ok=True)
    all_dil = tuple(dilations) * nb_stacks
    flat = weights.synthetic_weights(seed=seed, dilations=all_dil)
    h5weights.write_keras_weights(str(models / "sig2seq.h5"), flat, dilations=all_dil, attr_kind="nullpad")
    tcn = {"nb_filters": 256, "kernel_size": 3, "nb_stacks": nb_stacks, "dilations": list(dilations), "padding": "causal",
           "use_skip_connections": False, "dropout_rate": 0.0, "return_sequences": True, "activation": "relu",
           "kernel_initializer": "he_normal", "use_batch_norm": False}
    tcn.update(tcn_overrides or {})
    cfg = {"data": {"n_classes": 5, "window_size": 1024},
           "model": {"relu_units": 128, "softmax_units": 5, "timesteps": 1024, "tcn": tcn}}
    (models / "sig2seq.yaml").write_text(yaml.safe_dump(cfg))
    rng = np.random.default_rng(21)
    raw = {}
    for i in range(4 ** k):
        ctx = "".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k))
        raw[ctx] = [float(x) for x in rng.dirichlet([0.3] * 4)]
    (models / lm_name).write_text(json.dumps(raw))
    return flat, all_dil


def test_cli_default_artifact_route(tmp_path, golden_dir, oracle, monkeypatch):
    """The route a RADIAN user takes (basecall.py:28-30,48-62; model.py:42-45; utilities.py:16-18): NO --sig-model, NO --sig-config,
    NO --rna-model -- models/sig2seq.h5 goes through the Keras-h5 converter, models/sig2seq.yaml through load_dilations, the RNA
    model JSON at its default path through lm.load_json, all relative to the working directory, and the result feeds the device.
    FASTA == the `synthetic:1234` run with explicit flags == the oracle's decode of the GPU's probabilities."""
    from radian_amd import Backend, basecall, weights, lm
    ids, sig, in_dir, lm_path = _make_inputs(tmp_path, golden_dir)
    cwd = tmp_path / "cwd"
    cwd.mkdir()
    flat, dil = _write_default_artifacts(cwd, 1234, 3)
    assert np.array_equal(flat, weights.synthetic_weights(seed=1234))
    monkeypatch.chdir(cwd)
    out_a, out_b, out_c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for o in (out_a, out_b, out_c):
        o.mkdir()
    basecall.main([in_dir, str(out_a), "--context-len", "3"])                 # every artefact flag at its default
    basecall.main([in_dir, str(out_b), "--context-len", "3", "--sig-model", "synthetic:1234", "--sig-config", "none",
                   "--rna-model", lm_path])
    got = _read_fasta(str(out_a))
    assert len(got) == 5 and got == _read_fasta(str(out_b))
    table, k = lm.load_json(str(cwd / "models" / "rnamodel_12mer_pc.json"))
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = _expected(be, oracle, ids, sig, 1024, 128, 6, "global", table, k)
    be.close()
    assert got == exp
    # chunk mode never reads the RNA model (basecall.py:110-121): same default files, LM ignored
    basecall.main([in_dir, str(out_c), "--decode-type", "chunk", "--step-size", "512", "--beam-width", "10"])
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = _expected(be, oracle, ids, sig, 1024, 512, 10, "chunk")
    be.close()
    assert _read_fasta(str(out_c)) == exp


def test_cli_literal_entry_point_from_a_models_directory(tmp_path, golden_dir, oracle):
    """The command a RADIAN user types (README.md:56-59, basecall.py:28-30,143-144): `python3 <repo>/basecall.py fast5_dir fasta_dir`
    from a working directory that holds models/ -- a fresh interpreter, no flags for the artefacts, the package found beside the script
    (not through the cwd).  FASTA == the oracle's decode of the GPU's probabilities; stdout carries the reference's per-read lines."""
    import subprocess
    from radian_amd import Backend, weights, lm
    ids, sig, in_dir, _ = _make_inputs(tmp_path, golden_dir)
    cwd = tmp_path / "cwd"
    cwd.mkdir()
    _write_default_artifacts(cwd, 1234, 3)
    out = tmp_path / "out"
    out.mkdir()
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "basecall.py"), in_dir, str(out), "--context-len", "3"], cwd=str(cwd), env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("Basecalled read ")]
    assert [ln.split()[2] for ln in lines] == list(ids) and all(ln.endswith(" sec.") for ln in lines)
    table, k = lm.load_json(str(cwd / "models" / "rnamodel_12mer_pc.json"))
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp = _expected(be, oracle, ids, sig, 1024, 128, 6, "global", table, k)
    be.close()
    assert _read_fasta(str(out)) == exp


def test_cli_default_artifact_route_two_stacks_and_rejections(tmp_path, golden_dir, oracle, monkeypatch):
    """sig2seq.yaml drives the graph: nb_stacks 2 x dilations [1, 2, 4] (six blocks with dilations 1,2,4,1,2,4) loads a
    matching .h5 and basecalls like the oracle; a config the backend does not implement (batch norm, another filter count) and an
    .h5 whose tensor list does not fit the config raise ValueError before any read is touched."""
    from radian_amd import Backend, basecall, weights, h5weights
    ids, sig, in_dir, _ = _make_inputs(tmp_path, golden_dir)
    cwd = tmp_path / "cwd"
    cwd.mkdir()
    flat, dil = _write_default_artifacts(cwd, 77, 3, dilations=(1, 2, 4), nb_stacks=2)
    assert dil == (1, 2, 4, 1, 2, 4)
    monkeypatch.chdir(cwd)
    out = tmp_path / "o"
    out.mkdir()
    basecall.main([in_dir, str(out), "--rna-model", "None", "--step-size", "512", "--beam-width", "3"])
    be = Backend(0)
    be.load_weights(flat, dil)
    exp = []
    for r in ids:
        win, pad = oracle.get_windows(oracle.mad_normalise(sig[r], 4), 1024, 512)
        win = np.asarray(win, dtype=np.float32)
        probs = be.forward(win)
        assert float(np.abs(probs - oracle.tcn_forward(flat, win, dilations=dil)).max()) <= 1e-4
        exp.append((r, oracle.beam_search(oracle.assemble_matrices(probs, pad, 512), "ACGT", 3)[::-1]))
    be.close()
    assert _read_fasta(str(out)) == exp
    # 12-vs-13 tensors: the default six-block config against an .h5 that lacks the matching conv's bias
    _write_default_artifacts(cwd, 77, 3)
    from radian_amd import h5
    shapes = weights.tensor_shapes()
    bad = str(cwd / "models" / "sig2seq.h5")
    os.remove(bad)
    with h5.File(bad, "w") as f:
        names = []
        f.create_group("/tcn")
        for name, shape in shapes[:12]:
            f.write(f"/tcn/{name}:0", np.zeros(shape, dtype=np.float32))
            names.append(f"{name}:0")
        f.set_attr_str("/tcn", "weight_names", names, kind="nullpad")
        f.set_attr_str("/", "layer_names", ["tcn"], kind="nullpad")
    with pytest.raises(ValueError, match="weight tensors"):
        basecall.main([in_dir, str(out), "--rna-model", "None"])
    for override, msg in (({"use_batch_norm": True}, "does not implement"), ({"nb_filters": 128}, "geometry"),
                          ({"use_skip_connections": True}, "does not implement"), ({"dropout_rate": 0.1}, "does not implement")):
        _write_default_artifacts(cwd, 77, 3, tcn_overrides=override)
        with pytest.raises(ValueError, match=msg):
            basecall.main([in_dir, str(out), "--rna-model", "None"])
    # the default six-block config against an .h5 written for a three-block graph
    _write_default_artifacts(cwd, 77, 3)
    h5weights.write_keras_weights(bad, weights.synthetic_weights(seed=1, dilations=(1, 2, 4)), dilations=(1, 2, 4))
    with pytest.raises(ValueError):
        basecall.main([in_dir, str(out), "--rna-model", "None"])


def test_cli_sparse_rna_model_fails_like_the_reference(tmp_path, golden_dir, oracle):
    """A JSON that lacks a context loads (basecall.py:48-57 builds a dict of whatever it holds); the run dies with KeyError at the
    first read whose beam search looks that context up (decode.py:83) and the reads before it are in the FASTA -- through the
    in-context pipeline and through the blocking calls; reads that never reach the context decode as with the dense model."""
    from radian_amd import Backend, basecall, weights, lm
    ids, sig, in_dir, lm_path = _make_inputs(tmp_path, golden_dir, k=3)
    dense = json.load(open(lm_path))
    table, k = lm.load_json(lm_path)
    flat = weights.synthetic_weights(seed=1234).copy()
    flat[-645:-5] *= np.float32(0.05)          # soft head: labelings of hundreds of bases, so contexts are reached
    wpath = str(tmp_path / "soft.rdnw")
    open(wpath, "wb").write(weights.pack_blob(flat))
    be = Backend(0)
    be.load_weights(flat)
    raws = [sig[r] for r in ids]
    be.load_lm(table, k)
    full, _ = be.basecall_raw_global(raws, 4, 1024, 512, 6, True, 0.0, 9.0)
    found = None
    omats = []
    for r in ids:
        win, pad = oracle.get_windows(oracle.mad_normalise(sig[r], 4), 1024, 512)
        omats.append(oracle.assemble_matrices(be.forward(np.asarray(win, dtype=np.float32)), pad, 512))
    for c in range(4 ** k):
        t = table.copy()
        t[c] = np.nan
        be.load_lm(t, k)
        labs, status = be.basecall_raw_global(raws, 4, 1024, 512, 6, True, 0.0, 9.0)
        bad = [i for i, l in enumerate(labs) if l is None]
        splits = bool(bad) and bad[0] >= 1 and (len(bad) < len(ids) or c >= 12)
        # the oracle agrees read by read on the GPU's own probabilities (checked for the first two contexts and for the one the test goes on
        # with -- round 6, suite budget: the oracle's five searches per context were most of this test's 29 s)
        if c < 2 or splits:
            for i, mat in enumerate(omats):
                exp = oracle.beam_search_batch(mat, [0], [mat.shape[0]], 6, t, 0.0, 9.0, k)[0]
                assert (exp is None) == (labs[i] is None) and (exp is None or np.array_equal(exp, labs[i])), (c, i)
        if bad and bad[0] >= 1 and len(bad) < len(ids):
            assert all(labs[i] is None or np.array_equal(labs[i], full[i]) for i in range(len(ids)))
            found = (c, bad)
            break
        if c >= 12 and found is None and bad and bad[0] >= 1:
            found = (c, bad)
            break
    be.close()
    assert found is not None, "no absent context splits the five reads"
    c, bad = found
    ctx = "".join("ACGT"[(c >> (2 * (k - 1 - j))) & 3] for j in range(k))
    del dense[ctx]
    sparse_path = tmp_path / "sparse.json"
    sparse_path.write_text(json.dumps(dense))
    for extra in ([], ["--no-pipeline"]):
        out = tmp_path / ("o" + str(len(extra)))
        out.mkdir()
        with pytest.raises(KeyError, match="decode.py:83"):
            basecall.main([in_dir, str(out), "--sig-model", wpath, "--sig-config", "none", "--rna-model", str(sparse_path), "--context-len", "3",
                           "--step-size", "512", "--sig-threshold", "0.0", "--rna-threshold", "9.0"] + extra)
        got = _read_fasta(str(out))
        assert got == [(ids[i], "".join("ACGT"[x] for x in full[i])[::-1]) for i in range(bad[0])]


def test_cli_context_len_mismatch_fails_lazily_like_the_reference(tmp_path, golden_dir, oracle):
    """--context-len 5 with a model of 3-character keys: the reference loads it (basecall.py:48-57), basecalls every read whose beam
    search never keeps a labeling of 5 labels -- no lookup is ever made, so as without an LM -- and dies with KeyError at decode.py:83 on the
    first read that does (a tuple of 5 labels is no key of the dict).  Here: the same reads in the FASTA, then the KeyError; the expectation
    is the oracle's with a 5-label model that holds no context at all, on the GPU's own probabilities."""
    from radian_amd import Backend, basecall, fast5, weights
    ids, sig, _, lm_path = _make_inputs(tmp_path, golden_dir, k=3)
    rng = np.random.default_rng(5)
    reads = {"00000000-short-a": np.round(rng.normal(500, 80, size=6)).astype(np.int16),
             "00000000-short-b": np.round(rng.normal(500, 80, size=9)).astype(np.int16)}
    reads.update({r: sig[r] for r in ids[:2]})
    order = sorted(reads)       # (a multi-read fast5 iterates its groups in name order)
    in_dir = tmp_path / "mixed"
    in_dir.mkdir()
    fast5.write_multi_fast5(str(in_dir / "reads.fast5"), reads)
    flat = weights.synthetic_weights(seed=1234).copy()
    flat[-645:-5] *= np.float32(0.05)          # soft head: labelings of hundreds of bases on the long reads
    wpath = str(tmp_path / "soft.rdnw")
    open(wpath, "wb").write(weights.pack_blob(flat))
    be = Backend(0)
    be.load_weights(flat)
    absent = np.full((4 ** 5, 4), np.nan)
    exp = []
    for r in order:
        win, pad = oracle.get_windows(oracle.mad_normalise(reads[r], 4), 1024, 512)
        mat = oracle.assemble_matrices(be.forward(np.asarray(win, dtype=np.float32)), pad, 512)
        lab = oracle.beam_search_batch(mat, [0], [mat.shape[0]], 6, absent, 0.5, 0.5, 5)[0]
        exp.append(None if lab is None else (r, "".join("ACGT"[c] for c in lab)[::-1]))
    be.close()
    first_bad = next(i for i, e in enumerate(exp) if e is None)
    assert first_bad == 2, "the two short reads decode, the first long one reaches a 5-label labeling"
    for extra in ([], ["--no-pipeline"]):
        out = tmp_path / ("c" + str(len(extra)))
        out.mkdir()
        with pytest.raises(KeyError, match="decode.py:83"):
            basecall.main([str(in_dir), str(out), "--sig-model", wpath, "--sig-config", "none", "--rna-model", lm_path, "--context-len", "5",
                           "--step-size", "512"] + extra)
        assert _read_fasta(str(out)) == exp[:first_bad]


@pytest.mark.parametrize("W", [100, 260])
def test_cli_beam_width_above_the_lane_kernels(tmp_path, golden_dir, oracle, W):
    """--beam-width 100 / 260 (the reference slices `sort_labelings()[:beam_width]` with any width, decode.py:145): the reads of data/reads.fast5
    through the CLI -- the pipeline's groups launch the lane kernels' wide forms (65 ... 128 beams: five waves, the beam set in two halves; round 6) in both
    decode types, and csrc/decode_wide.hip (W = 260: chunk mode, the first two reads compared -- the oracle's side at that width is most of the test's time)
    -- equal to the oracle's decode of the GPU's own probabilities."""
    from radian_amd import Backend, basecall, weights, lm
    ids, sig, in_dir, lm_path = _make_inputs(tmp_path, golden_dir, k=3)
    table, k = lm.load_json(lm_path)
    n_cmp = len(ids) if W <= 256 else 2
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    exp_chunk = _expected(be, oracle, ids[:n_cmp], sig, 1024, 512, W, "chunk")
    exp_global = _expected(be, oracle, ids, sig, 1024, 512, W, "global", table, k) if W <= 256 else None
    be.close()
    for mode, exp, extra in (("chunk", exp_chunk, ["--rna-model", "None"]), ("global", exp_global, ["--rna-model", lm_path, "--context-len", "3"])):
        if exp is None:
            continue
        out = tmp_path / f"w{W}_{mode}"
        out.mkdir()
        basecall.main([in_dir, str(out), "--decode-type", mode, "--beam-width", str(W), "--step-size", "512", "--sig-model", "synthetic:1234",
                       "--sig-config", "none"] + extra)
        got = _read_fasta(str(out))
        assert len(got) == len(ids) and got[:n_cmp] == exp, mode


def test_cli_reads_filtered_fast5_like_raw(tmp_path, golden_dir):
    """The same five signals stored raw and through HDF5's shuffle + deflate + Fletcher-32 filters (csrc/fast5.hip undoes them natively;
    pre-VBZ MinKNOW files are gzip-compressed): identical FASTA from the CLI."""
    from radian_amd import basecall, fast5
    ids, sig, in_dir, _ = _make_inputs(tmp_path, golden_dir)
    gz_dir = tmp_path / "fast5_gz"
    gz_dir.mkdir()
    fast5.write_multi_fast5(str(gz_dir / "reads.fast5"), {r: sig[r] for r in ids}, filters=("shuffle", ("deflate", 1), "fletcher32"), chunk=1500)
    nf = fast5.NativeFile(str(gz_dir / "reads.fast5"))          # (no fall-back behind this call)
    names, samples, off = nf.batch(0, nf.n)
    nf.close()
    assert names == sorted(ids) and all(np.array_equal(samples[off[i]:off[i + 1]], sig[r]) for i, r in enumerate(names))
    outs = []
    for d in (in_dir, str(gz_dir)):
        out = tmp_path / ("out_" + os.path.basename(d))
        out.mkdir()
        basecall.main([d, str(out), "--decode-type", "chunk", "--beam-width", "3", "--step-size", "512", "--sig-model", "synthetic:7",
                       "--sig-config", "none", "--rna-model", "None"])
        outs.append(_read_fasta(str(out)))
    assert outs[0] == outs[1] and len(outs[0]) == len(ids)
