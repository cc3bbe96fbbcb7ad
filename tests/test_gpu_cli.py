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
    assert report["ranks"] == [{"device": 0, "contexts": 2, "transport": "rccl"}], report
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
