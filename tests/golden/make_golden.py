#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the upstream reference.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It imports the reference's pure-Python hot-path modules read-only
(/root/reference/radian/{decode,matrix_assembly,preprocess,sequence_assembly}.py),
feeds them seeded synthetic inputs and stores *inputs and outputs only* (no reference
source) as .npz/.json fixtures.  `decode.py` does `import tensorflow` solely for a type
annotation (decode.py:11,104); TensorFlow is not installed here, so an empty placeholder
module is registered before the import.

The fixtures pin oracle/ (the CPU restatement), which in turn is the checker for the HIP path.
Nothing in tests/, bench.py or the package reads /root/reference at run time.
"""
import json
import math
import os
import re
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/radian"

sys.dont_write_bytecode = True
_tf = types.ModuleType("tensorflow")
_tf.keras = types.SimpleNamespace(Model=object)
sys.modules.setdefault("tensorflow", _tf)
sys.path.insert(0, REF)

# The reference pins numpy~=1.19.5 (requirements.txt:5) and calls `np.lib.pad`
# (sequence_assembly.py:33), which numpy 2.x removed; in 1.19 it is the same object as np.pad.
if not hasattr(np.lib, "pad"):
    np.lib.pad = np.pad

import decode as ref_decode  # noqa: E402
import matrix_assembly as ref_asm  # noqa: E402
import preprocess as ref_pre  # noqa: E402
import sequence_assembly as ref_seq  # noqa: E402

BASES = "ACGT"


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def softmax_rows(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def make_matrix(rng, T, kind, dtype):
    """Seeded synthetic softmax matrices [T,5]. kinds:
    flat   - near-uniform rows (random-weight network; worst case for beam search)
    peaky  - logits x4, blank +2 (trained-network-like)
    hard   - includes exact 0.0 / 1.0 probabilities (float32 softmax underflow: -inf ties)
    dup    - rows with exactly equal base probabilities (tie order)
    """
    z = rng.normal(size=(T, 5))
    if kind == "flat":
        m = softmax_rows(z)
    elif kind == "peaky":
        z = z * 4.0
        z[:, 4] += 2.0
        m = softmax_rows(z)
    elif kind == "hard":
        z = z * 4.0
        m = softmax_rows(z)
        for t in range(T):
            r = rng.integers(0, 6)
            if r == 0:  # one-hot row
                k = rng.integers(0, 5)
                m[t] = 0.0
                m[t, k] = 1.0
            elif r == 1:  # two zeros
                ks = rng.choice(5, size=2, replace=False)
                m[t, ks] = 0.0
                m[t] /= m[t].sum()
            elif r == 2:  # blank zero
                m[t, 4] = 0.0
                m[t] /= m[t].sum()
    elif kind == "dup":
        m = softmax_rows(z)
        for t in range(T):
            if rng.integers(0, 2) == 0:
                a, b = rng.choice(4, size=2, replace=False)
                m[t, b] = m[t, a]
            if rng.integers(0, 4) == 0:
                m[t, :4] = m[t, 0]
        m = m / m.sum(axis=1, keepdims=True)
        # re-impose exact equality after the renormalisation
        m = m.astype(dtype)
        return np.ascontiguousarray(m)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(m.astype(dtype))


def make_lm(rng, k, alpha=0.3):
    """Dense synthetic k-mer LM: dict {tuple(int*k): [pA,pC,pG,pT]} with every context present,
    plus the same numbers as a dense table [4^k,4] (index = base-4 number, first label most significant)."""
    table = rng.dirichlet([alpha] * 4, size=4 ** k)
    table = np.ascontiguousarray(table.astype(np.float64))
    lm = {}
    for idx in range(4 ** k):
        ctx = tuple((idx >> (2 * (k - 1 - i))) & 3 for i in range(k))
        lm[ctx] = [float(x) for x in table[idx]]
    return lm, table


_captured = {}
_orig_sort = ref_decode.BeamList.sort_labelings


def _capturing_sort(self):
    # records the entries of the most recent BeamList that was sorted (the last call inside
    # beam_search is the final `last.sort_labelings()[0]`, decode.py:207)
    _captured["entries"] = self.entries
    return _orig_sort(self)


def run_beam(mat, W, lm=None, s_thr=None, r_thr=None, ctx=None, capture=False):
    cache = {} if lm is not None else None
    if capture:
        ref_decode.BeamList.sort_labelings = _capturing_sort
    try:
        seq = ref_decode.beam_search(mat, BASES, W, lm, s_thr, r_thr, ctx, cache)
    finally:
        ref_decode.BeamList.sort_labelings = _orig_sort
    if not capture:
        return seq, None
    ents = _captured.pop("entries")
    order = sorted(ents.values(), reverse=True, key=lambda x: x.pr_total)
    final = [
        {
            "labeling": "".join(BASES[c] for c in e.labeling),
            "pr_total": float(e.pr_total),
            "pr_blank": float(e.pr_blank),
            "pr_non_blank": float(e.pr_non_blank),
        }
        for e in order[:30]
    ]
    return seq, final


def fenc(x):
    """JSON-safe float encoding preserving bits: hex string."""
    return float(x).hex()


# ----------------------------------------------------------------------------------------------
# 1. beam_search without LM (chunk mode semantics; float32 matrices)
# ----------------------------------------------------------------------------------------------
def gen_beam_nolm():
    rng = np.random.default_rng(20240501)
    arrays = {}
    cases = []
    cid = 0
    for T in (0, 1, 2, 8, 64, 512, 1024):
        for kind in ("flat", "peaky", "hard", "dup"):
            if T >= 512 and kind == "dup":
                continue
            mat = make_matrix(rng, T, kind, np.float32)
            name = f"m{cid}"
            arrays[name] = mat
            widths = (1, 2, 6, 10, 25) if T <= 512 else (1, 10, 25)
            for W in widths:
                capture = T <= 64
                seq, final = run_beam(mat, W, capture=capture)
                c = {"mat": name, "T": T, "kind": kind, "W": W, "seq": seq}
                if final is not None:
                    c["final"] = [
                        {
                            "labeling": f["labeling"],
                            "pr_total": fenc(f["pr_total"]),
                            "pr_blank": fenc(f["pr_blank"]),
                            "pr_non_blank": fenc(f["pr_non_blank"]),
                        }
                        for f in final
                    ]
                cases.append(c)
            cid += 1
    # float64 matrices without LM (global mode with --rna-model absent is not reachable from the CLI,
    # but the function accepts it; pins the f64-input path)
    for T in (64, 300):
        for kind in ("flat", "peaky"):
            mat = make_matrix(rng, T, kind, np.float64)
            name = f"m{cid}"
            arrays[name] = mat
            for W in (1, 6, 10):
                seq, _ = run_beam(mat, W)
                cases.append({"mat": name, "T": T, "kind": kind, "W": W, "seq": seq})
            cid += 1
    np.savez_compressed(os.path.join(HERE, "beam_nolm_mats.npz"), **arrays)
    with open(os.path.join(HERE, "beam_nolm_cases.json"), "w") as f:
        json.dump({"source": "radian/decode.py:100-212 beam_search(mat,'ACGT',W,None,None,None,None,None)",
                   "cases": cases}, f, indent=0)
    print("beam_nolm:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------
# 2. beam_search with a k-mer LM (global mode semantics; float64 matrices)
# ----------------------------------------------------------------------------------------------
def gen_beam_lm():
    rng = np.random.default_rng(20240502)
    arrays = {}
    cases = []
    cid = 0
    for k in (1, 3, 5):
        lm, table = make_lm(rng, k)
        arrays[f"lm_k{k}"] = table
        for T, kind in ((8, "flat"), (64, "flat"), (64, "peaky"), (200, "hard"), (400, "flat"), (400, "peaky")):
            mat = make_matrix(rng, T, kind, np.float64)
            name = f"m{cid}"
            cid += 1
            arrays[name] = mat
            for W in (1, 6, 10):
                for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf), (math.inf, 0.0), (0.0, 1.0), (1.2, 1.2)):
                    if T >= 200 and W == 1 and (s_thr, r_thr) != (0.5, 0.5):
                        continue
                    capture = T <= 64 and (s_thr, r_thr) == (0.5, 0.5)
                    seq, final = run_beam(mat, W, lm, s_thr, r_thr, k, capture=capture)
                    c = {"mat": name, "lm": f"lm_k{k}", "k": k, "T": T, "kind": kind, "W": W,
                         "s_thr": fenc(s_thr), "r_thr": fenc(r_thr), "seq": seq}
                    if final is not None:
                        c["final"] = [
                            {"labeling": f_["labeling"], "pr_total": fenc(f_["pr_total"]),
                             "pr_blank": fenc(f_["pr_blank"]), "pr_non_blank": fenc(f_["pr_non_blank"])}
                            for f_ in final
                        ]
                    cases.append(c)
    # per-call pairs for the LM gate
    pair_rng = np.random.default_rng(7)
    lm3, table3 = make_lm(pair_rng, 3)
    arrays["lm_pairs_k3"] = table3
    pairs = []
    for i in range(200):
        s = pair_rng.dirichlet([0.5] * 5).astype(np.float64)
        if i % 17 == 0:
            s[pair_rng.integers(0, 4)] = 0.0
        ctx = tuple(int(x) for x in pair_rng.integers(0, 4, size=3))
        s_ent = ref_decode.entropy(ref_decode.normalise(s[:-1]))
        for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf)):
            out = ref_decode.apply_rna_model(s, ctx, lm3, {}, s_ent, r_thr, s_thr)
            pairs.append({"s": [fenc(x) for x in s], "ctx": list(ctx), "s_entropy": fenc(s_ent),
                          "r_entropy": fenc(ref_decode.entropy(np.asarray(lm3[ctx]))),
                          "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                          "out": [fenc(x) for x in np.asarray(out, dtype=np.float64)],
                          "combined": [fenc(x) for x in ref_decode.combine_dists(np.asarray(lm3[ctx]), s)]})
    np.savez_compressed(os.path.join(HERE, "beam_lm_mats.npz"), **arrays)
    with open(os.path.join(HERE, "beam_lm_cases.json"), "w") as f:
        json.dump({"source": "radian/decode.py:79-96,100-212 (LM path), float64 matrices, dense Dirichlet(0.3) LM",
                   "cases": cases, "pairs": pairs}, f, indent=0)
    print("beam_lm:", len(cases), "cases,", len(pairs), "pairs")


# ----------------------------------------------------------------------------------------------
# 2b. beam_search with a SPARSE k-mer LM: the dict lacks some contexts; decode.py:83 `model[context]` raises KeyError when -- and
#     only when -- a kept labeling's context is absent at some time step (round 4: VERDICT r3, "lazy KeyError LM semantics")
# ----------------------------------------------------------------------------------------------
def gen_beam_lm_sparse():
    rng = np.random.default_rng(20261004)
    arrays = {}
    cases = []
    cid = 0
    for k in (1, 2, 3):
        lm, table = make_lm(rng, k)
        arrays[f"lm_k{k}"] = table
        n_ctx = 4 ** k
        for T, kind, reps in ((k + 1, "flat", 12), (k + 2, "peaky", 12), (6, "flat", 10), (12, "flat", 10), (12, "peaky", 10), (40, "flat", 6),
                              (40, "hard", 4), (120, "peaky", 4)):
            for _ in range(reps):
                mat = make_matrix(rng, T, kind, np.float64)
                name = f"m{cid}"
                cid += 1
                arrays[name] = mat
                W = int(rng.choice([1, 2, 3, 6, 10]))
                s_thr, r_thr = [(0.5, 0.5), (0.0, math.inf), (math.inf, 0.0)][int(rng.integers(0, 3))]
                how = int(rng.integers(0, 3))
                n_gone = 1 if how == 0 else max(1, n_ctx // 8) if how == 1 else max(1, n_ctx // 2)
                gone = sorted(int(x) for x in rng.choice(n_ctx, size=min(n_gone, n_ctx), replace=False))
                sparse = {c: d for i, (c, d) in enumerate(lm.items()) if i not in set(gone)}   # (make_lm inserts contexts in index order)
                try:
                    seq, _ = run_beam(mat, W, sparse, s_thr, r_thr, k)
                    outcome = {"seq": seq}
                except KeyError as e:
                    outcome = {"key_error": [int(x) for x in e.args[0]]}
                cases.append({"mat": name, "lm": f"lm_k{k}", "k": k, "T": T, "kind": kind, "W": W, "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                              "missing": gone, **outcome})
    np.savez_compressed(os.path.join(HERE, "beam_lm_sparse_mats.npz"), **arrays)
    with open(os.path.join(HERE, "beam_lm_sparse_cases.json"), "w") as f:
        json.dump({"source": "radian/decode.py:79-96,100-212 with an RNA-model dict that lacks the contexts listed in `missing` "
                             "(table row indices): the labeling, or the KeyError decode.py:83 raises", "cases": cases}, f, indent=0)
    print("beam_lm_sparse:", len(cases), "cases,", sum("key_error" in c for c in cases), "raise KeyError")


# ----------------------------------------------------------------------------------------------
# 3. assemble_matrices
# ----------------------------------------------------------------------------------------------
def gen_assemble():
    rng = np.random.default_rng(20240503)
    arrays = {}
    cases = []
    # (chunk, step, N): the real-size case of BASELINE config 4 plus scaled analogues of the
    # reference defaults (1024,128,12833), step==chunk (1024,1024,2048) and a short read (1024,512,700)
    for (chunk, step, N) in ((1024, 512, 4096), (128, 16, 1605), (64, 64, 128), (64, 32, 44), (64, 16, 64),
                             (64, 48, 200), (32, 1, 40)):
        sig = np.arange(N, dtype=np.float64)
        windows, pad = ref_pre.get_windows(sig, chunk, step)
        nW = windows.shape[0]
        probs = rng.random(size=(nW, chunk, 5), dtype=np.float32) + np.float32(0.01)
        probs /= probs.sum(axis=2, keepdims=True)
        probs = np.ascontiguousarray(probs.astype(np.float32))
        mats = [probs[i] for i in range(nW)]
        mats[-1] = mats[-1][:-pad]  # basecall.py:96
        out = ref_asm.assemble_matrices(mats, step)
        tag = f"c{chunk}_s{step}_n{N}"
        arrays[f"probs_{tag}"] = probs
        arrays[f"out_{tag}"] = out
        cases.append({"tag": tag, "chunk": chunk, "step": step, "N": N, "nW": int(nW), "pad": int(pad),
                      "out_dtype": str(out.dtype), "out_shape": list(out.shape)})
    np.savez_compressed(os.path.join(HERE, "assemble.npz"), **arrays)
    with open(os.path.join(HERE, "assemble_cases.json"), "w") as f:
        json.dump({"source": "radian/matrix_assembly.py:6-53 after basecall.py:96 pad trim", "cases": cases}, f, indent=0)
    print("assemble:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------
# 4. mad_normalise / get_windows
# ----------------------------------------------------------------------------------------------
def gen_preprocess():
    rng = np.random.default_rng(20240504)
    arrays = {}
    cases = []

    def add(name, sig, clip=4, chunk=64, step=16):
        arrays[f"sig_{name}"] = sig
        c = {"name": name, "clip": clip, "chunk": chunk, "step": step}
        try:
            norm = ref_pre.mad_normalise(sig, clip)
            arrays[f"norm_{name}"] = norm
            c["norm_dtype"] = str(norm.dtype)
            windows, pad = ref_pre.get_windows(norm, chunk, step)
            arrays[f"win_{name}"] = windows
            c["pad"] = int(pad)
            c["win_dtype"] = str(windows.dtype)
        except ValueError as e:
            c["error"] = str(e.args[0])
        cases.append(c)

    base = np.round(rng.normal(500, 80, size=1000)).astype(np.int16)
    add("typical", base)
    out_first = base.copy()
    out_first[0] = 3000  # first sample an outlier -> int64 result (np.vectorize otypes inference)
    add("first_outlier_hi", out_first)
    out_first2 = base.copy()
    out_first2[0] = -3000
    add("first_outlier_lo", out_first2)
    add("mad_zero", np.full(100, 512, dtype=np.int16))
    add("empty", np.zeros(0, dtype=np.int16))
    add("short", base[:40])  # N < chunk
    add("mult_step", base[:64 + 16 * 5])  # N = chunk + 5*step
    add("exact_chunk", base[:64])
    add("clip2", base, clip=2)
    add("even_len", np.round(rng.normal(480, 30, size=500)).astype(np.int16), chunk=128, step=128)
    add("odd_len", np.round(rng.normal(480, 30, size=501)).astype(np.int16), chunk=128, step=64)
    spikes = base.copy()
    spikes[rng.integers(0, 1000, size=30)] = 32000
    spikes[rng.integers(0, 1000, size=30)] = -32000
    spikes[0] = 500
    add("spikes", spikes)
    # BASELINE's own read: 4096 samples of round(N(500, 80)) (SURVEY 8d cfg 2; the bench's generator, seed 0), chunk 1024 at step 512 and 128
    bench_read = np.round(np.random.default_rng(0).normal(500.0, 80.0, size=4096)).astype(np.int16)
    add("baseline_4096_step512", bench_read, chunk=1024, step=512)
    add("baseline_4096_step128", bench_read, chunk=1024, step=128)
    # get_windows argument errors
    errs = []
    for (chunk, step) in ((64, 0), (64, 65), (64, -1)):
        try:
            ref_pre.get_windows(np.zeros(100), chunk, step)
            errs.append({"chunk": chunk, "step": step, "error": None})
        except ValueError as e:
            errs.append({"chunk": chunk, "step": step, "error": str(e.args[0])})
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **arrays)
    with open(os.path.join(HERE, "preprocess_cases.json"), "w") as f:
        json.dump({"source": "radian/preprocess.py:4-49", "cases": cases, "window_errors": errs}, f, indent=0)
    print("preprocess:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------
# 5. simple_assembly + index2base
# ----------------------------------------------------------------------------------------------
def gen_seq_assembly():
    rng = np.random.default_rng(20240505)
    cases = []

    def rand_seq(n):
        return "".join(BASES[i] for i in rng.integers(0, 4, size=n))

    def overlapping(total, flen, step, err=0.0):
        truth = rand_seq(total)
        frags = []
        for s in range(0, max(1, total - flen + 1), step):
            fr = list(truth[s:s + flen])
            for i in range(len(fr)):
                if rng.random() < err:
                    fr[i] = BASES[rng.integers(0, 4)]
            frags.append("".join(fr))
        return frags

    frag_sets = [
        overlapping(200, 60, 30),
        overlapping(300, 80, 40, err=0.05),
        overlapping(1500, 150, 75, err=0.02),
        overlapping(2500, 199, 100, err=0.02),  # concensus growth beyond 1000 and 2000
        overlapping(1200, 200, 100),  # fragments of exactly 200 chars -> difflib autojunk
        overlapping(2000, 450, 225, err=0.01),  # long fragments (random-weight worst case)
        [rand_seq(50)],
        ["ACGTACGT", "", "GTACGTAA"],
        ["", "ACGT"],
        ["ACGT", ""],
        ["AAAAAAAAAA", "AAAAAAAAAAAA", "AAAAA"],
        ["ACGTTTGA", "TTGACCA", "GGGGACGTTTGA"],  # negative displacement
        [rand_seq(30), rand_seq(30), rand_seq(30)],  # unrelated fragments
        ["A"],
        ["", ""],
    ]
    for frags in frag_sets:
        cons = ref_seq.simple_assembly(frags)
        if cons.shape[1] == 0:
            seq = ""
        else:
            seq = ref_seq.index2base(np.argmax(cons, axis=0))
        cases.append({"fragments": frags, "consensus_shape": list(cons.shape),
                      "consensus": cons.astype(np.int64).tolist(), "seq": seq})
    with open(os.path.join(HERE, "seq_assembly_cases.json"), "w") as f:
        json.dump({"source": "radian/sequence_assembly.py:19-48,90-97 + basecall.py:122-123", "cases": cases}, f, indent=0)
    print("seq_assembly:", len(cases), "cases")


def gen_seq_assembly_random():
    """Round 2: 160 random fragment lists through the reference's simple_assembly (incl. lists on which it raises
    IndexError: its vote matrix grows by 1000 columns at most once per fragment).  Compact form: shape, consensus string
    and a SHA-256 of the int64 vote matrix instead of the matrix."""
    import hashlib
    rng = np.random.default_rng(20260101)
    cases = []

    def rand_seq(n, alphabet=BASES):
        return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=n))

    for ci in range(160):
        kind = ci % 8
        n = int(rng.integers(0, 8))
        base = rand_seq(int(rng.integers(0, 900)))
        frags, pos = [], 0
        for _ in range(n):
            if kind == 0 or not base:
                f = rand_seq(int(rng.choice([0, 3, 40, 260, 450])), "ACGTacgt" if kind == 0 else BASES)
            else:
                pos = min(len(base), pos + int(rng.integers(0, 130)))
                f = base[pos: pos + int(rng.integers(0, 430))]
                if f and rng.random() < 0.5:
                    i = int(rng.integers(0, len(f)))
                    f = f[:i] + BASES[int(rng.integers(0, 4))] + f[i + 1:]
            frags.append(f)
        if kind == 7:   # provoke the fixed growth step
            frags = [rand_seq(int(rng.integers(950, 1100)))] if ci % 16 == 7 else ["A" * int(rng.integers(900, 1000)), "C", "C" + rand_seq(int(rng.integers(900, 1200)))]
        try:
            cons = ref_seq.simple_assembly(frags)
        except IndexError:
            cases.append({"fragments": frags, "error": "IndexError"})
            continue
        seq = "" if cons.shape[1] == 0 else ref_seq.index2base(np.argmax(cons, axis=0))
        cases.append({"fragments": frags, "consensus_shape": list(cons.shape), "seq": seq,
                      "consensus_sha256": hashlib.sha256(np.ascontiguousarray(cons, dtype=np.int64).tobytes()).hexdigest()})
    with open(os.path.join(HERE, "seq_assembly_random_cases.json"), "w") as f:
        json.dump({"source": "radian/sequence_assembly.py:19-48,90-97 + basecall.py:122-123 (random lists, round 2)", "cases": cases}, f, indent=0)
    print("seq_assembly_random:", len(cases), "cases,", sum("error" in c for c in cases), "IndexError")


# ----------------------------------------------------------------------------------------------
# 6. raw signals of the reference's sample fast5 (data fixture)
# ----------------------------------------------------------------------------------------------
def gen_fast5_signals():
    path = "/root/reference/radian/data/reads.fast5"
    h5ls = "/opt/conda/bin/h5ls"
    h5dump = "/opt/conda/bin/h5dump"
    names = []
    for line in subprocess.check_output([h5ls, path], text=True).splitlines():
        g = line.split()[0]
        if g.startswith("read_"):
            names.append(g)
    arrays = {}
    ids = []
    for g in names:  # h5ls lists in name order == HDF5 group iteration order used by ont_fast5_api
        txt = subprocess.check_output([h5dump, "-y", "-w", "0", "-d", f"/{g}/Raw/Signal", path], text=True)
        start = txt.index("DATA {") + 6
        body = txt[start: txt.index("}", start)]
        vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int16)
        rid = g[len("read_"):]
        ids.append(rid)
        arrays[rid] = vals
    np.savez_compressed(os.path.join(HERE, "reads_fast5_signals.npz"), **arrays)
    with open(os.path.join(HERE, "reads_fast5_ids.json"), "w") as f:
        json.dump({"source": "radian/data/reads.fast5 /read_<id>/Raw/Signal (int16, unscaled DAQ values)",
                   "read_ids": ids, "lengths": [int(arrays[i].shape[0]) for i in ids]}, f, indent=0)
    print("fast5 signals:", [(i[:8], int(arrays[i].shape[0])) for i in ids])


# ----------------------------------------------------------------------------------------------
# 7. end-to-end of the decode side on synthetic probabilities (windows -> trim -> assemble -> beam / chunk stitch)
# ----------------------------------------------------------------------------------------------
def gen_pipeline():
    rng = np.random.default_rng(20240507)
    arrays = {}
    cases = []
    lm, table = make_lm(rng, 3)
    arrays["lm_k3"] = table
    for ci, (chunk, step, N, kind) in enumerate(((128, 32, 500, "peaky"), (128, 64, 777, "flat"), (256, 128, 1024, "peaky"),
                                               (128, 128, 300, "peaky"), (128, 32, 100, "peaky"))):
        sig = np.zeros(N)
        windows, pad = ref_pre.get_windows(sig, chunk, step)
        nW = windows.shape[0]
        probs = np.stack([make_matrix(rng, chunk, kind, np.float32) for _ in range(nW)])
        arrays[f"probs{ci}"] = probs
        mats = [probs[i] for i in range(nW)]
        mats[-1] = mats[-1][:-pad]
        for W in (1, 6):
            # global (basecall.py:99-109)
            matrix = ref_asm.assemble_matrices(mats, step)
            g = ref_decode.beam_search(matrix, BASES, W, lm, 0.5, 0.5, 3, {})
            # chunk (basecall.py:110-123)
            frags = [ref_decode.beam_search(m, BASES, W, None, None, None, None, None) for m in mats]
            cons = ref_seq.simple_assembly(frags)
            c = ref_seq.index2base(np.argmax(cons, axis=0)) if cons.shape[1] else ""
            cases.append({"probs": f"probs{ci}", "chunk": chunk, "step": step, "N": N, "pad": int(pad), "W": W,
                          "k": 3, "global_matrix_dtype": str(matrix.dtype),
                          "global_seq": g, "chunk_fragments": frags, "chunk_seq": c})
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **arrays)
    with open(os.path.join(HERE, "pipeline_cases.json"), "w") as f:
        json.dump({"source": "basecall.py:96-123 on synthetic window probabilities (sequence NOT yet reversed)",
                   "cases": cases}, f, indent=0)
    print("pipeline:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------
# 8. beam_search at the geometries BASELINE.json names (round 5; VERDICT r4 "Missing 3"):
#    (a) the reference's default context 11 (basecall.py:35) with a 4^11-entry model served lazily, on [4096,5] float64 matrices that the
#        reference's own assemble_matrices built at chunk 1024 / step 512 and step 128, W in {6, 10, 25}; dense and sparse (KeyError)
#    (b) no LM at W in {26, 40, 51, 64, 100} (the wide launch forms) at T in {8, 64, 1024}, and a 3-mer LM at those widths
#    (c) float32 single-window reads with an LM, chosen so that numpy 1.19 and 2.x agree (see `entropy_margin`)
# ----------------------------------------------------------------------------------------------
class LazyLM(dict):
    """An RNA model `{context tuple: [pA, pC, pG, pT]}` whose entries are materialised from a dense table when decode.py:83 first asks
    for them (`model[context]` on a dict subclass calls __missing__), so that k = 11 needs no 4^11-entry dict.  A context listed in
    `missing` raises the KeyError a plain dict without that key raises.  `if lm and ...` (decode.py:154,174) tests truthiness, and an
    empty dict is falsy: one present context is stored up front, as any loaded model is non-empty."""

    def __init__(self, table, k, missing=()):
        super().__init__()
        self._table, self._k, self._missing = table, k, frozenset(int(i) for i in missing)
        first = next(i for i in range(table.shape[0]) if i not in self._missing)
        ctx = tuple((first >> (2 * (k - 1 - j))) & 3 for j in range(k))
        self[ctx] = [float(x) for x in table[first]]

    def __missing__(self, ctx):
        idx = 0
        for c in ctx:
            idx = idx * 4 + int(c)
        if idx in self._missing:
            raise KeyError(ctx)
        v = [float(x) for x in self._table[idx]]
        self[ctx] = v
        return v


def _final_json(final):
    return [{"labeling": f_["labeling"], "pr_total": fenc(f_["pr_total"]), "pr_blank": fenc(f_["pr_blank"]),
             "pr_non_blank": fenc(f_["pr_non_blank"])} for f_ in final]


def _entropy_legacy(row32):
    """decode.py:66-76 on a float32 row under the PINNED numpy 1.19 (requirements.txt:5): builtin sum() of np.float32 scalars starts from
    int 0, and value-based casting makes 0 + float32 a float64, so the sum accumulates in float64; `dist / sum(dist)` stays float32 (the
    float64 scalar is cast down); `p * math.log(p)` is float32 * Python float = float64.  Used only to measure how far each row's entropy
    is from the threshold (the margin the case records); the expected strings come from the reference itself."""
    s = 0.0
    for p in row32[:-1]:
        s += float(p)
    if s == 0:
        d = row32[:-1]
    else:
        d = (row32[:-1] / np.float32(s)).astype(np.float32)
    e = 0.0
    for p in d:
        if p > 0:
            e += float(p) * math.log(float(p))
    return -e


def gen_beam_baseline():
    import time
    sys.path.insert(0, os.path.dirname(HERE))
    import _golden_lm
    t_start = time.time()
    rng = np.random.default_rng(20261006)
    arrays, cases = {}, []
    table = _golden_lm.k11_table()
    k = _golden_lm.K11
    meta = {"k11_seed": _golden_lm.K11_SEED, "k11_table_sha256": _golden_lm.table_sha256(table), "numpy": np.__version__}

    # ---- (a) [4096,5] float64 from the reference's assemble_matrices, k = 11
    mats = {}
    for (step, kind) in ((512, "peaky"), (512, "flat"), (128, "peaky")):
        windows, pad = ref_pre.get_windows(np.zeros(4096), 1024, step)
        nW = windows.shape[0]
        probs = np.stack([make_matrix(rng, 1024, kind, np.float32) for _ in range(nW)])
        parts = [probs[i] for i in range(nW)]
        parts[-1] = parts[-1][:-pad]                       # basecall.py:96
        mat = ref_asm.assemble_matrices(parts, step)       # basecall.py:101
        assert mat.shape == (4096, 5) and mat.dtype == np.float64
        name = f"g{step}_{kind}"
        arrays["probs_" + name] = probs
        mats[name] = (mat, step, int(pad), kind)
    for name, (mat, step, pad, kind) in mats.items():
        for W in (6, 10, 25):
            for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf)):
                if kind == "flat" and W == 25 and s_thr == 0.0:
                    continue
                t0 = time.time()
                seq, final = run_beam(mat, W, LazyLM(table, k), s_thr, r_thr, k, capture=True)
                cases.append({"group": "global_k11", "probs": "probs_" + name, "chunk": 1024, "step": step, "pad": pad, "kind": kind, "T": 4096,
                              "mat_sha256": _golden_lm.table_sha256(mat), "k": k, "W": W, "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                              "seq": seq, "final": _final_json(final[:8])})
                print(f"  global_k11 {name} W={W} thr=({s_thr},{r_thr}) len={len(seq)} {time.time() - t0:.1f}s", flush=True)
    # ---- sparse k = 11: the same matrices, models that lack a few contexts (listed by row index)
    mrng = np.random.default_rng(20261007)
    for name in ("g512_peaky", "g128_peaky"):
        mat, step, pad, kind = mats[name]
        for W in (6, 10):
            for frac in (2e-6, 2e-5, 2e-4, 2e-3):
                missing = sorted(int(x) for x in np.flatnonzero(mrng.random(4 ** k) < frac))
                try:
                    seq, _ = run_beam(mat, W, LazyLM(table, k, missing), 0.5, 0.5, k)
                    outcome = {"seq": seq}
                except KeyError as e:
                    outcome = {"key_error": [int(x) for x in e.args[0]]}
                cases.append({"group": "global_k11_sparse", "probs": "probs_" + name, "chunk": 1024, "step": step, "pad": pad, "kind": kind, "T": 4096,
                              "k": k, "W": W, "s_thr": fenc(0.5), "r_thr": fenc(0.5), "missing": missing, **outcome})
                print(f"  sparse {name} W={W} frac={frac} missing={len(missing)} -> {'KeyError' if 'key_error' in outcome else 'decodes'}", flush=True)

    # ---- (b) wide beams without an LM, float32 (chunk-mode semantics), and with a 3-mer LM on float64
    lm3, table3 = make_lm(rng, 3)
    arrays["lm_k3"] = table3
    cid = 0
    for T in (8, 64, 1024):
        for kind in ("flat", "peaky", "hard"):
            if T == 1024 and kind == "hard":
                continue
            mat = make_matrix(rng, T, kind, np.float32)
            name = f"w{cid}"
            cid += 1
            arrays[name] = mat
            for W in ((26, 40, 51, 64, 100) if T <= 64 else (26, 40, 51, 64)):
                seq, final = run_beam(mat, W, capture=T <= 64)
                c = {"group": "wide_nolm", "mat": name, "T": T, "kind": kind, "W": W, "seq": seq}
                if final is not None:
                    c["final"] = _final_json(final)
                cases.append(c)
    for T, kind in ((64, "flat"), (64, "peaky"), (300, "peaky")):
        mat = make_matrix(rng, T, kind, np.float64)
        name = f"w{cid}"
        cid += 1
        arrays[name] = mat
        for W in (26, 51, 64, 100):
            for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf)):
                seq, final = run_beam(mat, W, lm3, s_thr, r_thr, 3, capture=T <= 64)
                c = {"group": "wide_lm", "mat": name, "lm": "lm_k3", "k": 3, "T": T, "kind": kind, "W": W, "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                     "seq": seq}
                if final is not None:
                    c["final"] = _final_json(final)
                cases.append(c)

    # ---- (c) float32 single-window reads (N < chunk_len: the matrix basecall.py:99-109 decodes stays float32) with an LM
    for T, kind in ((300, "peaky"), (1000, "peaky"), (700, "flat")):
        mat = make_matrix(rng, T, kind, np.float32)
        name = f"s{cid}"
        cid += 1
        arrays[name] = mat
        ent = np.array([_entropy_legacy(mat[t]) for t in range(T)])
        ent_here = np.array([float(ref_decode.entropy(ref_decode.normalise(mat[t][:-1]))) for t in range(T)])
        for (kk, lm_obj, lm_name) in ((3, lm3, "lm_k3"), (k, None, "k11")):
            for W in (6, 10):
                for (s_thr, r_thr) in ((0.5, 0.5), (0.5, 0.9), (1.0, math.inf)):
                    margin = float(np.abs(ent - s_thr).min())
                    assert margin > 1e-5 and np.array_equal(ent > s_thr, ent_here > s_thr), "pick another seed: a row's entropy sits on the threshold"
                    model = lm_obj if lm_obj is not None else LazyLM(table, k)
                    seq, final = run_beam(mat, W, model, s_thr, r_thr, kk, capture=True)
                    cases.append({"group": "single_window_f32_lm", "mat": name, "lm": lm_name, "k": kk, "T": T, "kind": kind, "W": W,
                                  "s_thr": fenc(s_thr), "r_thr": fenc(r_thr), "entropy_margin": margin, "seq": seq,
                                  "winner_pr_total": fenc(final[0]["pr_total"])})
    np.savez_compressed(os.path.join(HERE, "beam_baseline_mats.npz"), **arrays)
    meta["source"] = ("radian/decode.py:79-96,100-212 + matrix_assembly.py:6-53 + basecall.py:35,96-109 at BASELINE.json's geometries; k = 11 model = "
                      "tests/_golden_lm.k11_table() served through a dict subclass with __missing__; group single_window_f32_lm holds only cases whose "
                      "per-row entropies stay `entropy_margin` away from s_thr, so numpy 1.19 (float64 entropies, the pinned version) and numpy 2.x "
                      "(float32) take the same gate decisions; their winner_pr_total is the reference's under the numpy named in `numpy`")
    meta["cases"] = cases
    with open(os.path.join(HERE, "beam_baseline_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)
    by = {}
    for c in cases:
        by[c["group"]] = by.get(c["group"], 0) + 1
    print("beam_baseline:", by, f"{time.time() - t_start:.0f}s")


# ----------------------------------------------------------------------------------------------
# 9. configs[2]'s chunk route at its own geometry, decode side (round 5): 4096 samples -> 8 windows of 1024 rows (step 512, last one
#    trimmed by the pad, basecall.py:96) -> per-window beam search W = 10 without an LM (basecall.py:110-121) -> simple_assembly + argmax
#    (basecall.py:122-123), on peaky and flat synthetic window probabilities; plus the reference's default step 128 (25 windows), W = 6
# ----------------------------------------------------------------------------------------------
def gen_pipeline_baseline():
    rng = np.random.default_rng(20261008)
    arrays, cases = {}, []
    for ci, (step, kind, W) in enumerate(((512, "peaky", 10), (512, "flat", 10), (128, "peaky", 6), (512, "peaky", 1))):
        windows, pad = ref_pre.get_windows(np.zeros(4096), 1024, step)
        nW = windows.shape[0]
        probs = np.stack([make_matrix(rng, 1024, kind, np.float32) for _ in range(nW)])
        arrays[f"probs{ci}"] = probs
        mats = [probs[i] for i in range(nW)]
        mats[-1] = mats[-1][:-pad]
        frags = [ref_decode.beam_search(m, BASES, W, None, None, None, None, None) for m in mats]
        cons = ref_seq.simple_assembly(frags)
        seq = ref_seq.index2base(np.argmax(cons, axis=0)) if cons.shape[1] else ""
        cases.append({"probs": f"probs{ci}", "chunk": 1024, "step": step, "N": 4096, "pad": int(pad), "nW": int(nW), "kind": kind, "W": W,
                      "chunk_fragments": frags, "chunk_seq": seq})
        print(f"  pipeline_baseline step={step} {kind} W={W}: {nW} windows, fragments of {min(map(len, frags))}..{max(map(len, frags))} bases, consensus {len(seq)}", flush=True)
    np.savez_compressed(os.path.join(HERE, "pipeline_baseline.npz"), **arrays)
    with open(os.path.join(HERE, "pipeline_baseline_cases.json"), "w") as f:
        json.dump({"source": "basecall.py:96,110-123 on synthetic window probabilities at BASELINE configs[2]'s geometry (sequence NOT yet reversed)",
                   "cases": cases}, f, indent=0)
    print("pipeline_baseline:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------
# 10. beam widths 65 ... 128 (round 6: the wave-per-sequence kernels' five-wave shape, the beam set in two halves): the reference's own
#     beam_search at W = 65 / 90 / 127 / 128 -- without an LM on float32 rows (chunk-mode semantics, final scores captured) and with a
#     3-mer LM on float64 rows.  A file of its own, so that the round-5 fixtures stay byte for byte what they were.
# ----------------------------------------------------------------------------------------------
def gen_beam_wide128():
    import time
    t_start = time.time()
    rng = np.random.default_rng(20261106)
    arrays, cases = {}, []
    lm3, table3 = make_lm(rng, 3)
    arrays["lm_k3"] = table3
    cid = 0
    for T in (8, 64, 300):
        for kind in ("flat", "peaky", "hard"):
            if T == 300 and kind == "hard":
                continue
            mat = make_matrix(rng, T, kind, np.float32)
            name = f"x{cid}"
            cid += 1
            arrays[name] = mat
            for W in (65, 90, 127, 128):
                seq, final = run_beam(mat, W, capture=True)
                cases.append({"group": "wide128_nolm", "mat": name, "T": T, "kind": kind, "W": W, "seq": seq, "final": _final_json(final[:8])})
        print(f"  wide128 no LM T={T}: {time.time() - t_start:.0f}s", flush=True)
    for T, kind in ((64, "flat"), (64, "peaky"), (200, "peaky")):
        mat = make_matrix(rng, T, kind, np.float64)
        name = f"x{cid}"
        cid += 1
        arrays[name] = mat
        for W in (65, 100, 128):
            for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf)):
                seq, final = run_beam(mat, W, lm3, s_thr, r_thr, 3, capture=True)
                cases.append({"group": "wide128_lm", "mat": name, "lm": "lm_k3", "k": 3, "T": T, "kind": kind, "W": W, "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                              "seq": seq, "final": _final_json(final[:8])})
        print(f"  wide128 LM T={T} {kind}: {time.time() - t_start:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "beam_wide128_mats.npz"), **arrays)
    with open(os.path.join(HERE, "beam_wide128_cases.json"), "w") as f:
        json.dump({"source": "radian/decode.py:100-212 (beam_search) at beam widths 65 / 90 / 100 / 127 / 128 on seeded synthetic matrices; final = the first entries of the "
                             "reference's last sorted beam list (labeling, pr_total, pr_blank, pr_non_blank)", "numpy": np.__version__, "cases": cases}, f, indent=0)
    print("beam_wide128:", len(cases), "cases", f"{time.time() - t_start:.0f}s")


# ----------------------------------------------------------------------------------------------
# 11. beam widths 129 ... 256 (round 6, last hours: the ten-wave shape, the beam set in four parts)
# ----------------------------------------------------------------------------------------------
def gen_beam_wide256():
    import time
    t_start = time.time()
    rng = np.random.default_rng(20261107)
    arrays, cases = {}, []
    lm3, table3 = make_lm(rng, 3)
    arrays["lm_k3"] = table3
    cid = 0
    for T in (8, 64, 200):
        for kind in ("flat", "peaky", "hard"):
            if T == 200 and kind == "hard":
                continue
            mat = make_matrix(rng, T, kind, np.float32)
            name = f"y{cid}"
            cid += 1
            arrays[name] = mat
            for W in (129, 200, 255, 256):
                seq, final = run_beam(mat, W, capture=True)
                cases.append({"group": "wide256_nolm", "mat": name, "T": T, "kind": kind, "W": W, "seq": seq, "final": _final_json(final[:8])})
        print(f"  wide256 no LM T={T}: {time.time() - t_start:.0f}s", flush=True)
    for T, kind in ((64, "flat"), (150, "peaky")):
        mat = make_matrix(rng, T, kind, np.float64)
        name = f"y{cid}"
        cid += 1
        arrays[name] = mat
        for W in (129, 256):
            for (s_thr, r_thr) in ((0.5, 0.5), (0.0, math.inf)):
                seq, final = run_beam(mat, W, lm3, s_thr, r_thr, 3, capture=True)
                cases.append({"group": "wide256_lm", "mat": name, "lm": "lm_k3", "k": 3, "T": T, "kind": kind, "W": W, "s_thr": fenc(s_thr), "r_thr": fenc(r_thr),
                              "seq": seq, "final": _final_json(final[:8])})
        print(f"  wide256 LM T={T} {kind}: {time.time() - t_start:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "beam_wide256_mats.npz"), **arrays)
    with open(os.path.join(HERE, "beam_wide256_cases.json"), "w") as f:
        json.dump({"source": "radian/decode.py:100-212 (beam_search) at beam widths 129 / 200 / 255 / 256 on seeded synthetic matrices; final = the first entries of the "
                             "reference's last sorted beam list", "numpy": np.__version__, "cases": cases}, f, indent=0)
    print("beam_wide256:", len(cases), "cases", f"{time.time() - t_start:.0f}s")


if __name__ == "__main__":
    if len(sys.argv) > 1:       # regenerate single fixture sets: make_golden.py gen_seq_assembly_random ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    gen_beam_nolm()
    gen_beam_lm()
    gen_beam_lm_sparse()
    gen_assemble()
    gen_preprocess()
    gen_seq_assembly()
    gen_seq_assembly_random()
    gen_fast5_signals()
    gen_pipeline()
    gen_beam_baseline()
    gen_pipeline_baseline()
    gen_beam_wide128()
    gen_beam_wide256()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("total bytes in tests/golden:", tot)
