"""CPU oracle: Python face of oracle/radian_oracle.c plus NumPy restatements of the host steps.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module, and only as the checker -- never as the thing measured or shipped.  The
product package (radian_amd/) must not import it and has no CPU fallback.

Each function cites the reference file:line it follows (paths under /root/reference/).
Parity status: decode / LM gate / assembly / preprocess / chunk stitch are pinned by the golden
fixtures in tests/golden (generated from the reference's own Python by make_golden.py);
the TCN forward is "parity unpinned" (third-party keras-tcn/TensorFlow arithmetic, weights absent).
"""
import ctypes
import difflib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libradian_oracle.so")
_lib = None

BASES = "ACGT"


def build(force=False):
    """Compile radian_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "radian_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.ro_logaddexp.restype = ctypes.c_double
        L.ro_logaddexp.argtypes = [ctypes.c_double, ctypes.c_double]
        L.ro_row_entropy.restype = ctypes.c_double
        L.ro_row_entropy.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ro_assemble.restype = ctypes.c_int64
        L.ro_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------------------------
# decode.py
# ------------------------------------------------------------------------------------------------
def logaddexp(x, y):
    return lib().ro_logaddexp(float(x), float(y))


def beam_search_labels(mat, beam_width, lm_table=None, s_threshold=0.0, r_threshold=0.0, len_context=0,
                       max_final=0):
    """decode.py:100-212.  mat [T,5] float32|float64; lm_table None or dense [4^k,4] float64.
    Returns (labels uint8[len], final) where final is None or a list of
    (labeling str, pr_total, pr_blank, pr_non_blank) for the sorted final BeamList."""
    mat = np.ascontiguousarray(mat)
    if mat.dtype not in (np.float32, np.float64):
        raise TypeError("mat must be float32 or float64")
    T = mat.shape[0]
    if mat.ndim != 2 or (T > 0 and mat.shape[1] != 5):
        raise ValueError("mat must be [T,5]")
    out = np.zeros(T + 1, dtype=np.uint8)
    out_len = ctypes.c_int(0)
    if lm_table is not None:
        lm_table = np.ascontiguousarray(lm_table, dtype=np.float64)
        assert lm_table.shape == (4 ** len_context, 4)
    nf = ctypes.c_int(0)
    fs = np.zeros(max(1, max_final) * 3, dtype=np.float64)
    fl = np.zeros(max(1, max_final), dtype=np.int32)
    fb = np.zeros(max(1, max_final) * (T + 1), dtype=np.uint8)
    rc = lib().ro_beam_search(
        _ptr(mat), ctypes.c_int(1 if mat.dtype == np.float64 else 0), ctypes.c_int(T), ctypes.c_int(beam_width),
        _ptr(lm_table), ctypes.c_int(len_context), ctypes.c_double(s_threshold), ctypes.c_double(r_threshold),
        _ptr(out), ctypes.byref(out_len), ctypes.c_int(max_final),
        ctypes.byref(nf) if max_final else None, _ptr(fs) if max_final else None,
        _ptr(fl) if max_final else None, _ptr(fb) if max_final else None)
    if rc != 0:
        raise RuntimeError("ro_beam_search failed")
    if out_len.value < 0:
        # sparse model (rows of NaN = absent contexts): the search looked one up -- the reference's KeyError, decode.py:83
        raise KeyError("the RNA model holds no entry for a context of the beam search (radian/decode.py:83)")
    labels = out[: out_len.value].copy()
    final = None
    if max_final:
        final = []
        off = 0
        for i in range(nf.value):
            n = int(fl[i])
            lab = "".join(BASES[c] for c in fb[off:off + n])
            off += n
            final.append((lab, float(fs[3 * i]), float(fs[3 * i + 1]), float(fs[3 * i + 2])))
    return labels, final


def beam_search(mat, bases, beam_width, lm_table=None, s_threshold=None, r_threshold=None, len_context=None):
    """Same call shape as decode.beam_search (decode.py:100-109) with the LM given as a dense table."""
    labels, _ = beam_search_labels(mat, beam_width, lm_table,
                                   0.0 if s_threshold is None else s_threshold,
                                   0.0 if r_threshold is None else r_threshold,
                                   0 if len_context is None else len_context)
    return "".join(bases[c] for c in labels)


def beam_search_batch(mats, seq_off, seq_len, beam_width, lm_table=None, s_threshold=0.0, r_threshold=0.0,
                      len_context=0, nthreads=0, hash_order=0):
    """Batch of independent sequences over concatenated rows (OpenMP over sequences).
    hash_order > 0: long-context synthetic LM (no reference behaviour; radian_oracle.c ro_ctx_hash): lm_table has
    4^hash_order rows addressed by a hash of the last len_context labels."""
    mats = np.ascontiguousarray(mats)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
    seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
    n = seq_len.shape[0]
    label_off = np.zeros(n, dtype=np.int64)
    if n:
        label_off[1:] = np.cumsum(seq_len[:-1].astype(np.int64) + 1)
    labels = np.zeros(int((seq_len.astype(np.int64) + 1).sum()) + 1, dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    if lm_table is not None:
        lm_table = np.ascontiguousarray(lm_table, dtype=np.float64)
    rc = lib().ro_beam_search_batch_ex(
        _ptr(mats), ctypes.c_int(1 if mats.dtype == np.float64 else 0), _ptr(seq_off), _ptr(seq_len), ctypes.c_int(n),
        ctypes.c_int(beam_width), _ptr(lm_table), ctypes.c_int(len_context), ctypes.c_int(hash_order), ctypes.c_double(s_threshold),
        ctypes.c_double(r_threshold), _ptr(labels), _ptr(label_off), _ptr(lens), ctypes.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("ro_beam_search_batch failed")
    # a sequence whose search looked up a context that a sparse table (rows of NaN) does not hold: None (the reference raises
    # KeyError at decode.py:83)
    return [labels[label_off[i]: label_off[i] + lens[i]].copy() if lens[i] >= 0 else None for i in range(n)]


def row_entropy(row):
    row = np.ascontiguousarray(row)
    return lib().ro_row_entropy(_ptr(row), ctypes.c_int(1 if row.dtype == np.float64 else 0))


def apply_rna_model(s_dist, ctx_index, lm_table, s_entropy, r_threshold, s_threshold):
    """decode.py:79-96 on a float64 row."""
    s = np.ascontiguousarray(s_dist, dtype=np.float64)
    lm_table = np.ascontiguousarray(lm_table, dtype=np.float64)
    out = np.zeros(5, dtype=np.float64)
    lib().ro_apply_rna_model_pub(_ptr(s), _ptr(lm_table), ctypes.c_int(ctx_index), ctypes.c_double(s_entropy),
                                 ctypes.c_double(r_threshold), ctypes.c_double(s_threshold), _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------
# matrix_assembly.py
# ------------------------------------------------------------------------------------------------
def assemble_matrices(probs, pad, step_size):
    """matrix_assembly.py:6-53 applied to window outputs probs [nW,chunk,5] float32 whose last window is
    trimmed by `pad` rows first (basecall.py:96).  Returns the [N,5] matrix with the reference's dtype."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    nW, chunk, _ = probs.shape
    cap = (nW - 1) * step_size + chunk
    out = np.zeros((max(cap, 1), 5), dtype=np.float64)
    isf64 = ctypes.c_int(0)
    N = lib().ro_assemble(_ptr(probs), ctypes.c_int(nW), ctypes.c_int(chunk), ctypes.c_int(pad), ctypes.c_int(step_size),
                          _ptr(out), ctypes.c_int64(out.shape[0]), ctypes.byref(isf64))
    if N < 0:
        raise RuntimeError("ro_assemble failed")
    out = out[:N]
    return out if isf64.value else out.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# preprocess.py
# ------------------------------------------------------------------------------------------------
def mad_normalise(signal, outlier_z_score):
    """preprocess.py:24-49.  np.vectorize without otypes takes the output dtype from the FIRST sample:
    if that sample is clipped the python-int clip value makes the whole result int64 (truncating)."""
    signal = np.asarray(signal)
    if signal.shape[0] == 0:
        raise ValueError("Signal must not be empty to normalise")
    median = np.median(signal)
    mad = np.median(np.abs(signal - median))
    if mad == 0:
        raise ValueError("MAD is zero, issue with signal.")
    z = (signal - median) / (1.4826 * mad)
    hi = z > outlier_z_score
    lo = z < -1 * outlier_z_score
    if hi[0] or lo[0]:
        res = np.trunc(z)
        res[hi] = outlier_z_score
        res[lo] = -1 * outlier_z_score
        return res.astype(np.int64)
    res = z.astype(np.float64)
    res[hi] = outlier_z_score
    res[lo] = -1 * outlier_z_score
    return res


def get_windows(signal, window_size, step_size):
    """preprocess.py:4-22."""
    if step_size <= 0:
        raise ValueError("Step size must be > 0")
    if step_size > window_size:
        raise ValueError("Step size must be <= window size")
    windows = []
    start = 0
    while start + window_size <= signal.shape[0]:
        windows.append(signal[start:start + window_size])
        start += step_size
    last = signal[start:]
    pad_end = window_size - len(last)
    windows.append(np.pad(last, (0, pad_end)))
    return np.asarray(windows), pad_end


# ------------------------------------------------------------------------------------------------
# sequence_assembly.py
# ------------------------------------------------------------------------------------------------
def simple_assembly(bpreads):
    """sequence_assembly.py:19-48 (difflib is the Python stdlib, as in the reference)."""
    census_len = 1000
    concensus = np.zeros([4, census_len])
    pos = 0
    length = 0
    idx = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}

    def add_count(start, segment):
        if start < 0:
            segment = segment[-start:]
            start = 0
        for i, base in enumerate(segment):
            concensus[idx[base]][start + i] += 1

    for i, bpread in enumerate(bpreads):
        if i == 0:
            add_count(0, bpread)
            continue
        d = difflib.SequenceMatcher(None, bpreads[i - 1], bpread)
        mb = max(d.get_matching_blocks(), key=lambda x: x[2])
        disp = mb[0] - mb[1]
        if disp + pos + len(bpread) > census_len:
            concensus = np.pad(concensus, ((0, 0), (0, 1000)), mode="constant", constant_values=0)
            census_len += 1000
        add_count(pos + disp, bpread)
        pos += disp
        length = max(length, pos + len(bpread))
    return concensus[:, :length]


def index2base(read):
    """sequence_assembly.py:90-97."""
    return "".join(BASES[x] for x in read)


def chunk_consensus(fragments):
    """basecall.py:122-123."""
    cons = simple_assembly(fragments)
    if cons.shape[1] == 0:
        return ""
    return index2base(np.argmax(cons, axis=0))


# ------------------------------------------------------------------------------------------------
# model.py (UNPINNED)
# ------------------------------------------------------------------------------------------------
def tcn_forward(weights, x, C=256, K=3, dilations=(1, 2, 4, 8, 16, 32), H=128, nthreads=0, acc64=False):
    """model.py:52-89 on windows x [B,T] -> probs [B,T,5] float32.  `weights` is the flat float32 array
    in Keras load_weights order (radian_oracle.c, ro_tcn_forward).  acc64=True accumulates every dot product in float64
    (the yardstick for float32 summation-order noise)."""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T = x.shape
    dil = np.ascontiguousarray(dilations, dtype=np.int32)
    probs = np.zeros((B, T, 5), dtype=np.float32)
    fn = lib().ro_tcn_forward_acc64 if acc64 else lib().ro_tcn_forward
    rc = fn(_ptr(w), ctypes.c_int(C), ctypes.c_int(K), ctypes.c_int(len(dil)), _ptr(dil), ctypes.c_int(H),
                              _ptr(x), ctypes.c_int(B), ctypes.c_int(T), _ptr(probs), ctypes.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("ro_tcn_forward failed")
    return probs


def num_threads():
    return lib().ro_num_threads()
