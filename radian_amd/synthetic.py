"""Seeded synthetic inputs of BASELINE.json's configs: Gaussian int16 reads."""
import numpy as np

from .preprocess import mad_normalise, get_windows


def synthetic_reads(n_reads, n_samples=4096, seed=0):
    """int16 = round(N(mu=500, sigma=80))."""
    rng = np.random.default_rng(seed)
    return np.round(rng.normal(500.0, 80.0, size=(n_reads, n_samples))).astype(np.int16)


def reads_to_windows(reads, chunk_len=1024, step=512, clip=4):
    """MAD-normalise + window every read (basecall.py:78,83).  Returns
    (windows float32 [nW_total, chunk], valid_len int32 [nW_total], read_win_off int32 [n_reads+1], pads int32 [n_reads])."""
    wins, valid, offs, pads = [], [], [0], []
    for r in reads:
        norm = mad_normalise(r, clip)
        w, pad = get_windows(norm, chunk_len, step)
        v = np.full(w.shape[0], chunk_len, dtype=np.int32)
        v[-1] = chunk_len - pad
        wins.append(w.astype(np.float32))
        valid.append(v)
        offs.append(offs[-1] + w.shape[0])
        pads.append(pad)
    return (np.concatenate(wins, axis=0), np.concatenate(valid), np.asarray(offs, dtype=np.int32),
            np.asarray(pads, dtype=np.int32))


def peaky_probs(n_windows, chunk_len=1024, seed=0, gain=4.0, blank_bias=2.0):
    """SURVEY.md section 8d's decode-only benchmark rows: softmax(gain * N(0,1) + blank_bias on the blank class), float32
    [n_windows, chunk_len, 5] -- peaked rows with the blank favoured, as a trained CTC model emits them (a base every few rows), unlike
    the random-weight model's saturated rows (~5 bases per window) or the soft head's (~200)."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n_windows, chunk_len, 5), dtype=np.float32) * np.float32(gain)
    z[..., 4] += np.float32(blank_bias)
    z -= z.max(axis=-1, keepdims=True)
    np.exp(z, out=z)
    z /= z.sum(axis=-1, keepdims=True)
    return z
