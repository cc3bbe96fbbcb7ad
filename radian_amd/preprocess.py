"""Host-side step before the hot path: MAD normalisation and windowing.

Mirrors radian/preprocess.py:4-49 (same names, argument meaning and error behaviour), vectorised
with NumPy instead of np.vectorize's per-sample Python call."""
import numpy as np


def get_windows(signal, window_size, step_size):
    """radian/preprocess.py:4-22.  Returns (windows [nW, window_size], pad_end); the zero-padded last
    window is always appended (pad_end >= 1)."""
    if step_size <= 0:
        raise ValueError("Step size must be > 0")
    if step_size > window_size:
        raise ValueError("Step size must be <= window size")
    signal = np.asarray(signal)
    n = signal.shape[0]
    n_full = 0 if n < window_size else (n - window_size) // step_size + 1
    start = n_full * step_size
    pad_end = window_size - (n - start)
    out = np.zeros((n_full + 1, window_size), dtype=signal.dtype)
    if n_full:
        idx = np.arange(n_full)[:, None] * step_size + np.arange(window_size)[None, :]
        out[:n_full] = signal[idx]
    out[n_full, : n - start] = signal[start:]
    return out, pad_end


def mad_normalise(signal, outlier_z_score):
    """radian/preprocess.py:24-49: modified z-score (x - median) / (1.4826 * MAD), clipped to +-outlier_z_score.
    Raises ValueError on an empty signal or MAD == 0 (basecall.py:77-82 then skips the read).
    Reproduces the reference's np.vectorize dtype inference: if the FIRST sample is clipped the
    python-int clip value makes the whole result int64 (values truncated toward zero)."""
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
        z = np.trunc(z)
        z[hi] = outlier_z_score
        z[lo] = -1 * outlier_z_score
        return z.astype(np.int64)
    z = np.asarray(z, dtype=np.float64)
    z[hi] = outlier_z_score
    z[lo] = -1 * outlier_z_score
    return z
