"""RNA k-mer language model: JSON -> table for rd_load_lm.

The reference loads `{ "ACGT..."(k chars): [pA, pC, pG, pT] }` and re-keys it by tuples of label indices
(radian/basecall.py:48-57); decode.py:83 looks a context up by its tuple.  The table is indexed by the base-4 number of
the context, first (oldest) character most significant.

A model that does not hold every one of the 4^k contexts (a sparse JSON) loads like in the reference: the rows of absent
contexts are NaN, which rd_load_lm turns into the "absent" mask the beam search checks; a read whose search reaches one
fails with the reference's KeyError (decode.py:83) -- and only such a read.

The real model has 4^11 = 4 194 304 keys (~420 MB of text).  load_json first hands the file to the library's one-pass reader
(csrc/lmjson.hip: the one shape such a file has, straight into the table); whatever that reader does not recognise -- and every
dict handed in by a caller -- goes through the standard parser and table_from_dict, which converts the keys to table rows in one
vectorised pass."""
import json

import numpy as np

_IDX = {"A": 0, "C": 1, "G": 2, "T": 3}
_CODE = np.full(256, 255, dtype=np.uint8)
for _c, _i in _IDX.items():
    _CODE[ord(_c)] = _i


def context_index(context):
    """context: str over ACGT or sequence of label ints."""
    i = 0
    for c in context:
        i = (i << 2) | (_IDX[c] if isinstance(c, str) else int(c))
    return i


def _indices_of_str_keys(keys, k):
    """base-4 row numbers of n k-character keys in one pass over their bytes"""
    raw = "".join(keys).encode("ascii")                       # (a non-ASCII key raises here: not an ACGT string)
    if len(raw) != len(keys) * k:
        raise ValueError("RNA model contexts have mixed lengths")
    codes = _CODE[np.frombuffer(raw, dtype=np.uint8)].reshape(len(keys), k)
    if (codes == 255).any():
        bad = keys[int(np.argmax((codes == 255).any(axis=1)))]
        raise ValueError(f"{bad!r} is not in list")            # bases.index(b) at basecall.py:56 raises ValueError on a non-ACGT character
    weights = (np.int64(4) ** np.arange(k - 1, -1, -1, dtype=np.int64))
    return codes.astype(np.int64) @ weights


def table_from_dict(model):
    """dict {context str | tuple of label ints: [4 probs]} -> (table float64 [4^k,4], k).  Contexts the dict does not hold
    become rows of NaN (sparse model: module docstring).  Keys must share one length."""
    if not model:
        raise ValueError("empty RNA model")
    keys = list(model.keys())
    k = len(keys[0])
    if not (1 <= k <= 13):
        raise ValueError(f"RNA model contexts of {k} labels: a table of 4^{k} rows is out of range (1..13)")
    n = 4 ** k
    if isinstance(keys[0], str):
        if any(len(c) != k for c in keys):
            raise ValueError("RNA model contexts have mixed lengths")
        idx = _indices_of_str_keys(keys, k)
    else:
        arr = np.asarray(keys, dtype=np.int64)
        if arr.ndim != 2 or arr.shape[1] != k:
            raise ValueError("RNA model contexts have mixed lengths")
        if ((arr < 0) | (arr > 3)).any():
            raise ValueError("RNA model context labels must be 0..3")
        idx = arr @ (np.int64(4) ** np.arange(k - 1, -1, -1, dtype=np.int64))
    try:
        vals = np.asarray(list(model.values()), dtype=np.float64)
    except ValueError as e:
        raise ValueError("RNA model distributions must be four probabilities each") from e
    if vals.shape != (len(keys), 4):
        raise ValueError("RNA model distributions must be four probabilities each")
    if np.isnan(vals).any():
        raise ValueError("RNA model holds NaN probabilities")
    table = np.full((n, 4), np.nan, dtype=np.float64) if len(keys) < n else np.empty((n, 4), dtype=np.float64)
    table[idx] = vals          # (JSON object keys are unique after json.load: a repeated key keeps its last value, as in the reference)
    return table, k


def n_missing(table):
    """contexts a (sparse) table does not hold"""
    return int(np.isnan(table[:, 0]).sum())


def _load_json_native(path):
    """The file scanned straight into the table by the library's reader (rd_lm_json_probe / rd_lm_json_fill: no Python object per
    key or number) -> (table, k), or None when the text is not of the one shape that reader handles -- the caller then uses
    json.load, whose errors are the reference's."""
    import ctypes
    import mmap
    from . import _lib
    L = _lib.load()
    with open(path, "rb") as f:
        size = f.seek(0, 2)
        if size == 0:
            return None
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            view = np.frombuffer(mm, dtype=np.uint8)          # (zero-copy: a pointer into the mapping)
            try:
                ptr = ctypes.c_void_p(view.ctypes.data)
                k = ctypes.c_int(0)
                if L.rd_lm_json_probe(ptr, size, ctypes.byref(k)) != 0:
                    return None
                table = np.full((4 ** k.value, 4), np.nan, dtype=np.float64)
                n_entries, n_contexts = ctypes.c_int64(0), ctypes.c_int64(0)
                rc = L.rd_lm_json_fill(ptr, size, k.value, table.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n_entries), ctypes.byref(n_contexts))
            finally:
                del view                                        # (the mapping cannot close under an exported buffer)
    if rc != 0 or n_entries.value == 0:
        return None
    return table, k.value


def load_json(path, native=True):
    """basecall.py:48-57.  Returns (table, k).  native: try the library's one-pass reader first (the 4^11-key model: ~2 s instead of
    ~25 s, no 2-GB object tree); any text it does not recognise goes through json.load + table_from_dict."""
    if native:
        try:
            got = _load_json_native(path)
        except (OSError, ValueError, BufferError, MemoryError):     # (a file that cannot be mapped, a table that does not fit: the standard route reports)
            got = None
        if got is not None:
            return got
    with open(path, "r") as f:
        raw = json.load(f)
    return table_from_dict(raw)
