"""RNA k-mer language model: JSON -> dense table for rd_load_lm.

The reference loads `{ "ACGT..."(k chars): [pA, pC, pG, pT] }` and re-keys it by tuples of label indices
(radian/basecall.py:48-57); decode.py:83 looks a context up by its tuple.  The dense table is indexed by
the base-4 number of the context, first (oldest) character most significant."""
import json

import numpy as np

_IDX = {"A": 0, "C": 1, "G": 2, "T": 3}


def context_index(context):
    """context: str over ACGT or sequence of label ints."""
    i = 0
    for c in context:
        i = (i << 2) | (_IDX[c] if isinstance(c, str) else int(c))
    return i


def table_from_dict(model):
    """dict {context str | tuple: [4 probs]} -> (table float64 [4^k,4], k).  Every context must be present
    (the reference raises KeyError on a missing one, decode.py:83; a dense table cannot represent that)."""
    if not model:
        raise ValueError("empty RNA model")
    k = len(next(iter(model)))
    n = 4 ** k
    if len(model) != n:
        raise ValueError(f"RNA model has {len(model)} contexts of length {k}; a dense table needs all {n}")
    table = np.empty((n, 4), dtype=np.float64)
    seen = np.zeros(n, dtype=bool)
    for ctx, dist in model.items():
        if len(ctx) != k:
            raise ValueError("RNA model contexts have mixed lengths")
        i = context_index(ctx)
        table[i] = dist
        seen[i] = True
    if not seen.all():
        raise ValueError("RNA model has duplicate / missing contexts")
    return table, k


def load_json(path):
    """basecall.py:48-57.  Returns (table, k)."""
    with open(path, "r") as f:
        raw = json.load(f)
    return table_from_dict(raw)
