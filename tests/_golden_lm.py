"""The seeded 12-mer RNA-model table (context 11 -> 4^11 x 4 float64, 128 MiB) that the BASELINE-geometry golden cases were
generated with.  Too large to commit, so the generator (tests/golden/make_golden.py gen_beam_baseline) and the tests rebuild it
from the seed and check its SHA-256 against the one the fixture records: a numpy whose Dirichlet stream differs fails loudly
instead of comparing against the wrong table."""
import hashlib

import numpy as np

K11 = 11
K11_SEED = 20261005
_cache = {}


def k11_table():
    """[4^11, 4] float64, Dirichlet(0.3) rows (cfg 4), index = base-4 number of the context, first label most
    significant."""
    if "t" not in _cache:
        rng = np.random.default_rng(K11_SEED)
        _cache["t"] = np.ascontiguousarray(rng.dirichlet([0.3] * 4, size=4 ** K11).astype(np.float64))
    return _cache["t"]


def table_sha256(table):
    return hashlib.sha256(np.ascontiguousarray(table, dtype=np.float64).tobytes()).hexdigest()


def checked_k11_table(expected_sha256):
    t = k11_table()
    got = table_sha256(t)
    assert got == expected_sha256, ("this numpy's default_rng(...).dirichlet stream differs from the one the k = 11 golden cases were "
                                    f"generated with (table sha256 {got} != {expected_sha256}); regenerate tests/golden/beam_baseline_*")
    return t


def sparse_table(table, missing):
    """The dense form of an RNA-model dict that lacks the contexts in `missing` (row indices): those rows are NaN (lm.py / the
    oracle's convention for an absent context)."""
    t = table.copy()
    t[np.asarray(missing, dtype=np.int64)] = np.nan
    return t
