"""Executes the ctypes binding printed in INTEGRATION.md (the stub a RADIAN maintainer pastes into basecall.py) against the
library and checks its three functions against radian_amd.Backend."""
import os
import re
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs():
    from radian_amd import Backend, weights, lib_path
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", md, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libradian_hip.so")', f'ctypes.CDLL({lib_path()!r})')
    rng = np.random.default_rng(0)
    k = 3
    table = np.ascontiguousarray(rng.dirichlet([0.3] * 4, size=4 ** k))
    flat = weights.synthetic_weights(seed=1234)
    ns = {"blob": weights.pack_blob(flat), "table": table, "args": types.SimpleNamespace(context_len=k)}
    exec(code, ns)
    be = Backend(0)
    be.load_weights(flat)
    be.load_lm(table, k)
    x = rng.normal(size=(3, 1024)).astype(np.float32)
    probs = ns["sig_model"].predict(x)
    assert np.array_equal(probs, be.forward(x))
    mats = [probs[0], probs[1], probs[2][:-300]]
    m = ns["assemble_matrices"](mats, 512)
    assert m.dtype == np.float64 and np.array_equal(m, be.assemble(probs, 300, 512))
    s = ns["beam_search"](m, "ACGT", 6, {"lm": 1}, 0.5, 0.5, k, {})
    assert s == "".join("ACGT"[c] for c in be.decode(m, 6, use_lm=True, s_threshold=0.5, r_threshold=0.5))
    s2 = ns["beam_search"](probs[0], "ACGT", 6, None, None, None, None, None)
    assert s2 == "".join("ACGT"[c] for c in be.decode(probs[0], 6))
    be.close()
