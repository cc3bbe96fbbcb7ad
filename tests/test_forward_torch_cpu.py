"""The oracle's TCN forward (oracle/radian_oracle.c ro_tcn_forward; model.py:52-89 + keras-tcn 3.5's residual block) against an
independent statement of the same graph in stock PyTorch operators on the CPU, float64: `F.conv1d` with `dilation=` after a causal
left pad of (k-1)d zeros (Keras Conv1D(padding="causal") and torch conv1d are both cross-correlations), ReLU, the 1x1 matching conv
of block 0, Dense/ReLU/Dense/softmax.  The forward is the one part of the path no reference fixture pins (keras-tcn / TensorFlow are
absent); this guards the restatement's conv indexing, padding, tensor layouts and block wiring with a second, widely used
implementation.  CPU only: PyTorch is test plumbing here, not the product."""
import numpy as np
import pytest


def _torch_forward(flat, x, dilations, C=256, K=3, H=128):
    import torch
    import torch.nn.functional as F
    from radian_amd import weights
    t = {}
    o = 0
    for name, shape in weights.tensor_shapes(dilations):
        n = int(np.prod(shape))
        t[name] = torch.from_numpy(flat[o:o + n].astype(np.float64).reshape(shape))
        o += n
    assert o == flat.size

    def causal(v, kernel, bias, d):            # v [B, C_in, T]; Keras kernel [k, C_in, C_out] -> torch [C_out, C_in, k]
        w = kernel.permute(2, 1, 0).contiguous()
        return F.conv1d(F.pad(v, ((kernel.shape[0] - 1) * d, 0)), w, bias, dilation=d)

    v = torch.from_numpy(x.astype(np.float64))[:, None, :]              # [B, 1, T]
    for b, d in enumerate(dilations):
        p = f"tcn/residual_block_{b}/"
        x1 = torch.relu(causal(v, t[p + "conv1D_0/kernel"], t[p + "conv1D_0/bias"], d))
        x1 = torch.relu(causal(x1, t[p + "conv1D_1/kernel"], t[p + "conv1D_1/bias"], d))
        res = causal(v, t[p + "matching_conv1D/kernel"], t[p + "matching_conv1D/bias"], 1) if b == 0 else v
        v = torch.relu(res + x1)
    h = torch.relu(v.transpose(1, 2) @ t["dense/kernel"] + t["dense/bias"])      # [B, T, H]
    z = h @ t["dense_1/kernel"] + t["dense_1/bias"]
    return torch.softmax(z, dim=-1).numpy()


@pytest.mark.parametrize("dilations", [(1, 2, 4, 8, 16, 32), (1, 3), (4,)])
def test_oracle_forward_matches_torch_operators(oracle, dilations):
    pytest.importorskip("torch")
    from radian_amd import weights
    rng = np.random.default_rng(5)
    flat = weights.synthetic_weights(seed=77, dilations=dilations, head_gain=0.3)
    x = rng.normal(size=(3, 150)).astype(np.float32)
    x[1, :40] = 0.0                                   # a window that starts with padding-like zeros
    ref = _torch_forward(flat, x, dilations)
    got64 = oracle.tcn_forward(flat, x, dilations=dilations, acc64=True)
    got32 = oracle.tcn_forward(flat, x, dilations=dilations)
    assert got64.shape == ref.shape == (3, 150, 5)
    assert np.abs(got64 - ref).max() <= 2e-6, np.abs(got64 - ref).max()      # float32 activations between layers vs float64 throughout
    assert np.abs(got32 - ref).max() <= 1e-5, np.abs(got32 - ref).max()
    assert np.abs(ref.sum(axis=2) - 1.0).max() < 1e-12
