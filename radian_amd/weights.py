"""Weight blob (de)serialisation for rd_load_weights, and seeded synthetic weights.

The blob is an rd_weights_header (include/radian_hip.h) followed by float32 tensors in the order
and layouts Keras' model.load_weights sees for the graph of radian/model.py:52-89
(SURVEY.md section 8c lists the expected tensor names/shapes).
"""
import struct

import numpy as np

MAGIC = 0x574E4452  # 'RDNW'
C, K, H, NCLS = 256, 3, 128, 5
DEFAULT_DILATIONS = (1, 2, 4, 8, 16, 32)  # radian/models/sig2seq.yaml:42 (nb_stacks 1)


def tensor_shapes(dilations=DEFAULT_DILATIONS):
    """[(name, shape)] in load_weights order."""
    shapes = []
    for b in range(len(dilations)):
        cin = 1 if b == 0 else C
        shapes.append((f"tcn/residual_block_{b}/conv1D_0/kernel", (K, cin, C)))
        shapes.append((f"tcn/residual_block_{b}/conv1D_0/bias", (C,)))
        shapes.append((f"tcn/residual_block_{b}/conv1D_1/kernel", (K, C, C)))
        shapes.append((f"tcn/residual_block_{b}/conv1D_1/bias", (C,)))
        if b == 0:
            shapes.append((f"tcn/residual_block_{b}/matching_conv1D/kernel", (1, 1, C)))
            shapes.append((f"tcn/residual_block_{b}/matching_conv1D/bias", (C,)))
    shapes.append(("dense/kernel", (C, H)))
    shapes.append(("dense/bias", (H,)))
    shapes.append(("dense_1/kernel", (H, NCLS)))
    shapes.append(("dense_1/bias", (NCLS,)))
    return shapes


def n_params(dilations=DEFAULT_DILATIONS):
    return int(sum(int(np.prod(s)) for _, s in tensor_shapes(dilations)))


def synthetic_weights(seed=1234, dilations=DEFAULT_DILATIONS, bias_scale=0.05, head_gain=1.0):
    """He-normal kernels (sig2seq.yaml:48 kernel_initializer; Keras he_normal is a truncated normal with
    stddev sqrt(2/fan_in) -- a plain normal is used here), small random biases.  Returns the flat float32
    array in load_weights order.  `head_gain` scales the last Dense kernel to make the softmax peakier."""
    rng = np.random.default_rng(seed)
    parts = []
    for name, shape in tensor_shapes(dilations):
        if name.endswith("kernel"):
            fan_in = int(np.prod(shape[:-1]))
            w = rng.normal(0.0, np.sqrt(2.0 / fan_in), size=shape)
            if name == "dense_1/kernel":
                w = w * head_gain
        else:
            w = rng.normal(0.0, bias_scale, size=shape)
        parts.append(w.astype(np.float32).ravel())
    return np.concatenate(parts)


def pack_blob(flat, dilations=DEFAULT_DILATIONS):
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    if flat.size != n_params(dilations):
        raise ValueError(f"expected {n_params(dilations)} parameters, got {flat.size}")
    dil = list(dilations) + [0] * (16 - len(dilations))
    header = struct.pack("<7I16II", MAGIC, 1, C, K, H, NCLS, len(dilations), *dil, flat.size)
    return header + flat.tobytes()


def unpack_blob(blob):
    hs = struct.calcsize("<7I16II")
    f = struct.unpack("<7I16II", blob[:hs])
    if f[0] != MAGIC:
        raise ValueError("bad weight blob magic")
    nb = f[6]
    dil = tuple(f[7:7 + nb])
    flat = np.frombuffer(blob, dtype=np.float32, offset=hs, count=f[23])
    return flat, dil
