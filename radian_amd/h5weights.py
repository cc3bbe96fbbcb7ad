"""Keras weights-only HDF5 (`model.save_weights("*.h5")`, radian/train.py:72-78) -> flat float32 parameters
in load_weights order (radian/model.py:44) for rd_load_weights.

Layout written by Keras 2.4: root attribute `layer_names`; per layer a group with attribute `weight_names`
and one dataset per weight at `/<layer>/<weight_name>`.  load_weights is positional, so tensors are taken
in file order and checked against the shapes the graph of model.py:52-89 implies
(radian_amd.weights.tensor_shapes)."""
import numpy as np

from . import h5
from .weights import tensor_shapes, DEFAULT_DILATIONS


def read_keras_weights(path, dilations=DEFAULT_DILATIONS):
    expected = tensor_shapes(dilations)
    tensors = []
    with h5.File(path, "r") as f:
        layers = f.attr("/", "layer_names")
        if layers is None:
            raise ValueError(f"{path}: no `layer_names` attribute - not a Keras weights file")
        if isinstance(layers, str):
            layers = [layers]
        for layer in layers:
            names = f.attr("/" + layer, "weight_names", default=[])
            if isinstance(names, str):
                names = [names]
            for wn in names:
                tensors.append((f"{layer}/{wn}", f.read(f"/{layer}/{wn}")))
    if len(tensors) != len(expected):
        raise ValueError(f"{path}: {len(tensors)} weight tensors, the Sig2Seq graph has {len(expected)}: "
                         + ", ".join(n for n, _ in tensors[:6]) + " ...")
    parts = []
    for (name, arr), (ename, eshape) in zip(tensors, expected):
        if tuple(arr.shape) != tuple(eshape):
            raise ValueError(f"{path}: tensor {name} has shape {arr.shape}, expected {eshape} for {ename}")
        parts.append(np.ascontiguousarray(arr, dtype=np.float32).ravel())
    return np.concatenate(parts)


def write_keras_weights(path, flat, dilations=DEFAULT_DILATIONS):
    """Inverse of read_keras_weights (fixtures / exporting synthetic weights in the reference's file format)."""
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    by_layer = {}
    off = 0
    for name, shape in tensor_shapes(dilations):
        n = int(np.prod(shape))
        layer = name.split("/")[0]
        by_layer.setdefault(layer, []).append((name + ":0", flat[off:off + n].reshape(shape)))
        off += n
    layer_names = ["inputs", "tcn", "dense", "activation", "dense_1", "activation_1"]
    with h5.File(path, "w") as f:
        for layer in layer_names:
            f.create_group("/" + layer)
            ws = by_layer.get(layer, [])
            for wn, arr in ws:
                f.write(f"/{layer}/{wn}", arr)
            if ws:
                f.set_attr_str("/" + layer, "weight_names", [wn for wn, _ in ws])
        f.set_attr_str("/", "layer_names", layer_names)
        f.set_attr_str("/", "backend", "tensorflow")
        f.set_attr_str("/", "keras_version", "2.4.0")
