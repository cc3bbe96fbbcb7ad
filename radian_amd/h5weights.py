"""Keras weights-only HDF5 (`model.save_weights("*.h5")`, radian/train.py:72-78) -> flat float32 parameters
in load_weights order (radian/model.py:44) for rd_load_weights.

Layout written by Keras 2.4: root attribute `layer_names`; per layer a group with attribute `weight_names`
and one dataset per weight at `/<layer>/<weight_name>`.  load_weights is positional, so tensors are taken
in file order and checked against the shapes the graph of model.py:52-89 implies
(radian_amd.weights.tensor_shapes)."""
import numpy as np

from . import h5
from .weights import tensor_shapes, DEFAULT_DILATIONS


def _str_list(value):
    """attribute value -> list of str: scalar or array, str or bytes (fixed 'S' arrays, variable-length strings)"""
    if value is None:
        return None
    if isinstance(value, (str, bytes)):
        value = [value]
    return [v.decode("utf-8") if isinstance(v, bytes) else str(v) for v in value]


def _names_attr(f, obj, name):
    """Keras' save_attributes_to_hdf5_group: one attribute `name`, or -- when the list would exceed HDF5's 64 KiB
    object-header limit -- chunks `name0`, `name1`, ... (hdf5_format.py in Keras 2.4)."""
    vals = _str_list(f.attr(obj, name))
    if vals is not None:
        return vals
    out, i = [], 0
    while True:
        part = _str_list(f.attr(obj, f"{name}{i}"))
        if part is None:
            break
        out += part
        i += 1
    return out if i else None


def read_keras_weights(path, dilations=DEFAULT_DILATIONS):
    """Accepts a weights-only file (`save_weights`, the reference's checkpoints: layer groups at the root) and a
    full-model file (`model.save`: the same tree under /model_weights)."""
    expected = tensor_shapes(dilations)
    tensors = []
    with h5.File(path, "r") as f:
        root = "/"
        layers = _names_attr(f, root, "layer_names")
        if layers is None and f.exists("/model_weights"):
            root = "/model_weights"
            layers = _names_attr(f, root, "layer_names")
        if layers is None:
            raise ValueError(f"{path}: no `layer_names` attribute - not a Keras weights file")
        base = root.rstrip("/")
        for layer in layers:
            names = _names_attr(f, f"{base}/{layer}", "weight_names") or []
            for wn in names:
                tensors.append((f"{layer}/{wn}", f.read(f"{base}/{layer}/{wn}")))
    if len(tensors) != len(expected):
        raise ValueError(f"{path}: {len(tensors)} weight tensors, the Sig2Seq graph has {len(expected)}: "
                         + ", ".join(n for n, _ in tensors[:6]) + " ...")
    parts = []
    for (name, arr), (ename, eshape) in zip(tensors, expected):
        if tuple(arr.shape) != tuple(eshape):
            raise ValueError(f"{path}: tensor {name} has shape {arr.shape}, expected {eshape} for {ename}")
        parts.append(np.ascontiguousarray(arr, dtype=np.float32).ravel())
    return np.concatenate(parts)


def write_keras_weights(path, flat, dilations=DEFAULT_DILATIONS, attr_kind="nullpad", root="/", chunk_names=0):
    """Inverse of read_keras_weights (fixtures / exporting synthetic weights in the reference's file format).
    attr_kind: HDF5 form of the name lists ("nullpad" = h5py's NumPy-'S'-array form that Keras 2.4 writes, "vlen" =
    variable-length strings, "nullterm"); root "/model_weights" gives the full-model layout; chunk_names > 0 splits
    `weight_names` into `weight_names0..` of at most that many entries (Keras' form for very long lists)."""
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    by_layer = {}
    off = 0
    for name, shape in tensor_shapes(dilations):
        n = int(np.prod(shape))
        layer = name.split("/")[0]
        by_layer.setdefault(layer, []).append((name + ":0", flat[off:off + n].reshape(shape)))
        off += n
    layer_names = ["inputs", "tcn", "dense", "activation", "dense_1", "activation_1"]
    base = root.rstrip("/")
    with h5.File(path, "w") as f:
        if base:
            f.create_group(base)
        for layer in layer_names:
            f.create_group(f"{base}/{layer}")
            ws = by_layer.get(layer, [])
            for wn, arr in ws:
                f.write(f"{base}/{layer}/{wn}", arr)
            names = [wn for wn, _ in ws]
            if names and chunk_names > 0:
                for i in range(0, len(names), chunk_names):
                    f.set_attr_str(f"{base}/{layer}", f"weight_names{i // chunk_names}", names[i:i + chunk_names], kind=attr_kind)
            elif names:
                f.set_attr_str(f"{base}/{layer}", "weight_names", names, kind=attr_kind)
        f.set_attr_str(base or "/", "layer_names", layer_names, kind=attr_kind)
        f.set_attr_str(base or "/", "backend", "tensorflow", kind=attr_kind)
        f.set_attr_str(base or "/", "keras_version", "2.4.0", kind=attr_kind)
