"""Keras weights-only HDF5 (`model.save_weights("*.h5")`, radian/train.py:72-78) -> flat float32 parameters
in load_weights order (radian/model.py:44) for rd_load_weights.

Layout written by Keras 2.4: root attribute `layer_names`; per layer a group with attribute `weight_names`
and one dataset per weight at `/<layer>/<weight_name>`.  load_weights is positional, so tensors are taken
in file order and checked against the shapes the graph of model.py:52-89 implies
(radian_amd.weights.tensor_shapes) and, when the names follow keras-tcn's scheme, against the names SURVEY.md 8c lists."""
import re

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


_BLOCK = re.compile(r"residual_block_(\d+)")
_CONV = re.compile(r"(?:^|/)(conv1D_(\d+)|matching_conv1D)(?:_\d+)?/")
_DENSE = re.compile(r"(?:^|/)dense(?:_(\d+))?/")
_KIND = re.compile(r"(kernel|bias)(?::\d+)?$")


def tensor_role(name):
    """What a Keras weight name says about its place in the Sig2Seq graph (model.py:52-89, keras-tcn 3.5's naming as SURVEY.md 8c
    lists it): ("conv", block, "conv1D_0" | "conv1D_1" | "matching_conv1D", "kernel" | "bias"), ("dense", None, None, kind) for either
    Dense layer (Keras numbers them per session -- dense_2, dense_3 ... -- so the two are told apart by shape, which differs), or None
    when the name does not follow that scheme (a renamed layer: nothing to check)."""
    kind = _KIND.search(name)
    if kind is None:
        return None
    b, c = _BLOCK.search(name), _CONV.search(name)
    if b is not None and c is not None:
        return ("conv", int(b.group(1)), c.group(1), kind.group(1))
    if b is None and c is None and _DENSE.search(name) is not None:
        return ("dense", None, None, kind.group(1))
    return None


def list_keras_tensors(path):
    """[(name, ndarray)] in the file's own order -- the order Keras' positional load_weights consumes (model.py:44) -- of a weights-only
    file (`save_weights`, the reference's checkpoints: layer groups at the root) or a full-model file (`model.save`: the same tree under
    /model_weights)."""
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
    return tensors


def check_tensor_list(tensors, dilations=DEFAULT_DILATIONS):
    """Compare a file's tensor list [(name, shape-or-array)] with the graph `dilations` implies.  Returns
    (rows, problems, unverified): rows = [(i, file name, file shape, expected name, expected shape, verdict)] with verdict "ok" (shape
    fits and the name says the same place), "shape only" (name follows no known scheme), or the mismatch; problems = the mismatches as
    text; unverified = same-shaped conv tensors whose ORDER nothing but position vouches for."""
    expected = tensor_shapes(dilations)
    rows, problems, unverified = [], [], []
    if len(tensors) != len(expected):
        problems.append(f"{len(tensors)} weight tensors, the Sig2Seq graph has {len(expected)}: " + ", ".join(n for n, _ in tensors[:6]) + " ...")
    for i in range(max(len(tensors), len(expected))):
        name, arr = tensors[i] if i < len(tensors) else (None, None)
        ename, eshape = expected[i] if i < len(expected) else (None, None)
        shape = None if arr is None else tuple(getattr(arr, "shape", arr))
        verdict = "ok"
        if name is None or ename is None:
            verdict = "missing in the file" if name is None else "not in the graph"
        elif shape != tuple(eshape):
            verdict = f"shape {shape}, expected {tuple(eshape)}"
            problems.append(f"tensor {i} {name} has shape {shape}, expected {tuple(eshape)} for {ename}")
        else:
            role, erole = tensor_role(name), tensor_role(ename + ":0")
            if role is None:
                verdict = "shape only"
                if erole[0] == "conv" and erole[1] > 0:
                    unverified.append(name)
            elif role != erole:
                verdict = f"name says {'/'.join(str(x) for x in role[1:] if x is not None) if role[0] == 'conv' else 'dense/' + role[3]}"
                problems.append(f"tensor {i} is named {name} but position {i} of the graph is {ename}: the file's order differs from "
                                "load_weights order (model.py:44) and same-shaped tensors would load into the wrong layers")
        rows.append((i, name, shape, ename, None if eshape is None else tuple(eshape), verdict))
    return rows, problems, unverified


def read_keras_weights(path, dilations=DEFAULT_DILATIONS, order="strict"):
    """Flat float32 parameters in load_weights order.  Tensors are taken in file order like Keras' positional load (model.py:44) and each
    is checked against the graph twice: its shape, and -- when its name follows keras-tcn 3.5's scheme (SURVEY.md 8c:
    `residual_block_<b>/conv1D_<j>` | `matching_conv1D`, `dense*`, `kernel` | `bias`) -- that the name says the same block, conv and
    kind as the position does.  conv1D_0 and conv1D_1 of blocks 1..5 all have shape [3,256,256]: only the names can tell a file whose
    order differs, and such a file raises ValueError instead of loading silently wrong.  order="by_name" places every tensor by what
    its name says instead (every name must then parse and the set must cover the graph exactly): tools/verify_h5.py --by-name."""
    expected = tensor_shapes(dilations)
    tensors = list_keras_tensors(path)
    if order == "by_name":
        def slot_key(role, shape):   # (the two Dense layers carry session-numbered names: their shapes tell them apart)
            return role + (tuple(shape),) if role[0] == "dense" else role
        slots = {slot_key(tensor_role(en + ":0"), es): i for i, (en, es) in enumerate(expected)}
        placed = [None] * len(expected)
        for name, arr in tensors:
            role = tensor_role(name)
            if role is None:
                raise ValueError(f"{path}: tensor {name} follows no known naming scheme; it cannot be placed by name")
            key = slot_key(role, arr.shape)
            if key not in slots or placed[slots[key]] is not None:
                raise ValueError(f"{path}: tensor {name} {tuple(arr.shape)} has no (free) place in the graph")
            placed[slots[key]] = (name, arr)
        if any(p is None for p in placed):
            raise ValueError(f"{path}: no tensor for " + ", ".join(en for (en, _), p in zip(expected, placed) if p is None))
        tensors = placed
    elif order != "strict":
        raise ValueError("order must be 'strict' or 'by_name'")
    _, problems, unverified = check_tensor_list(tensors, dilations)
    if problems:
        raise ValueError(f"{path}: " + "; ".join(problems[:4]) + (f" (+{len(problems) - 4} more)" if len(problems) > 4 else "")
                         + " -- tools/verify_h5.py prints the whole table")
    if unverified:
        import warnings
        warnings.warn(f"{path}: {len(unverified)} same-shaped conv tensors carry names this reader does not know (first: {unverified[0]}); "
                      "their order is taken on trust (positional, as Keras loads them)")
    return np.concatenate([np.ascontiguousarray(arr, dtype=np.float32).ravel() for _, arr in tensors])


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
