"""Minimal HDF5 access through ctypes -> libhdf5 (no h5py / ont_fast5_api in this image).

Only what the hot path's two file formats need: list a group's members in name order, read numeric
datasets and string attributes (fast5: radian/basecall.py:70-76; Keras weights-only .h5:
radian/model.py:44), and just enough writing to build fixtures and export files.
"""
import ctypes
import ctypes.util
import os

import numpy as np

hid_t = ctypes.c_int64
herr_t = ctypes.c_int
hsize_t = ctypes.c_uint64

_lib = None
_types = {}

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3


class H5Error(RuntimeError):
    pass


def _find_lib():
    cands = []
    if os.environ.get("RADIAN_HDF5_LIB"):
        cands.append(os.environ["RADIAN_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        cands.append(found)
    cands += ["libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
              "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"]
    for c in cands:
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    raise H5Error("libhdf5 not found (set RADIAN_HDF5_LIB=/path/to/libhdf5.so); needed to read fast5 / Keras .h5 files")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = _find_lib()
    sig = {
        "H5open": (herr_t, []),
        "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
        "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Oopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Oclose": (herr_t, [hid_t]),
        "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Literate": (herr_t, [hid_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.c_void_p, ctypes.c_void_p]),
        "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Screate": (hid_t, [ctypes.c_int]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Tget_class": (ctypes.c_int, [hid_t]),
        "H5Tget_size": (ctypes.c_size_t, [hid_t]),
        "H5Tget_sign": (ctypes.c_int, [hid_t]),
        "H5Tis_variable_str": (ctypes.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, ctypes.c_size_t]),
        "H5Tset_strpad": (herr_t, [hid_t, ctypes.c_int]),
        "H5Tget_strpad": (ctypes.c_int, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Aexists": (ctypes.c_int, [hid_t, ctypes.c_char_p]),
        "H5Aopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Acreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Aget_type": (hid_t, [hid_t]),
        "H5Aget_space": (hid_t, [hid_t]),
        "H5Aread": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Awrite": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Aclose": (herr_t, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_chunk": (herr_t, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t)]),
        "H5Pset_deflate": (herr_t, [hid_t, ctypes.c_uint]),
        "H5Pset_shuffle": (herr_t, [hid_t]),
        "H5Pset_fletcher32": (herr_t, [hid_t]),
        "H5Pset_fill_value": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.H5open() < 0:
        raise H5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)  # errors are reported through return codes -> H5Error

    def g(name):
        return hid_t.in_dll(L, name).value

    _types.update({
        np.dtype(np.int8): g("H5T_NATIVE_INT8_g"), np.dtype(np.uint8): g("H5T_NATIVE_UINT8_g"),
        np.dtype(np.int16): g("H5T_NATIVE_INT16_g"), np.dtype(np.uint16): g("H5T_NATIVE_UINT16_g"),
        np.dtype(np.int32): g("H5T_NATIVE_INT32_g"), np.dtype(np.uint32): g("H5T_NATIVE_UINT32_g"),
        np.dtype(np.int64): g("H5T_NATIVE_INT64_g"), np.dtype(np.uint64): g("H5T_NATIVE_UINT64_g"),
        np.dtype(np.float32): g("H5T_NATIVE_FLOAT_g"), np.dtype(np.float64): g("H5T_NATIVE_DOUBLE_g"),
        "c_s1": g("H5T_C_S1_g"), "dcpl": g("H5P_CLS_DATASET_CREATE_ID_g"),
    })
    _lib = L
    return L


_ITER_CB = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else s


class File:
    """HDF5 file through libhdf5; for reading, falls back to the pure-Python reader of the classic layout
    (radian_amd.h5pure) when no libhdf5 can be loaded or RADIAN_HDF5_PURE=1."""

    def __init__(self, path, mode="r"):
        self.path = str(path)
        self._pure = None
        if mode == "r":
            force = os.environ.get("RADIAN_HDF5_PURE") == "1"
            try:
                if force:
                    raise H5Error("forced")
                lib()
            except H5Error:
                from .h5pure import PureFile
                self._pure = PureFile(self.path)
                self.id = None
                return
        L = lib()
        if mode == "r":
            self.id = L.H5Fopen(_b(self.path), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self.id = L.H5Fcreate(_b(self.path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.id < 0:
            raise H5Error(f"cannot open HDF5 file {self.path!r} (mode {mode})")

    def close(self):
        if self._pure is not None:
            self._pure.close()
            self._pure = None
            return
        if self.id is not None and self.id >= 0:
            lib().H5Fclose(self.id)
        self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ reading
    def exists(self, path):
        """True if every component of the absolute path exists."""
        if self._pure is not None:
            return self._pure.exists(path)
        L = lib()
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if L.H5Lexists(self.id, _b(cur), H5P_DEFAULT) <= 0:
                return False
        return True

    def keys(self, group="/"):
        """Member names of a group in increasing name order (what h5py / ont_fast5_api iterate in)."""
        if self._pure is not None:
            return self._pure.keys(group)
        L = lib()
        gid = L.H5Gopen2(self.id, _b(group), H5P_DEFAULT)
        if gid < 0:
            raise H5Error(f"no group {group!r} in {self.path}")
        names = []

        def cb(_g, name, _info, _data):
            names.append(name.decode("utf-8"))
            return 0

        cbf = _ITER_CB(cb)
        idx = hsize_t(0)
        rc = L.H5Literate(gid, 0, 0, ctypes.byref(idx), ctypes.cast(cbf, ctypes.c_void_p), None)  # H5_INDEX_NAME, H5_ITER_INC
        L.H5Gclose(gid)
        if rc < 0:
            raise H5Error(f"H5Literate failed on {group!r}")
        return sorted(names)

    def read(self, path):
        """Whole numeric dataset -> numpy array (native dtype of matching class/size/sign)."""
        if self._pure is not None:
            return self._pure.read(path)
        L = lib()
        did = L.H5Dopen2(self.id, _b(path), H5P_DEFAULT)
        if did < 0:
            raise H5Error(f"no dataset {path!r} in {self.path}")
        try:
            sid = L.H5Dget_space(did)
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            L.H5Sclose(sid)
            shape = tuple(int(dims[i]) for i in range(nd))
            tid = L.H5Dget_type(did)
            cls, size, sign = L.H5Tget_class(tid), L.H5Tget_size(tid), L.H5Tget_sign(tid)
            L.H5Tclose(tid)
            if cls == H5T_INTEGER:
                dt = np.dtype(("i" if sign else "u") + str(size))
            elif cls == H5T_FLOAT:
                dt = np.dtype("f" + str(size))
            else:
                raise H5Error(f"dataset {path!r}: unsupported HDF5 type class {cls}")
            out = np.empty(shape, dtype=dt)
            if out.size:
                rc = L.H5Dread(did, _types[dt], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
                if rc < 0:
                    raise H5Error(f"H5Dread failed on {path!r} (a compression filter plugin such as VBZ may be missing)")
            return out
        finally:
            L.H5Dclose(did)

    def attr(self, obj_path, name, default=None):
        """String / numeric attribute of an object; arrays of strings come back as a list."""
        if self._pure is not None:
            return self._pure.attr(obj_path, name, default)
        L = lib()
        oid = L.H5Oopen(self.id, _b(obj_path), H5P_DEFAULT)
        if oid < 0:
            raise H5Error(f"no object {obj_path!r} in {self.path}")
        try:
            if L.H5Aexists(oid, _b(name)) <= 0:
                return default
            aid = L.H5Aopen(oid, _b(name), H5P_DEFAULT)
            tid = L.H5Aget_type(aid)
            sid = L.H5Aget_space(aid)
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            n = 1
            for i in range(nd):
                n *= int(dims[i])
            cls, size = L.H5Tget_class(tid), L.H5Tget_size(tid)
            try:
                if cls == H5T_STRING:
                    if L.H5Tis_variable_str(tid) > 0:
                        buf = (ctypes.c_char_p * n)()
                        if L.H5Aread(aid, tid, buf) < 0:
                            raise H5Error(f"H5Aread failed on {obj_path}@{name}")
                        vals = [(b or b"").decode("utf-8", "replace") for b in buf]
                    else:
                        raw = ctypes.create_string_buffer(size * n)
                        if L.H5Aread(aid, tid, raw) < 0:
                            raise H5Error(f"H5Aread failed on {obj_path}@{name}")
                        vals = [raw.raw[i * size:(i + 1) * size].split(b"\0")[0] for i in range(n)]
                        if L.H5Tget_strpad(tid) == 2:          # H5T_STR_SPACEPAD: h5py's conversion to a NumPy 'S' type drops the blanks
                            vals = [v.rstrip(b" ") for v in vals]
                        vals = [v.decode("utf-8", "replace") for v in vals]
                    return vals[0] if nd == 0 else vals
                if cls in (H5T_INTEGER, H5T_FLOAT):
                    sign = L.H5Tget_sign(tid)
                    dt = np.dtype(("f" if cls == H5T_FLOAT else ("i" if sign else "u")) + str(size))
                    out = np.empty(n, dtype=dt)
                    if L.H5Aread(aid, _types[dt], out.ctypes.data_as(ctypes.c_void_p)) < 0:
                        raise H5Error(f"H5Aread failed on {obj_path}@{name}")
                    return out[0].item() if nd == 0 else out
                raise H5Error(f"attribute {obj_path}@{name}: unsupported type class {cls}")
            finally:
                L.H5Sclose(sid)
                L.H5Tclose(tid)
                L.H5Aclose(aid)
        finally:
            L.H5Oclose(oid)

    # ------------------------------------------------------------------ writing (fixtures / export)
    def create_group(self, path):
        L = lib()
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if L.H5Lexists(self.id, _b(cur), H5P_DEFAULT) > 0:
                continue
            gid = L.H5Gcreate2(self.id, _b(cur), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if gid < 0:
                raise H5Error(f"cannot create group {cur!r}")
            L.H5Gclose(gid)

    def write(self, path, array, chunks=None, filters=(), fill=None, store=True):
        """filters (chunked datasets only), applied in the order given: "shuffle", "fletcher32", ("deflate", level).
        fill: the dataset's fill value (H5Pset_fill_value); store=False creates the dataset without writing its elements -- storage
        stays unallocated and reads return the fill value (fixtures for the native fast5 reader's "no verdict" cases)."""
        L = lib()
        a = np.ascontiguousarray(array)
        parent = path.rsplit("/", 1)[0]
        if parent:
            self.create_group(parent)
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        sid = L.H5Screate_simple(a.ndim, dims, None)
        dcpl = H5P_DEFAULT
        if fill is not None or (chunks is not None and a.size):
            dcpl = L.H5Pcreate(_types["dcpl"])
        if fill is not None:
            fv = np.asarray(fill, dtype=a.dtype).reshape(1)
            if L.H5Pset_fill_value(dcpl, _types[a.dtype], fv.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise H5Error("cannot set the fill value")
        if chunks is not None and a.size:
            cd = (hsize_t * a.ndim)(*chunks)
            L.H5Pset_chunk(dcpl, a.ndim, cd)
            for f in filters:
                rc = (L.H5Pset_shuffle(dcpl) if f == "shuffle" else L.H5Pset_fletcher32(dcpl) if f == "fletcher32"
                      else L.H5Pset_deflate(dcpl, int(f[1])) if f[0] == "deflate" else -1)
                if rc < 0:
                    raise H5Error(f"cannot add filter {f!r} (this libhdf5 may lack it)")
        did = L.H5Dcreate2(self.id, _b(path), _types[a.dtype], sid, H5P_DEFAULT, dcpl, H5P_DEFAULT)
        if did < 0:
            raise H5Error(f"cannot create dataset {path!r}")
        if store and a.size and L.H5Dwrite(did, _types[a.dtype], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p)) < 0:
            raise H5Error(f"H5Dwrite failed on {path!r}")
        L.H5Dclose(did)
        L.H5Sclose(sid)
        if dcpl != H5P_DEFAULT:
            L.H5Pclose(dcpl)

    def set_attr_str(self, obj_path, name, value, kind="nullterm"):
        """String attribute: scalar (str) or 1-D array (list of str), in one of the three forms found in the wild:
        kind "nullterm" fixed-length, NUL-terminated (the C default); "nullpad" fixed-length NUL-padded with no room for
        a terminator -- what h5py writes for a NumPy 'S' array, i.e. Keras 2.4's `layer_names` / `weight_names`;
        "vlen" variable-length strings -- what h5py 3 writes for str objects; "spacepad" fixed-length padded with blanks (Fortran's form)."""
        L = lib()
        vals = [value] if isinstance(value, str) else list(value)
        enc = [v.encode("utf-8") for v in vals]
        tid = L.H5Tcopy(_types["c_s1"])
        keep = None
        if kind == "vlen":
            L.H5Tset_size(tid, ctypes.c_size_t(-1).value)   # H5T_VARIABLE
            keep = [ctypes.create_string_buffer(e) for e in enc]
            buf = (ctypes.c_char_p * len(enc))(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
        else:
            longest = max(len(e) for e in enc) if enc else 0
            size = max(1, longest + (0 if kind == "nullpad" else 4 if kind == "spacepad" else 1))
            L.H5Tset_size(tid, size)
            if kind == "nullpad":
                L.H5Tset_strpad(tid, 1)                      # H5T_STR_NULLPAD
            elif kind == "spacepad":
                L.H5Tset_strpad(tid, 2)                      # H5T_STR_SPACEPAD
            elif kind != "nullterm":
                raise ValueError(f"unknown string attribute kind {kind!r}")
            buf = ctypes.create_string_buffer(b"".join(e.ljust(size, b" " if kind == "spacepad" else b"\0") for e in enc), max(1, size * len(enc)))
        if isinstance(value, str):
            sid = L.H5Screate(0)  # H5S_SCALAR
        else:
            dims = (hsize_t * 1)(len(vals))
            sid = L.H5Screate_simple(1, dims, None)
        oid = L.H5Oopen(self.id, _b(obj_path), H5P_DEFAULT)
        aid = L.H5Acreate2(oid, _b(name), tid, sid, H5P_DEFAULT, H5P_DEFAULT)
        rc = L.H5Awrite(aid, tid, buf)
        L.H5Aclose(aid)
        L.H5Oclose(oid)
        L.H5Sclose(sid)
        L.H5Tclose(tid)
        del keep
        if rc < 0:
            raise H5Error(f"cannot write attribute {obj_path}@{name}")
