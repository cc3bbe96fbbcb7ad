"""fast5 (HDF5) reader for the driver loop: replaces `ont_fast5_api.get_fast5_file(path).get_reads()`,
`read.get_raw_data()` and `read.read_id` (radian/basecall.py:7,70-76).

Multi-read files: root groups `read_<uuid>` with `Raw/Signal` (int16 DAQ values, unscaled), iterated in
name order; read id = group name without the `read_` prefix.  Single-read files: `Raw/Reads/Read_<n>/Signal`
with the id in that group's `read_id` attribute.  VBZ-compressed signals need ONT's HDF5 filter plugin
(HDF5_PLUGIN_PATH), as with ont_fast5_api.

Two readers behind the same interface.  The NATIVE batch reader (csrc/fast5.hip, rd_fast5_*: a bounds-checked walk of the classic HDF5
layout over a mapping of the file) resolves and copies a block of reads per call, with the interpreter lock released -- ~2 us per
4096-sample read against ~63 through libhdf5 plus the per-read Python hand-off; it answers RD_ERR_FORMAT without a verdict for anything it
does not recognise (newer layouts, filters other than HDF5's built-in deflate / shuffle / Fletcher-32, other sample types), and then -- for that file, from that read on -- the ctypes ->
libhdf5 reader (h5.py) does the work as before and its errors are the verdict.  RADIAN_FAST5_NATIVE=0 switches the native reader off."""
import ctypes
import os
from pathlib import Path

import numpy as np

from . import h5

RD_ERR_FORMAT = -6
_BLOCK = 256          # reads per native call
_ID_STRIDE = 160


class MemRead:
    """a read whose signal has been read already (the native batch reader's product): the fast5 read interface of basecall.py:72-76"""
    __slots__ = ("read_id", "_sig")

    def __init__(self, read_id, sig):
        self.read_id, self._sig = read_id, sig

    def get_raw_data(self):
        return self._sig


class NativeFile:
    """rd_fast5 handle of one file.  Raises Unrecognised where the native reader has no verdict."""

    class Unrecognised(Exception):
        pass

    def __init__(self, path):
        from . import _lib
        self._L = _lib.load()
        h = ctypes.c_void_p()
        rc = self._L.rd_fast5_open(os.fsencode(path), ctypes.byref(h))
        if rc == RD_ERR_FORMAT:
            raise NativeFile.Unrecognised(self._L.rd_last_error().decode("utf-8", "replace"))
        if rc != 0:
            raise OSError(self._L.rd_last_error().decode("utf-8", "replace"))
        self._h = h
        n = ctypes.c_int64(0)
        self._L.rd_fast5_count(self._h, ctypes.byref(n))
        self.n = n.value

    def batch(self, lo, hi):
        """reads [lo, hi) -> (ids, samples int16 [total], offsets int64 [hi - lo + 1])"""
        k = hi - lo
        lens = np.zeros(k, dtype=np.int64)
        rc = self._L.rd_fast5_lengths(self._h, lo, hi, lens.ctypes.data_as(ctypes.c_void_p))
        if rc == 0:
            tot = int(lens.sum())
            samples = np.empty(tot, dtype=np.int16)
            off = np.zeros(k + 1, dtype=np.int64)
            ids = ctypes.create_string_buffer(k * _ID_STRIDE)
            rc = self._L.rd_fast5_read_batch(self._h, lo, hi, samples.ctypes.data_as(ctypes.c_void_p), tot, off.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.cast(ids, ctypes.c_void_p), _ID_STRIDE)
        if rc == RD_ERR_FORMAT:
            raise NativeFile.Unrecognised(self._L.rd_last_error().decode("utf-8", "replace"))
        if rc != 0:
            raise OSError(self._L.rd_last_error().decode("utf-8", "replace"))
        raw = ids.raw
        names = [raw[i * _ID_STRIDE: raw.index(b"\0", i * _ID_STRIDE)].decode("utf-8") for i in range(k)]
        return names, samples, off

    def close(self):
        if self._h is not None:
            self._L.rd_fast5_close(self._h)
            self._h = None


def _native_enabled():
    return os.environ.get("RADIAN_FAST5_NATIVE", "1") != "0" and not os.environ.get("RADIAN_HDF5_PURE")


def _open_native(path):
    if not _native_enabled():
        return None
    try:
        return NativeFile(path)
    except NativeFile.Unrecognised:
        return None


class Fast5Read:
    def __init__(self, f, read_id, signal_path):
        self._f = f
        self.read_id = read_id
        self._signal_path = signal_path

    def get_raw_data(self):
        return self._f.read(self._signal_path)


def iter_reads(path):
    """Yield the reads of one file in ont_fast5_api's order (objects with .read_id and .get_raw_data())."""
    nat = _open_native(path)
    if nat is not None:
        src = Fast5Source(path, _native=nat)
        try:
            for _, r in src.reads(0, src.n_reads()):
                yield r
        finally:
            src.close()
        return
    with h5.File(path, "r") as f:
        names = f.keys("/")
        multi = [n for n in names if n.startswith("read_")]
        if multi:
            for n in multi:
                yield Fast5Read(f, n[len("read_"):], f"/{n}/Raw/Signal")
        elif f.exists("/Raw/Reads"):
            for n in f.keys("/Raw/Reads"):
                rid = f.attr(f"/Raw/Reads/{n}", "read_id", default=n)
                if isinstance(rid, bytes):
                    rid = rid.decode()
                yield Fast5Read(f, rid, f"/Raw/Reads/{n}/Signal")


def list_files(fast5_dir):
    """Every *.fast5 under fast5_dir, recursive, in Path.rglob order (basecall.py:70)."""
    return [str(p) for p in Path(fast5_dir).rglob("*.fast5")]


def iter_directory(fast5_dir):
    """Every read of every *.fast5 under fast5_dir, recursive, in Path.rglob order (basecall.py:70-72)."""
    for p in list_files(fast5_dir):
        for r in iter_reads(p):
            yield r


class Fast5Source:
    """One fast5 file as a unit of the multi-GPU work queue (dist.FileReadQueue): opened lazily, by the ranks that claim
    reads from it only.  n_reads() / reads(lo, hi) address the file's reads in iter_reads order."""

    def __init__(self, path, _native=None):
        self.path = path
        self._f = None
        self._entries = None   # [(read_id or None, group path, signal path)]
        self._nat = _native
        self._nat_tried = _native is not None

    def _open_nat(self):
        if not self._nat_tried:
            self._nat_tried = True
            self._nat = _open_native(self.path)
        return self._nat

    def _open(self):
        if self._f is None:
            self._f = h5.File(self.path, "r")     # (again, after a close(): the work queue may count a file's reads ahead of time)
        if self._entries is not None:
            return
        f = self._f
        names = f.keys("/")
        multi = [n for n in names if n.startswith("read_")]
        if multi:
            self._entries = [(n[len("read_"):], None, f"/{n}/Raw/Signal") for n in multi]
        elif f.exists("/Raw/Reads"):
            self._entries = [(None, f"/Raw/Reads/{n}", f"/Raw/Reads/{n}/Signal") for n in f.keys("/Raw/Reads")]
        else:
            self._entries = []

    def n_reads(self):
        nat = self._open_nat()
        if nat is not None:
            return nat.n
        self._open()
        return len(self._entries)

    def reads(self, lo, hi):
        nat = self._open_nat()
        if nat is not None:
            hi = min(hi, nat.n)
            while lo < hi:
                b = min(hi, lo + _BLOCK)
                try:
                    ids, samples, off = nat.batch(lo, b)
                except NativeFile.Unrecognised:
                    # a read the native reader has no verdict on (a compressed signal, say): libhdf5 takes the file over from this read on
                    nat.close()
                    self._nat = None
                    break
                for j in range(b - lo):
                    yield lo + j, MemRead(ids[j], samples[off[j]: off[j + 1]])
                lo = b
            else:
                return
        self._open()
        for i in range(lo, min(hi, len(self._entries))):
            rid, grp, sig = self._entries[i]
            if rid is None:
                rid = self._f.attr(grp, "read_id", default=grp.rsplit("/", 1)[1])
                if isinstance(rid, bytes):
                    rid = rid.decode()
            yield i, Fast5Read(self._f, rid, sig)

    def close(self):
        if self._nat is not None:
            self._nat.close()
            self._nat = None
            self._nat_tried = False     # (again, after a close(): the work queue may count a file's reads ahead of time)
        if self._f is not None:
            self._f.close()
            self._f = None
            self._entries = None


class ListSource:
    """An in-memory sequence of reads with the Fast5Source interface (synthetic runs, tests)."""

    def __init__(self, reads):
        self._reads = reads

    def n_reads(self):
        return len(self._reads)

    def reads(self, lo, hi):
        for i in range(lo, min(hi, len(self._reads))):
            yield i, self._reads[i]

    def close(self):
        pass


def write_multi_fast5(path, reads, filters=(), chunk=4096):
    """Write {read_id: int16 array} as a multi-read fast5 (fixtures / synthetic runs).  filters: h5.File.write's, e.g. (("deflate", 1),)
    = the gzip level-1 signals of pre-VBZ MinKNOW files."""
    with h5.File(path, "w") as f:
        for rid, sig in reads.items():
            sig = np.ascontiguousarray(sig, dtype=np.int16)
            g = f"/read_{rid}"
            f.create_group(g + "/Raw")
            f.write(g + "/Raw/Signal", sig, chunks=(max(1, min(len(sig), chunk)),), filters=filters)
            f.set_attr_str(g + "/Raw", "read_id", rid)
