"""fast5 (HDF5) reader for the driver loop: replaces `ont_fast5_api.get_fast5_file(path).get_reads()`,
`read.get_raw_data()` and `read.read_id` (radian/basecall.py:7,70-76).

Multi-read files: root groups `read_<uuid>` with `Raw/Signal` (int16 DAQ values, unscaled), iterated in
name order; read id = group name without the `read_` prefix.  Single-read files: `Raw/Reads/Read_<n>/Signal`
with the id in that group's `read_id` attribute.  VBZ-compressed signals need ONT's HDF5 filter plugin
(HDF5_PLUGIN_PATH), as with ont_fast5_api."""
from pathlib import Path

import numpy as np

from . import h5


class Fast5Read:
    def __init__(self, f, read_id, signal_path):
        self._f = f
        self.read_id = read_id
        self._signal_path = signal_path

    def get_raw_data(self):
        return self._f.read(self._signal_path)


def iter_reads(path):
    """Yield Fast5Read objects of one file in ont_fast5_api's order."""
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

    def __init__(self, path):
        self.path = path
        self._f = None
        self._entries = None   # [(read_id or None, group path, signal path)]

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
        self._open()
        return len(self._entries)

    def reads(self, lo, hi):
        self._open()
        for i in range(lo, min(hi, len(self._entries))):
            rid, grp, sig = self._entries[i]
            if rid is None:
                rid = self._f.attr(grp, "read_id", default=grp.rsplit("/", 1)[1])
                if isinstance(rid, bytes):
                    rid = rid.decode()
            yield i, Fast5Read(self._f, rid, sig)

    def close(self):
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


def write_multi_fast5(path, reads):
    """Write {read_id: int16 array} as a multi-read fast5 (fixtures / synthetic runs)."""
    with h5.File(path, "w") as f:
        for rid, sig in reads.items():
            sig = np.ascontiguousarray(sig, dtype=np.int16)
            g = f"/read_{rid}"
            f.create_group(g + "/Raw")
            f.write(g + "/Raw/Signal", sig, chunks=(max(1, min(len(sig), 4096)),))
            f.set_attr_str(g + "/Raw", "read_id", rid)
