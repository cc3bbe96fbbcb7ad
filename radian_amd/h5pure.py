"""Pure-Python reader for the classic HDF5 on-disk layout -- the fallback of radian_amd.h5 when no libhdf5 is available.

Covers what a multi-/single-read fast5 written by libhdf5's default settings uses (and what the reference's sample
`radian/data/reads.fast5` is): superblock version 0/1, version-1 object headers with
continuation blocks, symbol-table groups (v1 B-tree + local heap + SNOD nodes), contiguous and chunked datasets
(v1 chunk B-tree) of fixed-point / IEEE float types WITHOUT filters, version-1 attributes with fixed-length strings
or numbers.  Anything newer (superblock >= 2, link messages, fractal heaps, filtered chunks such as VBZ) raises
H5PureError with the reason, so the caller can ask for a real libhdf5.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5PureError(RuntimeError):
    pass


class PureFile:
    def __init__(self, path):
        self.path = str(path)
        with open(self.path, "rb") as f:
            self.buf = f.read()
        base = -1
        off = 0
        while off < len(self.buf):
            if self.buf[off:off + 8] == SIG:
                base = off
                break
            off = 512 if off == 0 else off * 2
        if base < 0:
            raise H5PureError(f"{self.path}: not an HDF5 file")
        ver = self.buf[base + 8]
        if ver > 1:
            raise H5PureError(f"{self.path}: superblock version {ver} (only 0/1 are supported without libhdf5)")
        self.O = self.buf[base + 13]
        self.L = self.buf[base + 14]
        if self.O != 8 or self.L != 8:
            raise H5PureError("only 8-byte offsets/lengths are supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self._u64(p)
        p += 8 * 4  # base, free-space, eof, driver-info addresses
        # root group symbol table entry
        self.root = self._u64(p + 8) + self.base_addr
        self._hdr_cache = {}

    # ------------------------------------------------------------------ primitives
    def _u16(self, p):
        return struct.unpack_from("<H", self.buf, p)[0]

    def _u32(self, p):
        return struct.unpack_from("<I", self.buf, p)[0]

    def _u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    def close(self):
        self.buf = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr):
        """[(type, flags, data_offset, size)] of a version-1 object header, following continuation blocks."""
        if addr in self._hdr_cache:
            return self._hdr_cache[addr]
        b = self.buf
        if b[addr] != 1:
            if b[addr:addr + 4] == b"OHDR":
                raise H5PureError("version-2 object headers need libhdf5")
            raise H5PureError(f"bad object header at {addr}")
        nmsg = self._u16(addr + 2)
        size = self._u32(addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = self._u16(p), self._u16(p + 2), b[p + 4]
                data = p + 8
                if mtype == 0x0010:  # continuation
                    blocks.append((self._u64(data) + self.base_addr, self._u64(data + 8)))
                out.append((mtype, mflags, data, msize))
                p = data + msize
        self._hdr_cache[addr] = out
        return out

    def _msg(self, addr, mtype):
        for t, _, data, size in self._messages(addr):
            if t == mtype:
                return data, size
        return None

    # ------------------------------------------------------------------ groups
    def _group_entries(self, addr):
        """{name: object header address} of a group.  Memoised per group (the file is read-only): resolving /read_x/Raw/Signal
        walked the root group's whole symbol table for every read -- 15 ms per read in a 16 384-read file, 0.3 M samples/s
        (tools/host_feed_bench.py read)."""
        cache = self.__dict__.setdefault("_ent_cache", {})
        got = cache.get(addr)
        if got is None:
            got = cache[addr] = self._group_entries_uncached(addr)
        return got

    def _group_entries_uncached(self, addr):
        m = self._msg(addr, 0x0011)
        if m is None:
            return self._compact_links(addr)
        btree = self._u64(m[0]) + self.base_addr
        heap = self._u64(m[0] + 8) + self.base_addr
        if self.buf[heap:heap + 4] != b"HEAP":
            raise H5PureError("bad local heap")
        heap_data = self._u64(heap + 24) + self.base_addr
        out = {}

        def name_at(off):
            s = heap_data + off
            e = self.buf.index(b"\0", s)
            return self.buf[s:e].decode("utf-8")

        def walk(node):
            if self.buf[node:node + 4] == b"SNOD":
                n = self._u16(node + 6)
                p = node + 8
                for _ in range(n):
                    out[name_at(self._u64(p))] = self._u64(p + 8) + self.base_addr
                    p += 40
                return
            if self.buf[node:node + 4] != b"TREE":
                raise H5PureError("bad group B-tree node")
            used = self._u16(node + 6)
            p = node + 8 + 16  # skip siblings
            for _ in range(used):
                p += 8  # key
                walk(self._u64(p) + self.base_addr)
                p += 8

        walk(btree)
        return out

    def _compact_links(self, addr):
        """{name: address} of a "new-style" group whose links sit in its object header (compact storage)."""
        info = self._msg(addr, 0x0002)
        links = [(d, n) for t, _, d, n in self._messages(addr) if t == 0x0006]
        if info is None and not links:
            raise H5PureError("object is not a group")
        if info is not None:
            p = info[0]
            flags = self.buf[p + 1]
            q = p + 2 + (8 if flags & 1 else 0)
            if self._u64(q) != UNDEF:
                raise H5PureError("dense (fractal-heap) group storage needs libhdf5")
        out = {}
        for p, _ in links:
            if self.buf[p] != 1:
                raise H5PureError("unknown link message version")
            flags = self.buf[p + 1]
            q = p + 2
            ltype = 0
            if flags & 0x08:
                ltype = self.buf[q]
                q += 1
            if flags & 0x04:
                q += 8
            if flags & 0x10:
                q += 1
            lsz = 1 << (flags & 3)
            nlen = int.from_bytes(self.buf[q:q + lsz], "little")
            q += lsz
            name = self.buf[q:q + nlen].decode("utf-8")
            q += nlen
            if ltype != 0:
                continue  # soft / external links are not followed
            out[name] = self._u64(q) + self.base_addr
        return out

    def _resolve(self, path):
        addr = self.root
        for part in [q for q in path.split("/") if q]:
            ents = self._group_entries(addr)
            if part not in ents:
                raise KeyError(path)
            addr = ents[part]
        return addr

    def exists(self, path):
        try:
            self._resolve(path)
            return True
        except (KeyError, H5PureError):
            return False

    def keys(self, group="/"):
        return sorted(self._group_entries(self._resolve(group)))

    # ------------------------------------------------------------------ types / spaces
    def _dtype(self, p):
        cls = self.buf[p] & 0x0F
        bits0 = self.buf[p + 1]
        size = self._u32(p + 4)
        if cls == 0:
            if bits0 & 1:
                raise H5PureError("big-endian integers are not supported")
            return np.dtype(("<i" if bits0 & 8 else "<u") + str(size)), 8 + 4
        if cls == 1:
            if bits0 & 1:
                raise H5PureError("big-endian floats are not supported")
            return np.dtype("<f" + str(size)), 8 + 12
        if cls == 3:
            return np.dtype("S" + str(size)), 8
        raise H5PureError(f"datatype class {cls} needs libhdf5")

    def _shape(self, p):
        ver, rank = self.buf[p], self.buf[p + 1]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
        else:
            raise H5PureError("unknown dataspace version")
        return tuple(self._u64(q + 8 * i) for i in range(rank))

    # ------------------------------------------------------------------ datasets
    def read(self, path):
        addr = self._resolve(path)
        dt_m, sp_m, lay_m = self._msg(addr, 0x0003), self._msg(addr, 0x0001), self._msg(addr, 0x0008)
        if dt_m is None or sp_m is None or lay_m is None:
            raise H5PureError(f"{path} is not a dataset")
        if self._msg(addr, 0x000B) is not None:
            raise H5PureError(f"{path} is stored with a filter pipeline (compression): needs libhdf5 (+ plugin)")
        dt, _ = self._dtype(dt_m[0])
        shape = self._shape(sp_m[0])
        n = int(np.prod(shape)) if shape else 1
        p = lay_m[0]
        if self.buf[p] != 3:
            raise H5PureError("only version-3 data layout messages are supported")
        cls = self.buf[p + 1]
        out = np.zeros(n, dtype=dt)
        if cls == 1:  # contiguous
            a = self._u64(p + 2)
            if a != UNDEF and n:
                out[:] = np.frombuffer(self.buf, dtype=dt, count=n, offset=a + self.base_addr)
        elif cls == 2:  # chunked
            rank = self.buf[p + 2]
            bt = self._u64(p + 3)
            cdims = [self._u32(p + 11 + 4 * i) for i in range(rank)]
            if rank - 1 != 1 or len(shape) != 1:
                raise H5PureError("only one-dimensional chunked datasets are supported without libhdf5")
            clen = cdims[0]
            if bt != UNDEF:
                for off, caddr, csize in self._chunks(bt + self.base_addr, rank):
                    cnt = min(clen, n - off)
                    if cnt > 0:
                        out[off:off + cnt] = np.frombuffer(self.buf, dtype=dt, count=cnt, offset=caddr)
        elif cls == 0:  # compact
            sz = self._u16(p + 2)
            out[:] = np.frombuffer(self.buf, dtype=dt, count=min(n, sz // dt.itemsize), offset=p + 4)
        else:
            raise H5PureError("unknown layout class")
        return out.reshape(shape).astype(dt.newbyteorder("="))

    def _chunks(self, node, rank):
        """(first element offset, data address, stored size) of every chunk under a v1 chunk B-tree node."""
        if self.buf[node:node + 4] != b"TREE" or self.buf[node + 4] != 1:
            raise H5PureError("bad chunk B-tree node")
        level, used = self.buf[node + 5], self._u16(node + 6)
        p = node + 8 + 16
        keysz = 8 + 8 * rank
        for _ in range(used):
            csize, fmask = self._u32(p), self._u32(p + 4)
            off0 = self._u64(p + 8)
            child = self._u64(p + keysz) + self.base_addr
            if level == 0:
                if fmask:
                    raise H5PureError("filtered chunk")
                yield off0, child, csize
            else:
                yield from self._chunks(child, rank)
            p += keysz + 8

    # ------------------------------------------------------------------ attributes
    def attr(self, obj_path, name, default=None):
        addr = self._resolve(obj_path)
        for t, _, data, size in self._messages(addr):
            if t != 0x000C:
                continue
            ver = self.buf[data]
            if ver not in (1, 2, 3):
                continue
            nsz, tsz, ssz = self._u16(data + 2), self._u16(data + 4), self._u16(data + 6)
            p = data + 8 + (1 if ver == 3 else 0)
            pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
            aname = self.buf[p:p + nsz].split(b"\0")[0].decode("utf-8")
            p += pad(nsz)
            tp = p
            p += pad(tsz)
            sp = p
            p += pad(ssz)
            if aname != name:
                continue
            cls = self.buf[tp] & 0x0F
            if cls == 9:
                raise H5PureError("variable-length string attributes need libhdf5")
            dt, _ = self._dtype(tp)
            shape = self._shape(sp) if ssz else ()
            n = int(np.prod(shape)) if shape else 1
            vals = np.frombuffer(self.buf, dtype=dt, count=n, offset=p)
            if dt.kind == "S":
                out = [v.split(b"\0")[0].decode("utf-8", "replace") for v in vals.tolist()]
                return out[0] if not shape else out
            return vals[0].item() if not shape else vals.copy()
        return default
