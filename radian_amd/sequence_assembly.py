"""Host-side step after the hot path (chunk mode): stitch the per-window fragments of a read into one sequence.

Interface of radian/sequence_assembly.py:19-48,90-97 (`simple_assembly`, `add_count`, `index2base` -- same names,
arguments, results and error behaviour) plus the two lines of radian/basecall.py:122-123 that turn the vote matrix into
a string.  The formulation is this repository's own: the placement of every fragment is computed first (one
`difflib.SequenceMatcher` per consecutive pair -- the stdlib class the reference calls, including its autojunk heuristic
for fragments >= 200 characters), then all votes are counted in one `numpy.bincount` over a matrix sized once.
Behaviour is pinned by tests/golden/seq_assembly_cases.json (from the reference) and by the randomised comparison in tests/test_host_cpu.py.
"""
import difflib
import os

import numpy as np

_BASES = "ACGT"
_GROW = 1000                                      # the reference's vote matrix starts at, and grows by, 1000 columns
_CODE = np.full(256, -1, dtype=np.int64)          # byte -> row of the vote matrix
for _row, _pair in enumerate(("Aa", "Cc", "Gg", "Tt")):
    for _ch in _pair:
        _CODE[ord(_ch)] = _row


def _codes(fragment):
    """'ACGTacgt' string -> int64 rows 0..3; any other character raises KeyError like the reference's dict lookup."""
    raw = np.frombuffer(fragment.encode("latin-1", "replace"), dtype=np.uint8)
    rows = _CODE[raw]
    if rows.size and rows.min() < 0:
        raise KeyError(fragment[int(np.argmin(rows))])
    return rows


def fragment_starts(fragments):
    """Column of the vote matrix at which each fragment starts (may be negative).  Fragment i is shifted against
    fragment i-1 by (a - b) of the FIRST longest matching block (a, b, size) difflib reports for the pair."""
    starts, at = [], 0
    for i, frag in enumerate(fragments):
        if i:
            blocks = difflib.SequenceMatcher(None, fragments[i - 1], frag).get_matching_blocks()
            best = blocks[0]
            for blk in blocks[1:]:
                if blk.size > best.size:
                    best = blk
            at += best.a - best.b
        starts.append(at)
    return starts


def _check_reference_capacity(starts, lengths):
    """The reference's matrix gains 1000 columns at most once per fragment and its element-wise writes raise IndexError
    beyond that (sequence_assembly.py:29-33,47); keep that failure instead of silently producing votes it never could."""
    cap = _GROW
    for i, (st, n) in enumerate(zip(starts, lengths)):
        if i and st + n > cap:
            cap += _GROW
        last = max(st, 0) + (n + min(st, 0)) - 1       # last column written (the part left of column 0 is dropped)
        if n + min(st, 0) > 0 and last >= cap:
            raise IndexError(f"index {last} is out of bounds for axis 0 with size {cap}")


def add_count(votes, start, segment):
    """Add one vote per character of `segment` to votes[base, start + i]; characters left of column 0 are dropped."""
    rows = _codes(segment)
    cols = start + np.arange(rows.size)
    keep = cols >= 0
    if cols.size and cols[-1] >= votes.shape[1]:
        raise IndexError(f"index {int(cols[-1])} is out of bounds for axis 0 with size {votes.shape[1]}")
    np.add.at(votes, (rows[keep], cols[keep]), 1)


def simple_assembly(bpreads):
    """[4, L] float64 vote matrix of the fragments laid out by fragment_starts.  L is the furthest column reached by the
    SECOND and later fragments (as in the reference: a read with a single fragment gives L = 0)."""
    bpreads = list(bpreads)
    starts = fragment_starts(bpreads)
    lengths = [len(f) for f in bpreads]
    _check_reference_capacity(starts, lengths)
    width = max([0] + [st + n for st, n in list(zip(starts, lengths))[1:]])
    flat = np.zeros(4 * width, dtype=np.int64)
    if width:
        where = []
        for st, frag in zip(starts, bpreads):
            rows = _codes(frag)
            cols = st + np.arange(rows.size)
            ok = (cols >= 0) & (cols < width)
            where.append(rows[ok] * width + cols[ok])
        flat = np.bincount(np.concatenate(where), minlength=4 * width) if where else flat
    else:
        for frag in bpreads:
            _codes(frag)                                 # still reject foreign characters
    return flat.reshape(4, width).astype(np.float64)


def index2base(read):
    """Integer labels 0..3 -> 'ACGT' string."""
    return "".join(map(_BASES.__getitem__, read))


def consensus_sequence(fragments):
    """radian/basecall.py:122-123: index2base(np.argmax(simple_assembly(fragments), axis=0))."""
    votes = simple_assembly(fragments)
    if votes.shape[1] == 0:
        return ""
    return index2base(np.argmax(votes, axis=0))


_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def labels_to_str(labels):
    """index2base for a label array as the device returns it: one table lookup over the whole array instead of a Python-level
    join per base (a global-mode read is thousands of bases; at ~10 M characters/s the join was a third of a short job)."""
    a = np.asarray(labels)
    if a.dtype == np.uint8 and a.ndim == 1:
        if a.size and a.max() > 3:
            raise IndexError("list index out of range")     # what index2base raises for a label outside 0..3
        return _ASCII[a].tobytes().decode("ascii")
    return index2base(labels)


def consensus_batch(labels, lens, windows_per_read, threads=None):
    """consensus_sequence for a batch of reads straight from the device's output: labels uint8 [n_windows, chunk_len], lens
    int32 [n_windows] (labels[w, :lens[w]] is window w's fragment), windows_per_read the number of windows of each read, in
    order.  Runs the library's native restatement of simple_assembly + difflib (rd_stitch_chunk, csrc/stitch.hip) on
    `threads` host threads -- the pure-Python stitch above costs ~4.5 ms per read once fragments are ~200 bases long.
    A read for which the reference raises (its IndexError capacity rule) is handed to the Python path, which raises it."""
    import ctypes
    from . import _lib
    L = _lib.load()
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    nwin, chunk_len = labels.shape
    nw = np.asarray(windows_per_read, dtype=np.int64)
    n_reads = int(nw.shape[0])
    win_off = np.zeros(n_reads + 1, dtype=np.int32)
    win_off[1:] = np.cumsum(nw)
    if int(win_off[-1]) != nwin or lens.shape[0] != nwin:
        raise ValueError("windows_per_read does not add up to the label matrix")
    csum = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
    cap = csum[win_off[1:]] - csum[win_off[:-1]]          # labels of the read's windows: its consensus cannot be longer
    seq_off = np.zeros(n_reads, dtype=np.int64)
    seq_off[1:] = np.cumsum(cap[:-1])
    out = np.zeros(int(cap.sum()) + 1, dtype=np.uint8)
    seq_len = np.zeros(n_reads, dtype=np.int32)
    if threads is None:
        threads = max(1, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)) - 2))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = L.rd_stitch_chunk(p(labels), p(lens), int(chunk_len), p(win_off), n_reads, p(out), p(seq_off), p(seq_len), int(threads))
    if rc != 0:
        raise RuntimeError(L.rd_last_error().decode("utf-8", "replace"))
    text = _ASCII[out].tobytes().decode("ascii")
    res = []
    for r in range(n_reads):
        if seq_len[r] < 0:   # the reference raises here: let the Python mirror do it
            w0 = int(win_off[r])
            res.append(consensus_sequence([labels_to_str(labels[w0 + i, : lens[w0 + i]]) for i in range(int(nw[r]))]))
            continue
        o = int(seq_off[r])
        res.append(text[o: o + int(seq_len[r])])
    return res
