"""Host-side step after the hot path (chunk mode): stitch the per-window fragments of a read into one sequence.

Interface of radian/sequence_assembly.py:19-48,90-97 (`simple_assembly`, `add_count`, `index2base` -- same names,
arguments, results and error behaviour) plus the two lines of radian/basecall.py:122-123 that turn the vote matrix into
a string.  The formulation is this repository's own: the placement of every fragment is computed first (one
`difflib.SequenceMatcher` per consecutive pair -- the stdlib class the reference calls, including its autojunk heuristic
for fragments >= 200 characters), then all votes are counted in one `numpy.bincount` over a matrix sized once.
Behaviour is pinned by tests/golden/seq_assembly_cases.json (from the reference) and by the randomised comparison in tests/test_host_cpu.py.
"""
import difflib

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


def labels_to_str(labels):
    return index2base(labels)
