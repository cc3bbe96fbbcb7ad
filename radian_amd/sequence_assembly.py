"""Host-side step after the hot path (chunk mode): stitch per-window fragments into one read.

Mirrors radian/sequence_assembly.py:19-48,90-97 (simple_assembly, add_count, index2base; same names and
results, including difflib's autojunk behaviour for fragments >= 200 characters) and the two lines of
radian/basecall.py:122-123 that turn the vote matrix into a string."""
import difflib

import numpy as np

_BASE_INDEX = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}
_BASES = "ACGT"


def add_count(concensus, start_indx, segment):
    """radian/sequence_assembly.py:42-48."""
    if start_indx < 0:
        segment = segment[-start_indx:]
        start_indx = 0
    if not segment:
        return
    idx = np.fromiter((_BASE_INDEX[b] for b in segment), dtype=np.int64, count=len(segment))
    np.add.at(concensus, (idx, start_indx + np.arange(len(segment))), 1)


def simple_assembly(bpreads):
    """radian/sequence_assembly.py:19-39: align consecutive fragments on the first longest difflib matching block,
    accumulate per-column base votes.  Returns the [4, L] vote matrix (float64, like the reference)."""
    concensus = np.zeros([4, 1000])
    pos = 0
    length = 0
    census_len = 1000
    for indx, bpread in enumerate(bpreads):
        if indx == 0:
            add_count(concensus, 0, bpread)
            continue
        d = difflib.SequenceMatcher(None, bpreads[indx - 1], bpread)
        match_block = max(d.get_matching_blocks(), key=lambda x: x[2])
        disp = match_block[0] - match_block[1]
        if disp + pos + len(bpread) > census_len:
            concensus = np.pad(concensus, ((0, 0), (0, 1000)), mode="constant", constant_values=0)
            census_len += 1000
        add_count(concensus, pos + disp, bpread)
        pos += disp
        length = max(length, pos + len(bpread))
    return concensus[:, :length]


def index2base(read):
    """radian/sequence_assembly.py:90-97."""
    return "".join(_BASES[x] for x in read)


def consensus_sequence(fragments):
    """radian/basecall.py:122-123: index2base(np.argmax(simple_assembly(fragments), axis=0))."""
    cons = simple_assembly(fragments)
    if cons.shape[1] == 0:
        return ""
    return index2base(np.argmax(cons, axis=0))


def labels_to_str(labels):
    return "".join(_BASES[c] for c in labels)
