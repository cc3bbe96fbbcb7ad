"""Golden raw signals (tests/golden/reads_fast5_signals.npz) as objects with the fast5 read interface."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Read:
    def __init__(self, rid, sig):
        self.read_id, self._sig = rid, sig

    def get_raw_data(self):
        return self._sig


def golden_reads(truncate=None, extra_bad=False):
    ids = json.load(open(os.path.join(GOLDEN, "reads_fast5_ids.json")))["read_ids"]
    sig = np.load(os.path.join(GOLDEN, "reads_fast5_signals.npz"))
    reads = [_Read(r, sig[r][:truncate] if truncate else sig[r]) for r in ids]
    if extra_bad:
        reads.insert(2, _Read("flat-signal", np.full(300, 512, dtype=np.int16)))  # MAD == 0 -> skipped
        reads.insert(4, _Read("empty-signal", np.zeros(0, dtype=np.int16)))       # empty -> skipped
    return reads
