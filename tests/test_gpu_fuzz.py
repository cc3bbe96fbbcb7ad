"""Bounded, seeded runs of the three differential fuzzers inside `pytest -m gpu`: the campaigns that found the
pipeline's bugs (tests/fuzz_decode.py, fuzz_reads.py, fuzz_pipe.py; hundreds to thousands of rounds, builder-run) leave a fixed
slice here so that every driver run repeats it.  Sizes are chosen for about a minute in all on one MI355X."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_decode_bounded():
    """beam search vs the oracle: 30 rounds (60 until round 6: the suite's budget, and the width list now reaches 257, where the oracle's side is slow) of random width (1..160: the two-sequences-per-wave widths, the wave-per-sequence kernels, the
    general kernel above 51), launch form (incl. the work queue), arithmetic, row type and LM order; every labeling identical"""
    import fuzz_decode
    total, bad = fuzz_decode.run(rounds=30, seed=404, tmax=300, nseq=400)
    assert total == 30 * 400 and bad == 0


def test_fuzz_reads_bounded():
    """streamed reads-level paths vs the window-level paths: 300 random geometries (chunk 64..2048, any step, ragged read sets, both
    decode types, both precisions); labels identical"""
    import fuzz_reads
    n, bad = fuzz_reads.run(iters=300, seed=404, max_len=4000)
    assert n == 300 and bad == 0


def test_fuzz_pipe_bounded():
    """the in-context pipeline vs the blocking entry points: 80 rounds (150 until round 6) of random submit sequences on one context (decode type, geometry,
    width, LM, thresholds, logits, precision, lanes, group size, partition changing between rounds; progress polled at random)"""
    import fuzz_pipe
    n, bad = fuzz_pipe.run(rounds=80, seed=404, max_len=5000, max_reads=16)
    assert n >= 160 and bad == 0
