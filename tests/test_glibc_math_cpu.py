"""radian_amd/csrc/glibc_math.h (the beam search's "glibc" arithmetic: exp, log, log1p restated operation for operation from
glibc 2.35's x86-64 FMA build) against the running libm, bit for bit, on the host; and its tables against libm.so.6."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, glibc_math_check_binary


def test_restated_routines_are_bit_identical_to_libm():
    exe = glibc_math_check_binary(force=True)
    assert exe is not None, "gcc not available"
    r = subprocess.run([exe, "6"], capture_output=True, text=True)     # 6 million arguments per range, ~30 M per routine
    assert r.returncode == 0, r.stdout[-2000:]
    assert "exp 0, log 0, log1p 0, logaddexp 0" in r.stdout, r.stdout[-500:]


def test_tables_header_is_what_the_generator_writes():
    """tools/gen_glibc_tables.py re-reads libm.so.6 and must reproduce the committed header (skipped on another libm build)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_glibc_tables.py"), "--check"], capture_output=True, text=True)
    if r.returncode not in (0, 3):
        pytest.skip("this host's libm.so.6 is not the build the addresses were taken from: " + r.stderr[-300:])
    assert r.returncode == 0, r.stdout[-300:]
