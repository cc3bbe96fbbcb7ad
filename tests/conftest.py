import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 10          # every run prints its ten slowest tests: the GPU suite has a 900-s budget on the driver's box


# GPU files that measure (throughput ratios, self-measured policy figures), rehearse multi-process start-up or replay fuzz slices run AFTER
# the parity files: under the driver's `pytest -x` a noisy measurement can then no longer hide a parity result.
_LATE = ("test_gpu_two_ranks", "test_gpu_standin_rccl", "test_gpu_fuzz", "test_gpu_policy")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _LATE.index(mod) + 1 if mod in _LATE else 0
    items.sort(key=rank)          # (stable: files and tests keep their order inside each class)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds oracle/libradian_oracle.so with gcc if missing."""
    from oracle import oracle as orc
    orc.build()
    return orc


def glibc_math_check_binary(force=False):
    """tests/build/glibc_math_check: radian_amd/csrc/glibc_math.h compiled for the host, compared with the running libm
    (tests/glibc_math_check.c).  Built with gcc when missing; None when it cannot be built."""
    import shutil
    import subprocess
    out = os.path.join(ROOT, "tests", "build", "glibc_math_check")
    src = os.path.join(ROOT, "tests", "glibc_math_check.c")
    deps = [src, os.path.join(ROOT, "radian_amd", "csrc", "glibc_math.h"), os.path.join(ROOT, "radian_amd", "csrc", "glibc_tables.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    if shutil.which("gcc") is None:
        return out if os.path.exists(out) else None
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", out, src, "-lm"])
    return out


@pytest.fixture(scope="session")
def host_libm_is_glibc235_fma():
    """True when this host's exp / log / log1p are bit-identical to glibc_math.h (glibc 2.35, x86-64 FMA + AVX2 variant) on a
    million arguments each: then the oracle's scores are what the beam search's "glibc" arithmetic mode reproduces."""
    import subprocess
    exe = glibc_math_check_binary()
    if exe is None:
        return False
    return subprocess.run([exe, "1"], capture_output=True).returncode == 0
