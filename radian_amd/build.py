"""Build libradian_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libradian_hip.so")
SOURCES = ["api.hip", "plan.hip", "pipe_reads.hip", "forward.hip", "decode.hip", "decode_wide.hip", "assemble.hip", "preprocess.hip", "stitch.hip", "lmjson.hip", "fast5.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", h) for h in ("radian_hip.h", "radian_hip_diag.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: compile with extra -D options into another file (one-off measurements: the sources hold no conditional on an RD_ macro,
    tests/test_abi_cpu.py); the product build uses neither."""
    if not force and not _stale() and out is None:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o") if out is None else os.path.basename(out) + "." + src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function",
               "-I/opt/rocm/include", *[f"-D{d}" for d in defines], "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs + ["-ldl", "-lz"]
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[2:] for a in sys.argv[1:] if a.startswith("-o")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else None))
