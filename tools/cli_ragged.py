#!/usr/bin/env python3
"""End-to-end CLI on reads of RAGGED lengths (every device batch needs a fresh tile plan): chunk and global mode."""
import os, sys, tempfile, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


def main():
    from radian_amd import fast5, basecall
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rng = np.random.default_rng(4)
    lens = rng.integers(1500, 12000, size=n_reads)
    reads = {f"{i:08d}-0000": np.round(rng.normal(500.0, 80.0, size=int(n))).astype(np.int16) for i, n in enumerate(lens)}
    total = int(lens.sum())
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "in"))
    fast5.write_multi_fast5(os.path.join(d, "in", "r.fast5"), reads)
    for mode, extra in (("chunk", ["--step-size", "512", "--beam-width", "10"]), ("global", ["--step-size", "128", "--beam-width", "6"])):
        out = os.path.join(d, "out_" + mode)
        os.makedirs(out)
        t0 = time.time()
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.main([os.path.join(d, "in"), out, "--decode-type", mode, "--sig-model", "synthetic:1234", "--sig-config", "none",
                           "--rna-model", "None"] + extra)
        finally:
            sys.stdout = so
        dt = time.time() - t0
        print(f"ragged {mode}: {n_reads} reads, {total / 1e6:.1f} M samples in {dt:.2f}s -> {total / dt / 1e6:.2f} M samples/s end to end")


if __name__ == "__main__":
    main()
