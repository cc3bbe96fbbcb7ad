#!/usr/bin/env python3
"""Chunk-mode CLI end to end on N synthetic reads with different numbers of stitch workers."""
import os, sys, tempfile, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


def main():
    from radian_amd import fast5, basecall, synthetic
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "in"))
    reads = synthetic.synthetic_reads(n_reads, 4096, seed=3)
    fast5.write_multi_fast5(os.path.join(d, "in", "r.fast5"), {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
    for workers in (0, 4, 8):
        out = os.path.join(d, f"out{workers}")
        os.makedirs(out)
        t0 = time.time()
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.main([os.path.join(d, "in"), out, "--decode-type", "chunk", "--sig-model", "synthetic:1234", "--sig-config", "none",
                           "--rna-model", "None", "--step-size", "512", "--beam-width", "10", "--stitch-workers", str(workers)])
        finally:
            sys.stdout = so
        dt = time.time() - t0
        print(f"chunk, {workers} stitch workers: {dt:.2f}s -> {n_reads * 4096 / dt / 1e6:.2f} M samples/s end to end ({n_reads / dt:.0f} reads/s)")


if __name__ == "__main__":
    main()
