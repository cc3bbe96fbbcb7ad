#!/usr/bin/env python3
"""End-to-end CLI timing on a synthetic multi-read fast5 (GPU box): fast5 in -> FASTA out, both decode types."""
import os, sys, tempfile, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import fast5, basecall, synthetic

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    d = tempfile.mkdtemp()
    reads = synthetic.synthetic_reads(n_reads, 4096, seed=3)
    t0 = time.time()
    fast5.write_multi_fast5(os.path.join(d, "in", "r.fast5") if os.makedirs(os.path.join(d, "in")) is None else "", {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
    print(f"wrote {n_reads} reads in {time.time()-t0:.1f}s")
    for mode, extra in (("chunk", ["--step-size", "512", "--beam-width", "10"]), ("global", ["--step-size", "128", "--beam-width", "6"]),
                        ("global", ["--step-size", "512", "--beam-width", "10"])):
        out = os.path.join(d, "out_" + mode + extra[1])
        os.makedirs(out)
        t0 = time.time()
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.main([os.path.join(d, "in"), out, "--decode-type", mode, "--sig-model", "synthetic:1234", "--sig-config", "none",
                           "--rna-model", "None", "--gpu-batch-windows", "4096"] + extra)
        finally:
            sys.stdout = so
        dt = time.time() - t0
        print(f"{mode} {extra}: {dt:.2f}s -> {n_reads*4096/dt/1e6:.2f} M samples/s end to end ({n_reads/dt:.0f} reads/s)")



if __name__ == "__main__":   # the CLI's stitch workers are spawned interpreters that re-import __main__
    main()
