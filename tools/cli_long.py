#!/usr/bin/env python3
"""End-to-end CLI timing on LONG reads (GPU box): n reads x L samples in one multi-read fast5 -> FASTA, global decode at the
reference's defaults (step 128, beam 6, no LM), for several --gpu-batch-windows.  A read's beam search is one serial chain
(L steps x ~1.7 us), so a device batch must hold enough rows for its forward to cover the longest chain.
usage: cli_long.py [n_reads=96] [L=100000] [batch windows | auto [:device contexts] ...]"""
import os, sys, tempfile, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import fast5, basecall, synthetic


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    sizes = [a for a in sys.argv[3:]] or ["4096", "16384", "auto"]
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "in"))
    reads = synthetic.synthetic_reads(n_reads, L, seed=3)
    fast5.write_multi_fast5(os.path.join(d, "in", "r.fast5"), {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
    for sz in sizes:
        out = tempfile.mkdtemp(prefix="out_" + sz + "_", dir=d)
        extra = [] if sz.startswith("auto") else ["--gpu-batch-windows", sz.split(":")[0]]
        if ":" in sz:                      # "<size>:<device contexts>"
            extra += ["--device-contexts", sz.split(":")[1]]
        t0 = time.time()
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.main([os.path.join(d, "in"), out, "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", "None"] + extra)
        finally:
            sys.stdout = so
        dt = time.time() - t0
        print(f"--gpu-batch-windows {sz}: {dt:.2f}s -> {n_reads * L / dt / 1e6:.2f} M samples/s end to end", flush=True)


if __name__ == "__main__":
    main()
