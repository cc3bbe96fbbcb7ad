#!/usr/bin/env python3
"""The one-rank full CLI (fast5 in -> FASTA out) with its start-up timed apart from its steady state (VERDICT r3 #5c): the same steps
as radian_amd.basecall.main -- parse the artefacts, create the device context, upload, run the driver loop over a multi-read fast5,
close -- each on the clock.   usage: cli_startup.py [n_reads=16384] [len=4096]"""
import os, sys, tempfile, time
t_proc = time.time()
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import fast5, basecall, synthetic
from radian_amd.backend import Backend
t_import = time.time() - t_proc


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "in"))
    reads = synthetic.synthetic_reads(n_reads, L, seed=3)
    fast5.write_multi_fast5(os.path.join(d, "in", "r.fast5"), {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
    print(f"imports {t_import:.2f} s; {n_reads} reads x {L} in one multi-read fast5", flush=True)
    for mode, extra in (("chunk", ["--step-size", "512", "--beam-width", "10"]), ("global", ["--step-size", "128", "--beam-width", "6"])):
        out = os.path.join(d, "out_" + mode)
        os.makedirs(out)
        args = basecall.build_parser().parse_args([os.path.join(d, "in"), out, "--decode-type", mode, "--sig-model", "synthetic:1234",
                                                   "--sig-config", "none", "--rna-model", "None"] + extra)
        so = sys.stdout
        t0 = time.time()
        art = basecall.load_artifacts(args)
        t1 = time.time()
        be = Backend(args.device)
        basecall.apply_artifacts(args, be, art)
        be.sync()
        t2 = time.time()
        writer = basecall.FastaWriter(out)
        sys.stdout = open(os.devnull, "w")
        try:
            basecall.run(args, [be], writer=writer)
        finally:
            sys.stdout = so
        writer.close()
        t3 = time.time()
        be.close()
        t4 = time.time()
        tot = t4 - t0
        print(f"{mode:6s}: artefacts {t1 - t0:.2f} s | context + upload {t2 - t1:.2f} s | run {t3 - t2:.2f} s = {n_reads * L / (t3 - t2) / 1e6:.2f} M samples/s steady | "
              f"close {t4 - t3:.2f} s | whole job {tot:.2f} s = {n_reads * L / tot / 1e6:.2f} M samples/s (+ {t_import:.2f} s of imports in a fresh process)", flush=True)


if __name__ == "__main__":
    main()
