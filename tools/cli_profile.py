#!/usr/bin/env python3
import os, sys, tempfile, cProfile, pstats, io
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from radian_amd import fast5, basecall, synthetic
n_reads = 4096
d = tempfile.mkdtemp(); os.makedirs(d + "/in"); os.makedirs(d + "/out")
reads = synthetic.synthetic_reads(n_reads, 4096, seed=3)
fast5.write_multi_fast5(d + "/in/r.fast5", {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
so = sys.stdout; sys.stdout = open(os.devnull, "w")
pr = cProfile.Profile(); pr.enable()
basecall.main([d + "/in", d + "/out", "--decode-type", "chunk", "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", "None",
               "--step-size", "512", "--beam-width", "10"])
pr.disable(); sys.stdout = so
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
