import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
def main():
    from radian_amd import fast5, basecall, synthetic
    d = tempfile.mkdtemp(); os.makedirs(d + "/in")
    reads = synthetic.synthetic_reads(600, 4096, seed=3)
    fast5.write_multi_fast5(d + "/in/r.fast5", {f"{i:08d}": reads[i] for i in range(600)})
    outs = {}
    for g in (1, 2):
        o = f"{d}/out{g}"; os.makedirs(o)
        so = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            basecall.main([d + "/in", o, "--decode-type", "chunk", "--step-size", "512", "--beam-width", "10", "--sig-model", "synthetic:1234",
                           "--sig-config", "none", "--rna-model", "None", "--gpus", str(g), "--queue-block", "64"])
        finally:
            sys.stdout = so
        outs[g] = open(o + "/reads-0.fasta").read()
    print("2-rank (one GPU, file fallback) FASTA == single:", outs[1] == outs[2], len(outs[1]))
if __name__ == "__main__":
    main()
