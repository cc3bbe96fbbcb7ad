import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
def main():
    from radian_amd import fast5, basecall, synthetic
    n_reads = 16384
    d = tempfile.mkdtemp(); os.makedirs(d + "/in")
    reads = synthetic.synthetic_reads(n_reads, 4096, seed=3)
    fast5.write_multi_fast5(d + "/in/r.fast5", {f"{i:08d}-0000": reads[i] for i in range(n_reads)})
    for ctxs in (1, 2, 3):
        for gbw in (2048, 4096, 8192):
            o = f"{d}/out_{ctxs}_{gbw}"; os.makedirs(o)
            t0 = time.time(); so = sys.stdout; sys.stdout = open(os.devnull, "w")
            try:
                basecall.main([d + "/in", o, "--decode-type", "global", "--sig-model", "synthetic:1234", "--sig-config", "none", "--rna-model", "None",
                               "--device-contexts", str(ctxs), "--gpu-batch-windows", str(gbw)])
            finally:
                sys.stdout = so
            dt = time.time() - t0
            print(f"global contexts={ctxs} batch={gbw}: {n_reads * 4096 / dt / 1e6:.2f} M samples/s")
if __name__ == "__main__":
    main()
