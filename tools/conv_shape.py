"""fp32 conv workgroup shapes (rd_set_conv_shape): bitwise equality of the probabilities and of streamed labels, then the bench's
roofline figures for each shape.  usage: python tools/conv_shape.py"""
import json, os, subprocess, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from radian_amd import Backend, synthetic, weights

be = Backend(0)
be.load_weights(weights.synthetic_weights(seed=1234))
reads = synthetic.synthetic_reads(64, 4096, seed=1000)
win = synthetic.reads_to_windows(reads, 1024, 512)[0]
ragged = [synthetic.synthetic_reads(1, n, seed=n)[0] for n in (1, 31, 33, 700, 1024, 1500, 4097, 9000)]
out = {}
for shape in (0, 1):
    be.set_conv_shape(shape)
    p = be.forward(win)
    lab, st = be.basecall_raw_chunk(ragged, 4, 1024, 512, 10)
    glab, _ = be.basecall_raw_global(ragged, 4, 1024, 128, 6, False)
    out[shape] = (p, lab, glab)
same = np.array_equal(out[0][0], out[1][0]) and all(np.array_equal(a, b) for x, y in zip(out[0][1], out[1][1]) for a, b in zip(x, y)) \
    and all(np.array_equal(a, b) for a, b in zip(out[0][2], out[1][2]))
print("probabilities and labels bit-identical across shapes:", same)
be.close()
for shape in (0, 1, 0, 1):
    r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--conv-shape", str(shape), "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    rf = d["roofline"]
    print(f"shape {shape}: headline {d['value'] / 1e6:.2f} M samples/s, conv {rf['avg_launch_ms']:.4f} ms per launch = {rf['frac']:.3f} of peak, pipeline_frac {rf['pipeline_frac']:.3f}", flush=True)
