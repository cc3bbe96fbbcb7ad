#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures of one HIP source as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py decode.hip [-DNAME=V ...]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-I/opt/rocm/include",
                      *extra, "-c", os.path.join(R, "radian_amd", "csrc", src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").split("(")[0]
    g = r.get
    print(f"{dn:60s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>3} sgpr-spill {g('SGPRs Spill'):>3} vgpr-spill {g('VGPRs Spill'):>2} "
          f"scratch {g('ScratchSize [bytes/lane]'):>3} LDS {g('LDS Size [bytes/block]'):>6} occ {g('Occupancy [waves/SIMD]')}")
