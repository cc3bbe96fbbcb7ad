#!/bin/bash
# Effective shader clock and MFMA-busy fraction of the conv launches for every experiment build under radian_amd/variants/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for so in radian_amd/variants/lib_*.so; do
  tag=$(basename $so .so)
  export RADIAN_HIP_LIB=$R/$so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/clock_$tag -- python3 tools/layer_times.py > gpurun_out/clock_$tag.log 2>&1 || exit 1
  python3 - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
p = glob.glob(f"gpurun_out/clock_{tag}/*/*_counter_collection.csv")[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(p)):
    if "tcn_gemm_kernel<4, 3, 0" in r["Kernel_Name"] or "tcn_gemm_kernel<4, 3, 1" in r["Kernel_Name"]:
        k = int(r["Dispatch_Id"])
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
        d[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[k]["grid"] = int(r["Grid_Size"])
by = collections.defaultdict(list)
for k, v in d.items():
    by[v["grid"]].append(v)
for g, vs in sorted(by.items()):
    vs = vs[len(vs) // 2:]
    ns = sum(v["ns"] for v in vs) / len(vs)
    cyc = sum(v["GRBM_GUI_ACTIVE"] for v in vs) / len(vs) / 8
    mf = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in vs) / len(vs) / 1024
    print(f"{tag} grid {g // 256} tiles: {ns / 1e3:.1f} us, clock {cyc / ns:.3f} GHz, mfma busy {mf / cyc:.3f}")
PY
done
