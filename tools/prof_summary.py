#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory into profiles/<tag>_{kernel_stats.csv,pmc_summary.json} and profiles/traffic.json.
usage: tools/prof_summary.py <gpurun_out/prof_TAG> <round tag>"""
import collections, csv, glob, json, os, shutil, sys

base, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]


def short(name):
    if "tcn_gemm_kernel<4, 3, 0" in name: return "conv_relu"
    if "tcn_gemm_kernel<4, 3, 1" in name: return "conv_res_ident"
    if "tcn_gemm_kernel<4, 3, 2" in name: return "conv_res_match"
    if "tcn_gemm_kernel<2" in name: return "head"
    if "beam" in name: return "beam_search"
    if "tcn_in" in name: return "conv_in"
    return None


out = {}
for f in ["pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds"]:
    p = glob.glob(base + f + "/*/*_counter_collection.csv")[0]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k:
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        for c, x in v.items():
            out.setdefault(k, {})[c] = {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)}
stats = glob.glob(base + "trace/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")   # one forward lane: launches back to back
two = glob.glob(base + "trace2/*/*_kernel_stats.csv")
if two:
    shutil.copy(two[0], f"profiles/{tag}_kernel_stats_default_2lanes.csv")   # default command: launches of two batches overlap
dur = {}
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    if k:
        dur[k] = float(r["AverageNs"])
derived = {}
for k in ["conv_relu", "conv_res_ident", "conv_res_match"]:
    v = out[k]
    cyc = v["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8
    mf = v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / 1024
    derived[k] = {"avg_ns_trace": dur.get(k), "cycles_per_xcd": cyc, "mfma_busy_cycles_per_simd": mf, "mfma_util": mf / cyc,
                  "eff_clock_ghz": cyc / dur[k] if k in dur else None,
                  "hbm_bytes": (2 * v["FETCH_SIZE"]["mean_per_dispatch"] + v["WRITE_SIZE"]["mean_per_dispatch"]) * 1024}
json.dump({"round": tag, "command": "tools/prof.sh (rocprofv3 --pmc <group>, one pass per counter group; python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary --lanes 1; durations from the --kernel-trace pass of the same command)",
           "units": {"FETCH_SIZE": "KiB; gfx950 tallies 16-B-per-lane loads (incl. LDS-DMA) at half their bytes -> doubled in hbm_bytes",
                     "WRITE_SIZE": "KiB", "SQ_*": "quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles summed over SIMDs)",
                     "GRBM_GUI_ACTIVE": "cycles summed over 8 XCDs"},
           "derived": derived, "kernels": out}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
mix = (5 * derived["conv_relu"]["hbm_bytes"] + 5 * derived["conv_res_ident"]["hbm_bytes"] + derived["conv_res_match"]["hbm_bytes"]) / 11
json.dump({"round": tag, "conv_hbm_bytes_per_launch": mix,
           "how": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch from separate rocprofv3 --pmc passes, averaged over the 11 conv launches "
                  "of a forward (5 relu + 5 residual-identity + 1 residual-match); FETCH_SIZE doubled per MI355X_MICROARCH.md"},
          open("profiles/traffic.json", "w"), indent=1)
for k, v in derived.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
print("traffic mix bytes/launch", mix)
for k, v in dur.items():
    print(k, v)
