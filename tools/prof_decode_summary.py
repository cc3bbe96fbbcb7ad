#!/usr/bin/env python3
"""gpurun_out/profdec_<tag> -> profiles/<tag>_beam_search_pmc.json: per-time-step counters of beam_search_kernel, this round's
decoder vs round 1's, same inputs.  usage: tools/prof_decode_summary.py gpurun_out/profdec_r02 r02"""
import csv, glob, json, re, sys
base, tag = sys.argv[1].rstrip("/"), sys.argv[2]
out = {}
for tr in sorted(glob.glob(base + "/trace_*.log")):
    name = re.search(r"trace_(.*)\.log", tr).group(1)
    m = re.search(r"timesteps_per_launch=(\d+)", open(tr).read())
    if not m:
        continue
    steps = int(m.group(1))
    n = int(name.split("_")[1])
    ent = {"timesteps_per_launch": steps, "waves": n}
    st = glob.glob(f"{base}/trace_{name}/*/*_kernel_stats.csv")
    for r in csv.DictReader(open(st[0])) if st else []:
        if "beam_search" in r["Name"]:
            ent["avg_launch_us"] = float(r["AverageNs"]) / 1e3
            ent["us_per_step_per_wave"] = float(r["AverageNs"]) / 1e3 / 1024 if n <= 1024 else None
            ent["timesteps_per_s"] = steps / (float(r["AverageNs"]) * 1e-9)
    pm = glob.glob(f"{base}/pmc_{name}/*/*_counter_collection.csv")
    acc = {}
    for r in csv.DictReader(open(pm[0])) if pm else []:
        if "beam_search" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in acc.items():
        ent[c + "_per_timestep"] = sum(v) / len(v) / steps
    if "SQ_WAVE_CYCLES_per_timestep" in ent:
        wc = ent["SQ_WAVE_CYCLES_per_timestep"]
        ent["active_inst_over_wave_cycles"] = ent.get("SQ_ACTIVE_INST_ANY_per_timestep", 0) / wc
        ent["wait_any_over_wave_cycles"] = ent.get("SQ_WAIT_ANY_per_timestep", 0) / wc
    out[name] = ent
json.dump({"round": tag, "how": "tools/prof_decode.sh: rocprofv3 --kernel-trace --stats and a separate --pmc pass of tools/decode_prof_run.py <n windows> <beam width> <soft head>; "
           "'old' = round 1's decode.hip linked into the current library; counters divided by the launch's time steps (SQ_* cycle counters in quad-cycles, SQ_INSTS_* in instructions)",
           "configs": out}, open(f"profiles/{tag}_beam_search_pmc.json", "w"), indent=1)
for k, v in out.items():
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
