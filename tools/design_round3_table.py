#!/usr/bin/env python3
"""Rewrites the round-3 measurement table of DESIGN.md (between the r03-table markers) from profiles/r03_bench.json, so that the
table always is the committed bench output.  usage: python tools/design_round3_table.py"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(R, "profiles", "r03_bench.json")))
r = d["roofline"]
M = lambda k, f="value": d[k][f] / 1e6
rows = f"""| quantity (one MI355X, driver-timed: `profiles/r03_bench.json`) | round 2 | round 3 |
|---|---|---|
| headline, exact fp32, chunk W = 10, 64-read steps | 26.4 M samples/s | **{d['value']/1e6:.1f} M** ({d['ms_per_step']:.2f} ms per step); `roofline.frac` {r['frac']:.3f} ({r['avg_launch_ms']:.4f} ms per conv launch), `pipeline_frac` {r['pipeline_frac']:.3f} |
| `secondary_soft_head`: the headline's step on soft rows (~{d['secondary_soft_head']['mean_bases_per_window']:.0f} bases per window instead of ~5) | — | {M('secondary_soft_head'):.1f} M |
| `secondary_global_lm`: configs[3] geometry, W = 10, 4^11-row LM, 64-read steps | 19.5 M (two contexts on two host threads) | **{M('secondary_global_lm'):.1f} M in ONE context** (`rd_pipe_submit_reads_global`, groups of two steps on the decode partition, 4 CUs per XCD); two contexts unpipelined {d['secondary_global_lm']['two_contexts_unpipelined']/1e6:.1f} M |
| `secondary_global_lm_soft_head`: the same with the soft head (~{d['secondary_global_lm_soft_head']['mean_bases_per_read']:.0f} bases per read, the gate fires) | — | {M('secondary_global_lm_soft_head'):.1f} M ({d['secondary_global_lm_soft_head']['two_contexts_unpipelined']/1e6:.1f} M) |
| `secondary_cfg5_w25_ctx256_f16` (configs[4]) | 15.8 M | {M('secondary_cfg5_w25_ctx256_f16'):.1f} M (21.2 M with a partition of 5 CUs per XCD: §4.11) |
| `secondary_e2e_raw`: host int16 → … → strings, {d['secondary_e2e_raw']['reads']} uniform reads, chunk W = 10 | 25.4 M (two contexts, 4 stitch processes) | {M('secondary_e2e_raw'):.1f} M (one context) |
| `secondary_e2e_raw_ragged`: the same on log-normal read lengths (1.5 k … 60 k, median 9 k): a plan per batch | — | {M('secondary_e2e_raw_ragged'):.1f} M |
| `secondary_e2e_raw_soft_head`: configs[2] end to end on soft rows | (3.6 M with the Python stitch, measured this round) | **{M('secondary_e2e_raw_soft_head'):.1f} M** (`rd_stitch_chunk`) |
| `secondary_reference_defaults`: the reference's defaults (global, step 128, beam 6, 12-mer LM), {d['secondary_reference_defaults']['reads']} ragged reads, soft head, ~{d['secondary_reference_defaults']['mean_bases_per_read']:.0f} bases per read | — (builder-run 25 M on uniform reads, saturated rows) | **{M('secondary_reference_defaults'):.1f} M** in a {d['secondary_reference_defaults']['seconds']:.1f}-s job (23.2 M on a quarter of the reads, a 2.1-s job of which the fill and drain of the groups were a third; 17.5 M there before the two-sequence wave and the table-lookup label strings) |
| `secondary_long_reads`: reference defaults, {d['secondary_long_reads']['reads']} reads × 100 000 samples | 19–26 M (builder-run, two contexts, 50 GB batches) | **{M('secondary_long_reads'):.1f} M** in a {d['secondary_long_reads']['seconds']:.1f}-s job (4096-unit batches, groups on the decode partition; 22.0 M on a quarter of the reads: the last group's 0.2-s chains drain alone) |
| `secondary_bf16x3` / `secondary_f16x3` | 33.9 / 58 M | {M('secondary_bf16x3'):.1f} / {M('secondary_f16x3'):.1f} M |
| beam search alone, 512 windows, W = 10: `decode_timesteps_per_s` (glibc arithmetic) and `decode_hbm_frac` = × 20 B ÷ 8 TB/s | 243 M | {r['decode_timesteps_per_s']/1e6:.0f} M; {r['decode_hbm_frac']:.1e} of HBM — issue / latency bound, as SURVEY §8d predicted |
| `cpu_baseline` (oracle port, {d['cpu_baseline']['cores']} threads / one thread) | 83 k / 5.3 k samples/s | {d['cpu_baseline']['value']/1e3:.0f} k / {d['cpu_baseline']['single_thread']['value']/1e3:.1f} k |
"""
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- r03-table-begin[^\n]*-->\n).*?(<!-- r03-table-end -->)", lambda m: m.group(1) + rows + m.group(2), s, flags=re.S)
open(p, "w").write(s)
print(rows)
