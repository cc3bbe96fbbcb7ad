"""The bench line's contract, checked on the lines this round committed under profiles/ (no GPU here; the driver runs bench.py itself): the keys the
driver and the judge read, their types and mutual consistency -- N = 1 (`profiles/r06_bench.json`) and the four-rank rehearsal on one GPU
(`profiles/r06_bench_4ranks_standin_one_gpu.json`)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def _common(d, n):
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] in baseline["metric"] and d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == n and d["data"] == "synthetic" and d["dtype"] == "f32" and d["vs_baseline"] is None      # (BASELINE.md publishes no number)
    assert "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["config"]["chunk_len"] == 1024 and d["config"]["beam_width"] == 10 and d["config"]["batch_windows"] == 512
    assert d["steps"] > 0 and d["warmup"] >= 0 and d["ms_per_step"] > 0
    # value = whole-job samples / the reported region's time, and that region is the median of value_runs
    per_step = d["config"]["samples_per_step_per_gpu"] * n
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    runs = sorted(d["value_runs"])
    assert len(runs) >= 3 and d["value"] == runs[(len(runs) - 1) // 2]


def test_single_gpu_line():
    d = _line("r06_bench.json")
    _common(d, 1)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.5 < r["frac"] < 1.0 and r["traffic"] and r["traffic"] > 1e8
    bv = r["by_variant"]
    assert set(bv) == {"relu", "res_ident", "res_match", "head"}
    # frac is launch-weighted over the conv variants: total FLOPs / total time
    fl = sum(bv[k]["launches"] * bv[k]["rows_per_launch"] * 393216 for k in ("relu", "res_ident", "res_match"))
    ms = sum(bv[k]["launches"] * bv[k]["avg_ms"] for k in ("relu", "res_ident", "res_match"))
    assert abs(fl / (ms * 1e-3) / 1e12 / r["peak"] - r["frac"]) < 1e-3
    assert min(bv[k]["frac"] for k in bv) < r["frac"] < max(bv[k]["frac"] for k in bv)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["gpu_over_cpu"] - d["value"] / c["value"]) < 1e-6 and d["gpu_over_cpu"] >= 50.0          # north_star: >= 50x the CPU path
    w = d["secondary_windowed"]
    assert w["model_rows_per_step"] == 512 * 1024 and w["flop_per_row"] == 4394240 and abs(w["frac"] - w["value"] / w["bound_samples_per_s"]) < 1e-9
    assert abs(w["bound_samples_per_s"] - 157.3e12 / (4394240 * 2)) < 1.0
    pk = d["secondary_decode_only_peaky"]["by_width"]
    assert set(pk) == {"6", "10", "25"} and all(v["timesteps_per_s"] > 1e8 and 0.0 < v["valu_issue_frac"] < 1.0 for v in pk.values())
    for key in ("secondary_forward_only", "secondary_global_lm", "secondary_e2e_raw", "secondary_e2e_fast5_to_fasta", "secondary_e2e_fast5_to_fasta_global_lm",
                "secondary_drna_like_head", "secondary_soft_head", "secondary_reference_defaults"):
        assert d[key]["value"] > 1e7 and d[key]["unit"] == "samples/s", key


def test_four_rank_rehearsal_line():
    d = _line("r06_bench_4ranks_standin_one_gpu.json")
    _common(d, 4)
    assert d["startup_comm"] == "rccl" and d["rccl_nranks"] == 4 and len(d["ms_per_step_per_rank"]) == 4
    assert abs(d["ms_per_step"] - max(d["ms_per_step_per_rank"])) < 1e-9                  # max over ranks
    for key, mode in (("secondary_e2e_fast5_to_fasta", "chunk"), ("secondary_e2e_fast5_to_fasta_global_lm", "global")):
        leg = d[key]
        assert "skipped" not in leg and leg["n_ranks"] == 4 and f"--decode-type {mode}" in leg["cli"]
        assert leg["reads"] == leg["records_written"] == 4 * 4 * 4096 and leg["samples"] == leg["reads"] * 4096
        assert abs(leg["value"] - leg["samples"] / leg["seconds"]) < 1e-3 and leg["value_to_merged_fasta"] <= leg["value"] * 1.001
        pr = leg["per_rank"]
        assert [p["rank"] for p in pr] == [0, 1, 2, 3] and sum(p["reads"] for p in pr) == leg["reads"] and all(p["cores"] >= 1 and p["cpu_bound"] for p in pr)
        assert leg["rccl_nranks"] == 4 and len(leg["fasta_sha256"]) == 64
