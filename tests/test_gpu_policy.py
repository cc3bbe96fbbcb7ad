"""The global-mode group policy of the reads pipeline (csrc/pipe_reads.hip: Calib, chain_rows, the close rule in submit) checked against
things it does not measure itself and on input it was not tuned on.  tools/policy_probe.py does the measuring:

  * the policy's own figures (rd_pipe_policy_read: ns per forward row, us per time step of a group's longest chain on the decode partition)
    stay within 2x of HIP-event measurements of blocking calls on the same reads (rd_timer_*) -- idle, and (within 2.5x: the other
    process's share of the chip moves from one window to the next) with a second PROCESS loading the same GPU in the background
    (everything ~1.4-2.5x slower: the figures must follow);
  * a stream whose batches ALTERNATE between 64 reads of 4 096 samples and 6 reads of 40 960 (the longest read jumps 10x from one batch to
    the next, same samples per batch) keeps up with the steady state (the harmonic mean of the two uniform streams) -- at the metric's
    width in exact fp32, and at W = 25 in bf16x3, where round 4's "close at the partition's sequence limit" rule fell to 0.25 (6.5 M against
    32 / 22 M samples/s).  Round 5 (work-aware close rule, work-queue beam search, the busy-slot fix in open_slot, a third group slot, and --
    what removed the run-to-run spread -- the pipeline's copies as kernels on their own stream's queue instead of hipMemcpyAsync, whose shared
    copy path let the labels' copy behind a running search hold up the next group's host-to-device copies): all three streams run at the
    forward's pace, alternating / steady 0.90-1.0 (profiles/r05_policy_probe.txt).  Asserted: the verdict's 0.8 and 2x.
Streams are ~100 M samples each."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _within(a, b, factor):
    return a > 0 and b > 0 and a <= factor * b and b <= factor * a


def _figures_ok(d, factor=2.0):
    """-> (ok, what): the policy's own figures against the independent ones.  Forward pace within `factor`; the chain pace at m waves per
    SIMD of the partition between the lone chain's pace / factor and factor x (1 + (m - 1) / 2) x that pace -- a chain that shares its SIMD
    with one or two other chains steps up to two or three times slower than alone (measured: W = 25, bf16x3, m = 3: 4.3 ... 10 us against a lone
    chain's 2.5), and the independent figure is the lone chain's."""
    ind = d["independent_long"]
    pol = d["policy_after_alternating"]
    ns = pol[1]["ns_per_row"]
    # (the policy keeps the smallest of its last five windows: on the low side of the blocking calls' figure by design)
    if not _within(ns, ind["ns_per_row"], factor):
        return False, ("forward pace", ns, ind["ns_per_row"])
    paces = {m: pol[m]["us_per_step"] for m in (1, 2, 3) if pol[m]["us_per_step"] > 0}
    if not paces:
        return False, "no chain pace was measured on the partition"
    lone = ind["us_per_step"]
    bad = {m: x for m, x in paces.items() if not (lone / factor <= x <= factor * (1.0 + 0.5 * (m - 1)) * lone)}
    return (not bad), ("chain pace by waves per SIMD", paces, "lone chain", lone, "outside", bad)


def _check_figures(d, factor=2.0):
    ok, what = d.get("figures", _figures_ok(d, factor))
    assert ok, what


BAR = 0.8


class _NoLoadWorker(Exception):
    pass


def _run_tool(argv):
    """tools/policy_probe.py in a process of its own -> its JSON line.  Round 6: inside the suite's long-lived process -- which by then has created and
    pooled CU-masked streams of every partition size the earlier files used, and those are never destroyed (forward.hip: destroying one can hang the
    runtime) -- the probe's blocking calls measured 57-80 ns per forward row against 27-36 in a fresh process, reproducibly within the process; the
    figures under test are properties of a basecalling process, so they are measured in one."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "policy_probe.py")] + list(argv), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    if r.returncode == 3:
        raise _NoLoadWorker(r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "no load worker")
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def _run_probe(prec, W, load):
    d = _run_tool([prec, str(W), "1" if load else "0", "-1", "0" if load else "1"])
    for key in ("policy_after_short", "policy_after_long", "policy_after_alternating"):       # (JSON made the occupancy keys strings)
        d[key] = {int(m): v for m, v in d[key].items()}
    return d


def _probe(prec, W, load, factor=2.0):
    """One probe.  A throughput ratio below the bar WARNS and is measured a second time (a shared box, streams of a few seconds each: one slow
    repetition in twenty is noise); it only counts as a finding when it reproduces -- the second measurement AND the mean of the two must
    clear the bar (best-of-two would let a regression that fails half the time pass most runs).  Both values are printed and
    attached to the result."""
    import warnings
    d = _run_probe(prec, W, load)      # (the ragged stream runs in the two unloaded tests)
    print({k: v for k, v in d.items() if not k.startswith("policy")})
    d["ratio_runs"] = [d["alternating_over_steady"]]
    d.setdefault("ragged_over_steady", 1.0)
    d["ragged_runs"] = [d["ragged_over_steady"]]
    d["figures"] = _figures_ok(d, factor)
    if d["alternating_over_steady"] >= BAR and d["ragged_over_steady"] >= BAR and not d["figures"][0] and not load:
        # only the independent figures missed (in the full suite the first blocking calls of a bf16x3 context have come out at 57-70 ns per row
        # against 27 alone): measure THEM again, in a fresh context, and judge the policy's figures -- which stand -- against the new ones
        warnings.warn(f"figures {d['figures']} ({prec}, W = {W}): measuring the independent figures once more")
        d["independent_long_first"] = d["independent_long"]
        d["independent_long"] = _run_tool(["--independent", prec, str(W)])
        print("independent figures, second measurement:", d["independent_long"])
        d["figures"] = _figures_ok(d, factor)
    elif d["alternating_over_steady"] < BAR or d["ragged_over_steady"] < BAR or not d["figures"][0]:
        warnings.warn(f"alternating / steady = {d['alternating_over_steady']:.3f}, ragged / steady = {d['ragged_over_steady']:.3f}, bar {BAR}; figures "
                      f"{d['figures']} ({prec}, W = {W}, load = {load}): measuring once more")
        d2 = _run_probe(prec, W, load)
        print("second measurement:", {k: v for k, v in d2.items() if not k.startswith("policy")})
        d2.setdefault("ragged_over_steady", 1.0)
        runs = [d["alternating_over_steady"], d2["alternating_over_steady"]]
        rruns = [d["ragged_over_steady"], d2["ragged_over_steady"]]
        d = d2
        d["ratio_runs"], d["ragged_runs"] = runs, rruns
        # what the assertions below see: the second run, capped by the mean of the two
        d["alternating_over_steady"] = min(runs[1], sum(runs) / 2.0)
        d["ragged_over_steady"] = min(rruns[1], sum(rruns) / 2.0)
        d["figures"] = _figures_ok(d, factor)      # (the figures of the second measurement: a miss has to reproduce to count)
    return d


def test_policy_figures_and_alternating_stream_fp32_beam10():
    d = _probe("fp32", 10, False)
    _check_figures(d)
    assert d["alternating_over_steady"] >= BAR, d["ratio_runs"]
    # round 6, a third stream shape the constants were not tuned on: ragged batches (log-normal read lengths, every batch another longest read,
    # read count and plan) keep up with the steady state too
    assert d["ragged_over_steady"] >= BAR, d["ragged_runs"]
    assert min(d["samples_per_s_short"], d["samples_per_s_long"], d["samples_per_s_alternating"], d["samples_per_s_ragged"]) > 12e6      # (nothing collapsed: ~20-28 M each)


def test_alternating_stream_wide_beam_bf16x3():
    d = _probe("bf16x3", 25, False, factor=2.5)
    _check_figures(d, factor=2.5)       # (W = 25: three chains per SIMD step at 4.3-5.8 us against a lone chain's 2.5)
    assert d["alternating_over_steady"] >= BAR, d["ratio_runs"]
    assert d["ragged_over_steady"] >= BAR, d["ragged_runs"]


def test_policy_follows_a_gpu_shared_with_another_process():
    try:
        d = _probe("fp32", 10, True, factor=2.5)
    except _NoLoadWorker as e:       # (the box would not start a second GPU process: nothing to measure against)
        pytest.skip(str(e))
    # (idle figures would be 1.7 us per step and 28 ns per row against 4.5-5.1 and 42-50 measured here: 2.6x and 1.7x off.  The chain pace
    # of a lone wave came out at 2.2 against 4.5 us in one of six runs -- a window in which the other process was between launches)
    _check_figures(d, factor=2.5)
    assert d["alternating_over_steady"] >= BAR, d["ratio_runs"]
