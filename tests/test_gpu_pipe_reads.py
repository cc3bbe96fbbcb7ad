"""The reads-level pipeline inside one context (rd_pipe_submit_reads_global / rd_pipe_submit_raw_global /
rd_pipe_submit_raw_chunk / rd_pipe_progress; csrc/pipe_reads.hip): forwards of consecutive batches on rotating lanes,
per-read assembly on the lane, the beam search of a group of batches on the decode stream.

Parity: every submitted batch's labels equal (a) the unpipelined entry point's on the same input and (b) the oracle's --
normalise -> windows -> assemble (matrix_assembly.py:6-53) -> LM beam search (decode.py:100-212) of the GPU's own
probabilities -- at BASELINE configs[3]'s geometry (64 reads x 4096, global, step 512, W = 10, 4^11-row LM, 0.5 / 0.5)
with the soft head (the LM gate fires), and over ragged batches (plan rebuilt and uploaded per submit, reads shorter than
a window, the streamed / windowed boundary at step 772 / 773, MAD-zero reads, f16 logits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK, READ_LEN = 1024, 4096


def _soft_weights():
    from radian_amd import weights
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(0.05)        # soft rows: ~1000-base labelings per read, the LM gate fires
    return w


@pytest.fixture(scope="module")
def dev():
    from radian_amd import Backend
    b = Backend(0)
    b.load_weights(_soft_weights())
    yield b
    b.close()


def _ragged(rng, n, lo=1, hi=9000, special=()):
    from radian_amd import synthetic
    lens = [int(x) for x in rng.integers(lo, hi, size=n)]
    for i, v in enumerate(special):
        lens[i % n] = v
    reads = [synthetic.synthetic_reads(1, L, seed=int(rng.integers(1 << 30)))[0] for L in lens]
    return reads


def test_pipe_global_cfg3_equals_unpipelined_and_oracle(dev, oracle):
    from radian_amd import synthetic
    k, W, step = 11, 10, 512
    table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** k)
    dev.load_lm(table, k)
    try:
        batches = [list(synthetic.synthetic_reads(64, READ_LEN, seed=1000 + b)) for b in range(3)]
        ref = [dev.basecall_raw_global(b, 4, CHUNK, step, W, True, 0.5, 0.5) for b in batches]
        dev.pipe_flush()
        dev.pipe_config(4)
        dev.pipe_set_lanes(2)
        base = dev.pipe_progress(0)
        tickets = [dev.pipe_submit_raw("global", batches[i % 3], 4, CHUNK, step, W, True, 0.5, 0.5) for i in range(7)]
        # 64 x 4096 rows per submit < 96 rows x 4096 chain steps: a group closes after two submits, so by now three groups
        # have been launched and the first ones delivered without anyone waiting for them
        assert dev.pipe_progress(0) >= base + 2
        for i, t in enumerate(tickets):
            got, status = t.result()
            assert not status.any()
            bad = [r for r in range(64) if not np.array_equal(got[r], ref[i % 3][0][r])]
            assert not bad, (i, bad[:8], len(bad))
        assert dev.pipe_progress(0) == base + 7
        # the oracle on batch 0: its own normalisation and windows, the GPU's probabilities
        mats, lens = [], []
        for raw in batches[0]:
            win, pad = oracle.get_windows(oracle.mad_normalise(raw, 4), CHUNK, step)
            m = oracle.assemble_matrices(dev.forward(np.asarray(win, dtype=np.float32)), pad, step)
            mats.append(m)
            lens.append(m.shape[0])
        lens = np.asarray(lens, dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        exp = oracle.beam_search_batch(np.concatenate(mats), off, lens, W, table, 0.5, 0.5, k)
        got = tickets[0].result()[0]
        bad = [r for r in range(64) if not np.array_equal(got[r], exp[r])]
        assert not bad, (bad[:8], len(bad))
        nolm = oracle.beam_search_batch(np.concatenate(mats), off, lens, W)
        assert sum(not np.array_equal(a, c) for a, c in zip(exp, nolm)) >= 32      # the gate really fired
    finally:
        dev.load_lm(None, 0)


@pytest.mark.parametrize("step,W,logits", [(128, 6, "f32"), (512, 10, "f32"), (772, 3, "f32"), (773, 10, "f32"), (1024, 6, "f32"),
                                           (512, 25, "f16"), (300, 13, "f32")])
def test_pipe_global_ragged_equals_unpipelined(dev, step, W, logits):
    """every submit has other read lengths (the lane's plan is rebuilt and uploaded behind its previous forward); reads of
    1 .. 9000 samples incl. shorter than a window (float32 rows, no assembly), exactly a window, a multiple of the step; a
    constant read (MAD = 0 -> status 1); group size 3 on 2 and 3 lanes"""
    rng = np.random.default_rng(step * 100 + W)
    table = np.random.default_rng(1).dirichlet([0.3] * 4, size=4 ** 3)
    dev.load_lm(table, 3)
    dev.set_logits(logits)
    try:
        for lanes in (2, 3):
            batches = []
            for i in range(7):
                reads = _ragged(rng, int(rng.integers(1, 40)), special=(1, CHUNK, CHUNK - 1, CHUNK + 1, 4 * step, CHUNK + 3 * step))
                if i % 2:
                    reads[len(reads) // 2] = np.full(777, 5, dtype=np.int16)
                batches.append(reads)
            ref = [dev.basecall_raw_global(b, 4, CHUNK, step, W, True, 0.2, 0.9) for b in batches]
            dev.pipe_flush()
            dev.pipe_config(3)
            dev.pipe_set_lanes(lanes)
            tickets = [dev.pipe_submit_raw("global", b, 4, CHUNK, step, W, True, 0.2, 0.9) for b in batches]
            dev.pipe_flush()
            for i, t in enumerate(tickets):
                assert t.done()
                got, status = t.result()
                assert np.array_equal(status, ref[i][1]), (i, status, ref[i][1])
                bad = [r for r in range(len(batches[i])) if status[r] == 0 and not np.array_equal(got[r], ref[i][0][r])]
                assert not bad, (step, W, lanes, i, bad[:8], len(bad))
            assert any(s.any() for _, s in ref)
    finally:
        dev.set_logits("f32")
        dev.load_lm(None, 0)
        dev.pipe_config(4)
        dev.pipe_set_lanes(2)


@pytest.mark.parametrize("step,W", [(512, 10), (128, 6), (1000, 25)])
def test_pipe_raw_chunk_ragged_equals_unpipelined(dev, step, W):
    rng = np.random.default_rng(step + W)
    batches = [_ragged(rng, int(rng.integers(1, 24)), hi=6000, special=(1, CHUNK, CHUNK + step)) for _ in range(6)]
    batches.append(batches[2])          # the same lengths again on the same lane two submits later: a plan hit
    batches.append(batches[2])
    ref = [dev.basecall_raw_chunk(b, 4, CHUNK, step, W) for b in batches]
    dev.pipe_flush()
    dev.pipe_config(3)
    tickets = [dev.pipe_submit_raw("chunk", b, 4, CHUNK, step, W) for b in batches]
    for i, t in enumerate(tickets):
        got, status = t.result()
        assert np.array_equal(status, ref[i][1])
        for r in range(len(batches[i])):
            assert len(got[r]) == len(ref[i][0][r])
            bad = [w for w in range(len(got[r])) if not np.array_equal(got[r][w], ref[i][0][r][w])]
            assert not bad, (step, W, i, r, bad[:5])
    dev.pipe_config(4)


def test_pipe_resident_global_and_progress_contract(dev):
    """rd_pipe_submit_reads_global on normalised reads resident in HBM; rd_pipe_progress: non-blocking with wait_for 0,
    blocking (and closing the open group) with wait_for > 0, an error beyond what was submitted"""
    from radian_amd import RadianHipError, synthetic
    n, W, step = 48, 6, 128
    raws = synthetic.synthetic_reads(n, READ_LEN, seed=77)
    norm = np.stack([synthetic.mad_normalise(r, 4) for r in raws]).astype(np.float32)
    ref = dev.basecall_reads_global(list(norm), CHUNK, step, W, False)
    d = dev.dev_alloc(norm.nbytes)
    dev.h2d(d, norm)
    off = np.arange(n + 1, dtype=np.int64) * READ_LEN
    lab_off = np.ascontiguousarray(off[:-1])
    try:
        dev.pipe_flush()
        dev.pipe_config(8)
        base = dev.pipe_progress(0)
        outs = [(np.zeros(n * READ_LEN + 1, dtype=np.uint8), np.full(n, -1, dtype=np.int32)) for _ in range(3)]
        # 48 x 4096 rows per submit >= 96 x 4096: every submit is a group of its own
        for lab, ln in outs:
            dev.pipe_submit_reads_global(d, off, n, CHUNK, step, W, False, 0.0, 0.0, lab, lab_off, ln)
        assert dev.pipe_progress(base + 1) >= base + 1
        assert outs[0][1].min() >= 0
        with pytest.raises(RadianHipError):
            dev.pipe_progress(base + 4)
        dev.pipe_flush()
        assert dev.pipe_progress(0) == base + 3
        for lab, ln in outs:
            for r in range(n):
                assert np.array_equal(lab[off[r]: off[r] + ln[r]], ref[r]), r
        # an unpipelined call between submits and their delivery shares the trie workspace: results stay right
        lab, ln = np.zeros(n * READ_LEN + 1, dtype=np.uint8), np.full(n, -1, dtype=np.int32)
        dev.pipe_submit_reads_global(d, off, n, CHUNK, step, W, False, 0.0, 0.0, lab, lab_off, ln)
        again = dev.basecall_reads_global(list(norm), CHUNK, step, W, False)
        dev.pipe_flush()
        assert all(np.array_equal(a, b) for a, b in zip(again, ref))
        assert all(np.array_equal(lab[off[r]: off[r] + ln[r]], ref[r]) for r in range(n))
    finally:
        dev.pipe_flush()
        dev.dev_free(d)
        dev.pipe_config(4)


def test_cli_driver_pipelined_equals_blocking(tmp_path):
    """radian_amd.basecall.run through the in-context pipeline (default) and with --no-pipeline: same results, same order,
    both decode types, ragged reads in small device batches (several groups in flight)"""
    from radian_amd import Backend, basecall
    rng = np.random.default_rng(5)
    reads = _ragged(rng, 90, lo=200, hi=7000)

    class R:
        def __init__(self, i, s):
            self.read_id, self._s = f"r{i:04d}", s

        def get_raw_data(self):
            return self._s
    for mode in ("global", "chunk"):
        res = {}
        for flag in ([], ["--no-pipeline"]):
            args = basecall.build_parser().parse_args(["-", "-", "--decode-type", mode, "--step-size", "512", "--beam-width", "6", "--rna-model", "None",
                                                       "--gpu-batch-windows", "64"] + flag)
            args._lm_loaded = False
            bes = [Backend(0) for _ in range(basecall.n_contexts(args))]
            try:
                bes[0].load_weights(_soft_weights())
                for b in bes[1:]:
                    b.clone_artifacts_from(bes[0])
                import contextlib, io
                with contextlib.redirect_stdout(io.StringIO()):
                    res[bool(flag)] = basecall.run(args, bes, reads=iter([R(i, s) for i, s in enumerate(reads)]), writer=None)
            finally:
                for b in bes:
                    b.close()
        assert len(res[False]) == 90 and res[False] == res[True], mode


def test_pipe_mode_and_partition_changes_between_batches(dev):
    """Two bugs tests/fuzz_pipe.py found in the first version of the pipeline, as regression cases: (1) a group slot first grown by
    a CHUNK-mode group had no assembled-matrix buffer when a GLOBAL-mode group used it next (out-of-bounds write of the assembly
    kernel); (2) a batch of few reads (partitioned forward lane) and a batch of many reads (plain lane) of the same logical lane ran
    on different streams but shared the lane's signal / descriptor buffers.  Alternating decode types and batch sizes, every
    batch against the blocking call."""
    rng = np.random.default_rng(99)
    big = _ragged(rng, 200, lo=300, hi=1500)          # many reads: every CU, whole-chip beam search
    small = _ragged(rng, 9, lo=2000, hi=9000)         # few reads: partitioned lane + partition decode
    plan = [("chunk", big), ("global", small), ("global", big), ("global", small), ("chunk", small), ("global", big), ("global", small),
            ("global", big)]
    ref = [dev.basecall_raw_global(r, 4, CHUNK, 256, 6, False) if m == "global" else dev.basecall_raw_chunk(r, 4, CHUNK, 256, 6) for m, r in plan]
    dev.pipe_flush()
    dev.pipe_config(2)
    dev.pipe_set_lanes(2)
    tickets = [dev.pipe_submit_raw(m, r, 4, CHUNK, 256, 6, False) for m, r in plan]
    for (m, r), t, (exp, st_exp) in zip(plan, tickets, ref):
        got, st = t.result()
        assert np.array_equal(st, st_exp)
        if m == "global":
            assert all(np.array_equal(a, b) for a, b in zip(got, exp))
        else:
            assert all(len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(got, exp))
    dev.pipe_config(4)


def test_pipe_policy_follows_precision_changes_mid_stream(dev):
    """Round 4: the group policy measures its own forward rate and chain pace (pipe_reads.hip, Calib) per matrix-product mode, beam
    width and arithmetic.  A stream of global-mode batches whose precision, arithmetic and width change between submits -- enough
    submits for every measurement window to fill and be used -- delivers what the blocking calls deliver in the same modes."""
    rng = np.random.default_rng(123)
    small = _ragged(rng, 6, lo=3000, hi=7000)
    big = _ragged(rng, 120, lo=300, hi=1200)
    plan = []
    for rep in range(3):
        for prec, math, W in (("fp32", "glibc", 6), ("f16x3", "glibc", 6), ("bf16x3", "fast", 10), ("fp32", "fast", 10), ("f16x3", "glibc", 10)):
            for reads in (small, small, big, small, small, big, small, small, small, small):
                plan.append((prec, math, W, reads))
    ref_cache = {}
    dev.pipe_flush()
    dev.pipe_config(2)
    dev.pipe_set_lanes(2)
    tickets = []
    try:
        for prec, math, W, reads in plan:
            key = (prec, math, W, id(reads))
            if key not in ref_cache:
                dev.pipe_flush()
                dev.set_precision(prec)
                dev.set_decode_math(math)
                ref_cache[key] = dev.basecall_raw_global(reads, 4, CHUNK, 256, W, False)
        cur = None
        for prec, math, W, reads in plan:
            if cur != (prec, math):
                dev.set_precision(prec)        # (context state: takes effect for the submits that follow; no flush)
                dev.set_decode_math(math)
                cur = (prec, math)
            tickets.append(dev.pipe_submit_raw("global", reads, 4, CHUNK, 256, W, False))
        for (prec, math, W, reads), t in zip(plan, tickets):
            got, st = t.result()
            exp, st_exp = ref_cache[(prec, math, W, id(reads))]
            assert np.array_equal(st, st_exp)
            assert all(np.array_equal(a, b) for a, b in zip(got, exp)), (prec, math, W, len(reads))
    finally:
        dev.pipe_flush()
        dev.set_precision("fp32")
        dev.set_decode_math("glibc")
        dev.pipe_config(4)


@pytest.mark.parametrize("W", [6, 10, 25])
def test_group_larger_than_its_decode_partition_delivers_the_blocking_calls_labels(dev, W):
    """Round 5: a global-mode group that gathers more sequences than its decode partition keeps resident (W = 10: 384) before its forward
    rows cover its longest chain (W = 6: 768 two to a wave, W = 25: 192) stays on the partition and is searched through the work queue (beam_search_queue_kernel: resident
    workgroups take the sequences longest-first).  Thirty batches of 60 short reads behind one 30 000-sample read: the crossing batch
    decides (work keeps up with the forward), rd_pipe_stats counts the queue launch, and every batch's labels are the blocking call's --
    with an LM whose gate fires, in both arithmetics' default."""
    rng = np.random.default_rng(2025)
    table = rng.dirichlet([0.3] * 4, size=4 ** 3)
    dev.pipe_flush()
    dev.set_precision("fp32")
    dev.set_decode_math("glibc")
    dev.load_lm(table, 3)
    batches = []
    for b in range(30):
        reads = _ragged(rng, 60, lo=900, hi=1300)
        if b == 0:
            reads = [np.round(rng.normal(500, 80, size=30000)).astype(np.int16)] + reads
        batches.append(reads)
    try:
        ref = [dev.basecall_raw_global(r, 4, CHUNK, 512, W, True, 0.3, 2.0) for r in batches]
        before = dev.pipe_stats()
        tickets = [dev.pipe_submit_raw("global", r, 4, CHUNK, 512, W, True, 0.3, 2.0) for r in batches]
        got = [t.result() for t in tickets]
        after = dev.pipe_stats()
        assert after["queue_launches"] > before["queue_launches"], (before, after)
        for (gl, gs), (rl, rs) in zip(got, ref):
            assert np.array_equal(gs, rs) and len(gl) == len(rl) and all(np.array_equal(a, b) for a, b in zip(gl, rl))
    finally:
        dev.pipe_flush()
        dev.load_lm(None, 0)


@pytest.mark.parametrize("W", [10, 25, 100])
def test_beam_search_workspace_budget_cuts_launches_without_changing_results(dev, W):
    """a launch's trie is W x ALL its rows (20-24 B per node, + decode_wide's scratch per sequence) -- a big group at a
    wide beam asked for hundreds of GB and died with RD_ERR_NOMEM mid-job.  Launches beyond the context's budget now run as several
    runs of sequences that share the workspace (rd_plan_trie_runs).  With a budget of 2 MB -- a handful of sequences per run -- every
    route gives the labels the default (one run) gives: rd_decode_batch, the blocking raw calls, the chunk pipeline over resident windows and
    the reads pipeline in both modes (global: float64 and float32 passes, LM on)."""
    from radian_amd import synthetic
    rng = np.random.default_rng(900 + W)
    table = np.random.default_rng(1).dirichlet([0.3] * 4, size=4 ** 3)
    dev.load_lm(table, 3)
    step = 512
    try:
        reads = _ragged(rng, 14, hi=5000, special=(1, CHUNK - 1, CHUNK, 3000))
        # probabilities of a few sequences for the plain decode entry point
        good = [r for r in reads if len(r) >= 64]
        windows, valid, _, _ = synthetic.reads_to_windows(good, CHUNK, step)
        win = windows[:12]
        probs = dev.forward(win)
        mats = probs.reshape(-1, 5)
        lens = np.asarray([CHUNK, 1000, 17, CHUNK, 1, 512, CHUNK, 3, 900, CHUNK, 64, 700][: win.shape[0]], dtype=np.int32)
        off = np.arange(win.shape[0], dtype=np.int64) * CHUNK
        d_win = dev.dev_alloc(windows.nbytes)
        dev.h2d(d_win, windows)

        def everything():
            out = {}
            out["batch"] = dev.decode_batch(mats, off, lens, W, with_scores=True)
            out["batch_lm"] = dev.decode_batch(mats.astype(np.float64), off, lens, W, use_lm=True, s_threshold=0.2, r_threshold=0.9)
            out["raw_global"] = dev.basecall_raw_global(reads, 4, CHUNK, step, W, True, 0.2, 0.9)
            out["raw_chunk"] = dev.basecall_raw_chunk(reads, 4, CHUNK, step, W)
            dev.pipe_flush()
            dev.pipe_config(2)
            t = [dev.pipe_submit_raw(mode, reads, 4, CHUNK, step, W, True, 0.2, 0.9) if mode == "global" else dev.pipe_submit_raw(mode, reads, 4, CHUNK, step, W)
                 for mode in ("global", "chunk", "global")]
            dev.pipe_flush()
            out["pipe"] = [x.result() for x in t]
            lab = np.zeros((windows.shape[0], CHUNK), dtype=np.uint8)
            ln = np.full(windows.shape[0], -1, dtype=np.int32)
            dev.pipe_submit(d_win, windows.shape[0], CHUNK, valid, W, lab, ln)
            dev.pipe_flush()
            out["pipe_windows"] = (lab.copy(), ln.copy())
            return out

        def same(a, b, path=""):
            if isinstance(a, (list, tuple)):
                assert type(a) is type(b) and len(a) == len(b), path
                for i, (x, y) in enumerate(zip(a, b)):
                    same(x, y, f"{path}[{i}]")
            elif a is None:
                assert b is None, path
            else:
                assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), path

        ref = everything()
        dev.set_trie_budget(2 << 20)
        try:
            got = everything()
        finally:
            dev.set_trie_budget(0)
        for key in ref:
            same(ref[key], got[key], key)
        ln = ref["pipe_windows"][1]
        assert ln.min() >= 0 and ln.sum() > 0 and any(len(x) for x in ref["raw_global"][0] if x is not None)
        dev.dev_free(d_win)
    finally:
        dev.load_lm(None, 0)
        dev.pipe_config(4)


def test_hashed_long_contexts_with_a_wide_beam_are_refused_at_the_submit(dev):
    """hashed contexts exist in the lane kernels (W <= 64) only; the combination with a wider beam used to fail when the group was
    LAUNCHED, after earlier reads had been written.  Now RD_ERR_ARG comes from the submit / the blocking call itself, and the pipeline
    stays usable."""
    from radian_amd import RadianHipError
    rng = np.random.default_rng(5)
    table = np.random.default_rng(1).dirichlet([0.3] * 4, size=4 ** 3)
    reads = _ragged(rng, 4, lo=1200, hi=3000)
    dev.load_lm_hashed(table, 3, 40)
    try:
        ok = dev.basecall_raw_global(reads, 4, CHUNK, 512, 25, True, 0.2, 0.9)
        for call in (lambda: dev.pipe_submit_raw("global", reads, 4, CHUNK, 512, 100, True, 0.2, 0.9),
                     lambda: dev.basecall_raw_global(reads, 4, CHUNK, 512, 65, True, 0.2, 0.9)):
            with pytest.raises(RadianHipError, match="hashed long-context"):
                call()
        # without the LM a wide beam is fine on the same context, and the pipeline still works at a lane-kernel width
        dev.pipe_submit_raw("global", reads, 4, CHUNK, 512, 100, False, 0.0, 0.0).result()
        got = dev.pipe_submit_raw("global", reads, 4, CHUNK, 512, 25, True, 0.2, 0.9).result()
        assert all(np.array_equal(a, b) for a, b in zip(got[0], ok[0]))
    finally:
        dev.load_lm(None, 0)
