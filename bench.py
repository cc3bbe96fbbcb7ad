#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: input signal samples/s basecalled (chunk=1024, beam=10).

Workload (BASELINE.json configs[2], SURVEY.md section 8d cfg 3): synthetic Gaussian int16 reads of 4096
samples, MAD-normalised, chunk 1024 / step 512 -> 8 windows per read; one STEP = one batch of 512 windows
(64 reads, the north_star batch) through the whole hot path: TCN forward (fp32 MFMA) -> per-window CTC
prefix beam search W=10 (chunk mode: LM unused, reference basecall.py:110-121) -> labels back on the host.
Inputs are resident in HBM when the timed region starts.  One process per GPU; ranks shard reads with no
data-path collective (weak scaling); the only RCCL traffic is the start-up broadcast of the weights.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process is only the LAUNCHER: before anything touches a GPU it
makes a private rendezvous directory, starts N fresh copies of itself with RANK / LOCAL_RANK / WORLD_SIZE set (one per
GPU), stops the job when one of them fails, and forwards rank 0's single JSON line.  A foreign launcher that sets those
variables itself (e.g. `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) is honoured too: then
the ranks find each other through a directory named after that launcher's pid.  No PyTorch in any bench process: the
128-byte RCCL id travels through the rendezvous directory, barrier and max-over-ranks through RCCL (rd_rccl_*).

The line: `value` = the median of --regions (3) timed regions of K steps each (`value_runs`: all of them); `roofline` = the dominant
kernel against the fp32-MFMA peak from live HIP events, launch-weighted, with `by_variant`; `cpu_baseline` = the oracle port on the host's
cores.  N = 1 adds secondary legs (other precisions, forward only, the literal windowed job, peaky decode-only, global + LM, heads of other
base densities, raw and files -> FASTA end to end); N > 1 adds the files -> FASTA legs through the multi-GPU route inside the ranks
(files_leg: work queue, native reader, per-rank core slice, rank files, merger process) -- the host feed that SURVEY 8e names as the
scaling limit, which the HBM-resident headline cannot see.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK, STEP, READ_LEN, BATCH_WINDOWS, BEAM = 1024, 512, 4096, 512, 10
FLOP_PER_CONV_ROW = 2 * 256 * 256 * 3  # one dilated conv, per window time step
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
F16_MFMA_PEAK_TFLOPS = 16 * 157.3      # dense f16 MFMA = 16x the fp32 MFMA rate (~2.5 PFLOP/s)


_T0 = time.perf_counter()


def note(msg):
    """progress line on stderr (stdout carries the one JSON line only)"""
    print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def effective_cores():
    """CPU cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box hands
    a job a share of the host -- os.cpu_count() reports every hardware thread of the machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse is not None:
                quota, period = parse(open(path).read())
            else:
                quota, period = open(path).read().strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
                break
        except Exception:
            continue
    return n


def cpu_baseline(windows, valid, n_reads, cores, n_reads_single):
    """The oracle (CPU restatement, kind 'port') on bounded samples of the same workload: (ii) one OpenMP thread per
    usable host core over disjoint windows and (i) a single thread, as the reference runs (basecall.py:70-72)."""
    from oracle import oracle as orc
    from radian_amd import weights
    orc.build()
    w = weights.synthetic_weights(seed=1234)

    def leg(nr, threads):
        win, val = windows[: nr * 8], valid[: nr * 8]
        t0 = time.perf_counter()
        probs = orc.tcn_forward(w, win, nthreads=threads)
        t1 = time.perf_counter()
        n, T = win.shape
        off = np.arange(n, dtype=np.int64) * T
        orc.beam_search_batch(probs.reshape(-1, 5), off, val, BEAM, nthreads=threads)
        t2 = time.perf_counter()
        return nr * READ_LEN / (t2 - t0), t1 - t0, t2 - t1, n

    v, fs, ds, n = leg(n_reads, cores)
    out = {
        "value": v, "unit": "samples/s", "cores": int(cores), "kind": "port",
        "sample": f"{n_reads} reads x {READ_LEN} samples ({n} windows) of the same workload; oracle forward "
                  f"{fs:.2f}s + beam search {ds:.2f}s, OpenMP over {cores} threads (= the cores this job may use; the host "
                  f"reports {os.cpu_count()} hardware threads)",
        "forward_s": fs, "decode_s": ds,
    }
    if n_reads_single > 0:
        v1, fs1, ds1, n1 = leg(n_reads_single, 1)
        out["single_thread"] = {"value": v1, "unit": "samples/s", "cores": 1,
                                "sample": f"{n_reads_single} reads ({n1} windows); forward {fs1:.2f}s + beam search {ds1:.2f}s on one thread "
                                          "(the reference is single-threaded, basecall.py:70-72)"}
    return out


def check_against_oracle(be, batch, labels_fn):
    """--check: batch 0 of this rank through the timed entry point, every window's labels against the oracle's beam search
    of the GPU's probabilities, and the probabilities against the oracle forward (the oracle is the checker here, never
    the thing measured; the same comparison runs in tests/test_gpu_baseline_configs.py)."""
    from oracle import oracle as orc
    from radian_amd import weights
    orc.build()
    win, valid = batch[2], batch[1]
    probs = be.forward(win)
    ref = orc.tcn_forward(weights.synthetic_weights(seed=1234), win, nthreads=effective_cores())
    err = float(np.abs(probs - ref).max())
    assert err <= 1e-4, f"forward differs from the oracle by {err}"
    lab, ln = labels_fn()
    off = np.arange(win.shape[0], dtype=np.int64) * CHUNK
    exp = orc.beam_search_batch(probs.reshape(-1, 5), off, valid, BEAM, nthreads=effective_cores())
    bad = [w for w in range(win.shape[0]) if ln[w] != len(exp[w]) or not np.array_equal(lab[w, : ln[w]], exp[w])]
    assert not bad, f"{len(bad)} windows differ from the oracle, first {bad[:5]}"
    print(f"[bench --check] {win.shape[0]} windows: labels identical to the oracle, max |d softmax| = {err:.2e}", file=sys.stderr)


class _MemRead:
    """a read as the fast5 reader hands it to the driver loop (read_id + get_raw_data)"""

    def __init__(self, rid, sig):
        self.read_id, self._sig = rid, sig

    def get_raw_data(self):
        return self._sig


def soft_head_weights(scale=0.05, blank_bias=0.0):
    """the bench model with its last Dense kernel x 0.05 (the tests' soft head): softmax rows of ~0.85 nat entropy instead of
    saturated ones, ~200-base fragments per window / ~1000 bases per read -- the beam search's and the stitch's real load.
    blank_bias is added to the blank class's bias (dense_1/bias[4]): x 0.05 with + 4.5 gives soft rows that the blank dominates,
    ~25 bases per 1024-row window -- the base density of direct-RNA signal (tools/head_scan.py)"""
    from radian_amd import weights
    w = weights.synthetic_weights(seed=1234).copy()
    w[-645:-5] *= np.float32(scale)
    w[-1] += np.float32(blank_bias)
    return w


def ragged_lengths(n, seed, lo=1500, hi=60000, median=9000.0, sigma=0.8):
    """seeded log-normal read lengths (dRNA reads are heavy-tailed): every device batch gets its own plan"""
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(median), sigma, size=n)), lo, hi).astype(np.int64)


def total_note(lengths, dt):
    return f"{int(np.sum(lengths)) / dt / 1e6:.2f} M samples/s ({dt:.2f} s)"


def driver_leg(device, pool, cli, lengths, seed, weights_flat, lm=None, warm_reads=1536, desc="", files=None):
    """A job through the product's driver loop (radian_amd.basecall.run = basecall.py:69-141) with the CLI flags `cli`:
    host int16 reads -> H2D -> MAD normalisation on the device -> streamed forward -> (assembly) -> beam search -> labels to
    the host -> strings (chunk mode: simple_assembly natively on host threads), results in input order; everything but fast5
    parsing and FASTA writing.  Device contexts / pipelining as the CLI defaults choose them."""
    import contextlib
    from radian_amd import Backend, basecall
    args = basecall.build_parser().parse_args(["-", "-"] + cli)
    bes = [Backend(device) for _ in range(basecall.n_contexts(args))]
    try:
        bes[0].load_weights(weights_flat)
        if lm is not None:
            bes[0].load_lm(lm[0], lm[1])
        for b in bes[1:]:
            b.clone_artifacts_from(bes[0])
        for b in bes:
            b.set_decode_partition(args.decode_partition)
            b.set_precision(args.precision)
            b.set_logits(args.logits)
            b.set_decode_math(args.decode_math)
        args._lm_loaded = lm is not None
        rng = np.random.default_rng(seed)

        def go(lens):
            reads = [_MemRead(f"{i:08d}", np.rint(rng.normal(500.0, 80.0, size=int(n))).astype(np.int16)) for i, n in enumerate(lens)]
            with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
                t0 = time.perf_counter()
                res = basecall.run(args, bes, reads=iter(reads), writer=None, stitch_pool=pool if args.decode_type == "chunk" else None)
                dt = time.perf_counter() - t0
            assert len(res) == len(lens) and all(len(r[2]) > 0 for r in res)
            return dt, float(np.mean([len(r[2]) for r in res]))
        def go_files(lens, filters):
            """the whole job as the command line runs it: a directory of multi-read fast5 files (4000 reads each, as MinKNOW writes them)
            -> reads-<n>.fasta files on disk.  The files are written before the clock starts (memory-backed directory where there is one)."""
            import shutil
            import tempfile
            from radian_amd import fast5
            root = tempfile.mkdtemp(prefix="radian_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
            try:
                os.mkdir(root + "/in")
                os.mkdir(root + "/out")
                for lo in range(0, len(lens), 4000):
                    fast5.write_multi_fast5(f"{root}/in/batch_{lo // 4000}.fast5",
                                            {f"{i:08d}-0000-4000-8000-{i:012d}": np.rint(rng.normal(500.0, 80.0, size=int(lens[i]))).astype(np.int16)
                                             for i in range(lo, min(lo + 4000, len(lens)))}, filters=filters)
                size = sum(os.path.getsize(f"{root}/in/{f}") for f in os.listdir(root + "/in"))
                args.fast5_dir, args.fasta_dir = root + "/in", root + "/out"
                with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
                    t0 = time.perf_counter()
                    writer = basecall.FastaWriter(args.fasta_dir)
                    try:
                        basecall.run(args, bes, writer=writer, stitch_pool=pool if args.decode_type == "chunk" else None)
                    finally:
                        writer.close()
                    dt = time.perf_counter() - t0
                n_rec = n_base = 0
                for f in os.listdir(root + "/out"):
                    for line in open(f"{root}/out/{f}"):
                        if line.startswith(">"):
                            n_rec += 1
                        else:
                            n_base += len(line) - 1
                assert n_rec == len(lens), (n_rec, len(lens))
                return dt, n_base / max(1, n_rec), size
            finally:
                shutil.rmtree(root, ignore_errors=True)

        go(lengths[:warm_reads])                  # warm-up with full-size device batches on every forward lane: allocations happen here
        file_bytes = None
        if files is None:
            dt, mean_bases = go(lengths)
        else:
            dt, mean_bases, file_bytes = go_files(lengths, files)
        note(f"  {total_note(lengths, dt)}")
    finally:
        for b in bes:
            b.close()
    total = int(np.sum(lengths))
    if files is not None:
        return {"value": total / dt, "unit": "samples/s", "reads": int(len(lengths)), "samples": total, "seconds": dt,
                "mean_bases_per_read": mean_bases, "cli": " ".join(cli), "fast5_bytes": file_bytes,
                "signal_storage": "raw int16 chunks" if not files else "chunks through " + " + ".join(f if isinstance(f, str) else f"{f[0]}({f[1]})" for f in files),
                "path": desc + "; multi-read fast5 files (4000 reads each) -> native reader (csrc/fast5.hip) -> H2D -> on-device mad_normalise -> streamed "
                               "forward -> beam search -> labels D2H -> strings -> reads-<n>.fasta on disk, through radian_amd.basecall.run as "
                               "basecall.main drives it: NOTHING excluded but start-up (weights, contexts)"}
    return {"value": total / dt, "unit": "samples/s", "reads": int(len(lengths)), "samples": total, "seconds": dt,
            "mean_bases_per_read": mean_bases, "cli": " ".join(cli),
            "device_contexts": len(bes), "pipelined": not args.no_pipeline,
            "stitch": ("native rd_stitch_chunk (simple_assembly + difflib restated in C++) on host threads" if args.decode_type == "chunk" and not args.no_pipeline
                       else "none (global mode)" if args.decode_type != "chunk" else "pure-Python difflib in worker processes"),
            "path": desc + "; host int16 -> H2D -> on-device mad_normalise -> streamed forward -> beam search -> labels D2H -> strings, "
                           "through radian_amd.basecall.run; excludes fast5 parsing and FASTA writing"}


def global_leg(device, batches_host, read_off, reads_per_batch, steps, W, table, table_order, context_len, hashed, logits, desc,
               weights_flat=None, warmup=4):
    """Global decode (one beam search per read over the assembled float64 matrix) of the bench batches on one GPU, inputs
    resident in HBM, labels on the host at stop.  `value`: ONE context, rd_pipe_submit_reads_global (forwards on two lanes,
    assembly on the lane, the beam search of a group of steps on the decode stream under the next group's forwards).
    `two_contexts_unpipelined`: round 2's way -- two contexts on two host threads taking the steps in turn."""
    import threading
    from radian_amd import Backend, weights
    bes, bufs = [], []
    try:
        for _ in range(2):
            b = Backend(device)
            if bes:
                b.clone_artifacts_from(bes[0])
            else:
                b.load_weights(weights_flat if weights_flat is not None else weights.synthetic_weights(seed=1234))
                if hashed:
                    b.load_lm_hashed(table, table_order, context_len)
                else:
                    b.load_lm(table, table_order)
            b.set_logits(logits)
            d = []
            for norm in batches_host:
                p = b.dev_alloc(norm.nbytes)
                b.h2d(p, norm)
                d.append(p)
            bes.append(b)
            bufs.append(d)
        label_off = np.ascontiguousarray(read_off[:-1])
        nb_host = len(batches_host)
        # ---- one context, pipelined
        be = bes[0]
        ring = [(np.zeros(reads_per_batch * READ_LEN + 1, dtype=np.uint8), np.full(reads_per_batch, -1, dtype=np.int32)) for _ in range(32)]

        def run_pipe(n):
            for i in range(n):
                lab, ln = ring[i % len(ring)]
                be.pipe_submit_reads_global(bufs[0][i % nb_host], read_off, reads_per_batch, CHUNK, STEP, W, True, 0.5, 0.5, lab, label_off, ln)
                if i >= 24:
                    be.pipe_progress(be.pipe_submitted() - 24)   # at most 24 steps' buffers in flight (radian_amd.basecall.run's global-mode cap)
            be.pipe_flush()
            be.sync()
        run_pipe(warmup)
        t0 = time.perf_counter()
        run_pipe(steps)
        dt_pipe = time.perf_counter() - t0
        assert all(o[1].min() > 0 for o in ring[: min(steps, len(ring))])
        mean_bases = float(np.mean(ring[0][1]))
        # ---- two contexts, unpipelined calls
        outs = [(np.zeros(reads_per_batch * READ_LEN + 1, dtype=np.uint8), np.zeros(reads_per_batch, dtype=np.int32)) for _ in range(2)]
        err = []

        def worker(k, lo, hi):
            try:
                lab, ln = outs[k]
                for i in range(lo + k, hi, 2):
                    bes[k].basecall_reads_global_resident(bufs[k][i % nb_host], read_off, reads_per_batch, CHUNK, STEP, W, True,
                                                          0.5, 0.5, lab, label_off, ln)
                bes[k].sync()
            except Exception as e:   # surfaced below
                err.append(e)

        def run(lo, hi):
            th = [threading.Thread(target=worker, args=(k, lo, hi)) for k in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if err:
                raise err[0]
        run(0, 4)
        steps2 = min(steps, 60)      # (a comparison figure: no groups to fill, 60 steps are steady state)
        t0 = time.perf_counter()
        run(0, steps2)
        dt = time.perf_counter() - t0
        assert all(o[1].min() > 0 for o in outs)
    finally:
        for b, d in zip(bes, bufs):
            for p in d:
                b.dev_free(p)
            b.close()
    return {"value": steps * reads_per_batch * READ_LEN / dt_pipe, "unit": "samples/s", "ms_per_step": dt_pipe / steps * 1e3, "steps": steps,
            "mean_bases_per_read": mean_bases,
            "two_contexts_unpipelined": steps2 * reads_per_batch * READ_LEN / dt,
            "config": desc + "; 64 reads x 4096 per step, --decode-type global, step 512; ONE device context, rd_pipe_submit_reads_global: "
                             "streamed forward on two lanes + per-read assembly (f64) on the lane, LM beam search of a group of steps (closed "
                             "as soon as its forward rows cover the 4096-step chain of a read: two steps) on the decode stream under the "
                             "next group's forwards; two_contexts_unpipelined = round 2's scheme (two contexts on two host threads)"}


def cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11' (the kernel's cpulist form: a 64-core slice is one short string in the JSON line)"""
    out, cpus = [], sorted(int(c) for c in cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def files_leg_reads(file_index, reads_per_file, read_len, seed=70003):
    """the reads of one input file of the N-rank leg: seeded by the file's index, so any process can regenerate any file"""
    rng = np.random.default_rng([seed, file_index])
    base = file_index * reads_per_file
    return {f"{base + i:08d}-0000-4000-8000-{base + i:012d}": np.rint(rng.normal(500.0, 80.0, size=int(read_len))).astype(np.int16)
            for i in range(reads_per_file)}


def files_leg(rank, world, be_src, host_budget, rdv_dir, cli, device=0, files_per_rank=4, reads_per_file=4096, read_len=READ_LEN,
              warm_reads=1536, timeout=60.0, backend_factory=None, comm_info=None, keep=None, lm=None, tag="chunk", what="BASELINE configs[2]"):
    """secondary_e2e_fast5_to_fasta at N > 1 -- the whole job as `radian_amd.basecall ... --gpus N` runs it, inside the bench's ranks:
    a memory-backed directory of multi-read fast5 shards -> per-node FileReadQueue (blocks of 256 reads, file by file) -> native reader ->
    H2D -> mad_normalise -> streamed forward -> beam search -> labels -> strings (chunk mode: native stitch on the rank's host threads) ->
    rank files -> StreamMerger in a process of its own (a child of rank 0 that never touches a GPU) -> reads-<n>.fasta, every rank on the
    core slice hostbudget gave it.  That loop is what the ranks replace (radian/basecall.py:70-76,129-138) and what SURVEY 8e names as the
    scaling limit; the headline's inputs are resident in HBM.  launch.run_rank / launch.merge_watch are the product's own halves.

    Every rank calls this after the headline's timed region.  No RCCL call inside: the ranks meet through files in rdv_dir with a
    deadline, a rank that fails says so in its "done" note, and whatever goes wrong comes back as {"skipped": reason} on rank 0 --
    the headline line is never at stake.  Returns the leg's dict on rank 0, None elsewhere."""
    import contextlib
    import shutil
    import subprocess
    import tempfile
    from radian_amd import basecall, dist, fast5, hostbudget, launch
    t_enter = time.time()
    rdv = dist.Rendezvous(os.path.join(rdv_dir, "files_leg_" + tag), rank, world, timeout=timeout)
    root, bes, merger, result = None, [], None, None
    mine = {"rank": rank, "error": None}
    met = set()

    def meet(phase, value="ok"):
        """all ranks reach `phase`, or everyone learns at once that one of them cannot (no waiting for the deadline)"""
        met.add(phase)
        t_meet = time.time()
        vals = rdv.gather(phase, value)
        mine.setdefault("waits", {})[phase] = round(time.time() - t_meet, 3)      # (how long this rank stood at the meeting point)
        bad = [f"rank {r}: {v[7:]}" for r, v in enumerate(vals) if v.startswith("failed:")]
        if bad:
            raise RuntimeError("; ".join(bad))
        return vals
    try:
        # ---- a directory every rank can see: rank 0 makes it, the others learn its name
        if rank == 0:
            need = 3 * world * files_per_rank * reads_per_file * (2 * read_len + 512)      # the fast5 shards, with room for the rank files and the FASTA
            base = None
            try:     # memory-backed when /dev/shm has the room (a container's default /dev/shm holds 64 MB), else the temporary directory
                st = os.statvfs("/dev/shm")
                if os.access("/dev/shm", os.W_OK) and st.f_bavail * st.f_frsize > need:
                    base = "/dev/shm"
            except OSError:
                pass
            root = tempfile.mkdtemp(prefix="radian_bench_nrank_", dir=base)
            for d in ("in", "out", "scratch"):
                os.mkdir(os.path.join(root, d))
        root = meet("root", root or "-")[0]
        n_files = files_per_rank * world
        for k in range(files_per_rank):          # every rank writes its share of the input (written before any clock starts)
            fi = rank + k * world
            fast5.write_multi_fast5(os.path.join(root, "in", f"batch_{fi:04d}.fast5"), files_leg_reads(fi, reads_per_file, read_len))
        # ---- this rank's device contexts, as the command line builds them, with the artefacts of the context that received the broadcast
        args = basecall.build_parser().parse_args([os.path.join(root, "in"), os.path.join(root, "out")] + list(cli))
        n_mine = len(host_budget["cpus"]) if (host_budget.get("bound") or host_budget.get("how") != "all") else max(1, len(host_budget["cpus"]) // max(1, world))
        if args.stitch_workers is None and args.decode_type == "chunk":
            args.stitch_workers = hostbudget.threads_for(n_mine, "chunk")["stitch_threads"]
        if backend_factory is None:
            from radian_amd import Backend
            backend_factory = lambda: Backend(device)      # noqa: E731 -- (tests pass a device-less stand-in)
        for i in range(basecall.n_contexts(args)):
            b = backend_factory()
            bes.append(b)
            basecall.apply_artifacts(args, b, None, clone_from=be_src)
            if lm is not None:
                b.load_lm(lm[0], lm[1])      # (every rank builds the same seeded table: the leg's own model, outside the job's broadcast)
        args._lm_loaded = lm is not None
        if warm_reads:
            warm = [_MemRead(f"w{i}", r) for i, r in enumerate(list(files_leg_reads(10 ** 6 + rank, min(warm_reads, reads_per_file), read_len).values()))]
            with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
                basecall.run(args, bes, reads=iter(warm), writer=None, on_result=lambda *a: None)
        meet("ready")
        if rank == 0:
            with open(os.path.join(root, "scratch", "files.json"), "w") as f:
                json.dump(fast5.list_files(args.fast5_dir), f)
            merger = subprocess.Popen([sys.executable, "-m", "radian_amd.launch", "--merge", os.path.join(root, "scratch"), str(world),
                                       args.fasta_dir, str(4 * timeout)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
        meet("go")
        with open(os.path.join(root, "scratch", "files.json")) as f:
            paths = json.load(f)
        assert len(paths) == n_files, (len(paths), n_files)
        sources = [fast5.Fast5Source(p) for p in paths]
        stats = {}
        mine["t0"] = time.time()
        with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
            launch.run_rank(args, bes[0], None, os.path.join(root, "scratch"), sources, rank, world, backends=bes, stats=stats)
        mine["t1"] = time.time()
        mine.update(reads=stats.get("reads", 0), samples=stats.get("samples", 0), cores=len(host_budget["cpus"]), cpus=cpulist(host_budget["cpus"]),
                    cpu_split=host_budget.get("how"), cpu_bound=host_budget.get("bound"), numa_node=host_budget.get("numa_node"),
                    stitch_threads=args.stitch_workers, device_contexts=len(bes))
    except BaseException as e:   # noqa: BLE001 -- nothing of this leg may cost the headline
        mine["error"] = f"{type(e).__name__}: {e}"
        for phase in ("root", "ready", "go"):      # the ranks that wait for this one at a later meeting point learn it now
            if phase not in met:
                try:
                    rdv.publish(phase, "failed:" + mine["error"][:300])
                except OSError:
                    pass
        if root and root != "-" and os.path.isdir(os.path.join(root, "scratch")):
            try:   # a rank that never got to its run still has to end its rank file, or the merger waits for it
                p_rank = os.path.join(root, "scratch", f"rank{rank}.jsonl")
                if not os.path.exists(p_rank):
                    with open(p_rank, "w") as f:
                        f.write(json.dumps({"end": True}) + "\n")
            except OSError:
                pass
    finally:
        for b in bes:
            try:
                b.close()
            except Exception:
                pass
    try:
        notes = [json.loads(x) for x in rdv.gather("done", json.dumps(mine))]
    except Exception as e:
        notes = None
        mine["error"] = mine["error"] or f"{type(e).__name__}: {e}"
    try:
        rdv.publish("bye", "1")      # (this rank has read what it needs: rank 0 removes the meeting directory only when everyone has said so --
    except OSError:                  # it used to remove it right after its own "done", under the feet of a rank still polling for rank 0's note,
        pass                         # which then sat out the whole deadline)
    if rank == 0:
        try:
            merged = None
            if merger is not None:
                try:
                    mo, me = merger.communicate(timeout=2 * timeout)
                    if merger.returncode == 0:
                        merged = json.loads(mo.decode().strip().splitlines()[-1])
                    else:
                        mine["error"] = mine["error"] or ("merger: " + me.decode(errors="replace").strip().splitlines()[-1][:300])
                except subprocess.TimeoutExpired:
                    merger.kill()
                    merger.communicate()
                    mine["error"] = mine["error"] or "merger: no end within its deadline"
            errs = [f"rank {n['rank']}: {n['error']}" for n in (notes or []) if n.get("error")] + ([f"rank 0: {mine['error']}"] if mine["error"] and not notes else [])
            if errs or notes is None or merged is None:
                result = {"skipped": "; ".join(errs) or mine["error"] or "no result", "n_ranks": world}
            else:
                total_reads, total = sum(n["reads"] for n in notes), sum(n["samples"] for n in notes)
                t0, t1 = min(n["t0"] for n in notes), max(n["t1"] for n in notes)
                n_rec = n_base = 0
                import hashlib
                h = hashlib.sha256()
                for fn in sorted(os.listdir(os.path.join(root, "out")), key=lambda x: int(x.split("-")[1].split(".")[0]) if x.startswith("reads-") else -1):
                    with open(os.path.join(root, "out", fn), "rb") as f:
                        data = f.read()
                    h.update(data)
                    n_rec += data.count(b">")
                    n_base += len(data)
                assert n_rec == total_reads == merged["records"] == n_files * reads_per_file, (n_rec, total_reads, merged["records"], n_files * reads_per_file)
                result = {
                    "value": total / (t1 - t0), "unit": "samples/s", "n_ranks": world, "reads": total_reads, "samples": total, "seconds": t1 - t0,
                    "value_to_merged_fasta": total / (merged["t_done"] - t0), "seconds_to_merged_fasta": merged["t_done"] - t0,
                    "records_written": n_rec, "fasta_sha256": h.hexdigest(), "fast5_files": n_files, "reads_per_file": reads_per_file, "cli": " ".join(cli),
                    "file_order": [os.path.basename(p) for p in paths],      # the directory's own enumeration order, as the reference's rglob gives it (basecall.py:70)
                    "per_rank": [{"rank": n["rank"], "reads": n["reads"], "samples": n["samples"], "seconds": n["t1"] - n["t0"],
                                  "value": n["samples"] / max(1e-9, n["t1"] - n["t0"]), "cores": n["cores"], "cpus": n["cpus"], "cpu_split": n["cpu_split"],
                                  "cpu_bound": n["cpu_bound"], "numa_node": n["numa_node"], "stitch_threads": n["stitch_threads"],
                                  "device_contexts": n["device_contexts"], "waits_s": n.get("waits")} for n in notes],
                    "startup_comm": (comm_info or {}).get("startup_comm"), "rccl_nranks": (comm_info or {}).get("rccl_nranks"),
                    "leg_seconds_incl_input_files_and_warmup": time.time() - t_enter, "input_directory": "memory-backed (/dev/shm)" if root.startswith("/dev/shm") else "temporary directory (disk)",
                    "path": what + " from a fast5 directory to FASTA files at N ranks: " + str(n_files) + " multi-read fast5 files (" + str(reads_per_file)
                            + " reads each, written by the ranks before the clock starts, memory-backed) -> per-node FileReadQueue (blocks of 256 reads) -> native "
                              "reader (csrc/fast5.hip) on each rank's read-ahead thread -> H2D -> on-device mad_normalise -> streamed forward -> beam search -> "
                              "labels D2H -> strings (chunk mode: native stitch on the rank's host threads) -> rank files -> StreamMerger in a child process of rank 0 (no GPU; it shares "
                              "rank 0's core slice) -> reads-<n>.fasta; value = samples / (last rank's end - first rank's start), value_to_merged_fasta = until the "
                              "merged FASTA is closed and renamed into place; ranks synchronise through files, no collective inside the leg",
                }
                if keep:
                    shutil.copytree(os.path.join(root, "out"), keep)
        except BaseException as e:   # noqa: BLE001
            result = {"skipped": f"{type(e).__name__}: {e}", "n_ranks": world}
        finally:
            if merger is not None and merger.poll() is None:
                merger.kill()
            t_bye = time.time()
            while len(rdv.peek("bye")) < world and time.time() - t_bye < 15.0:
                time.sleep(0.005)
            if root and root != "-":
                shutil.rmtree(root, ignore_errors=True)
            shutil.rmtree(os.path.join(rdv_dir, "files_leg_" + tag), ignore_errors=True)
    return result


def self_launch(world, argv, worker_cmd=None):
    """--gpus N without a launcher: be the launcher.  Never creates a Backend, never loads the HIP library: the ranks are
    fresh child processes (radian_amd.launch.run_ranks: one failing rank stops the others, exit codes come back).  Rank 0's
    stdout carries the job's one JSON line and is forwarded; everything else the ranks print goes to stderr.
    worker_cmd: what to start per rank (tests pass a stand-in); default = this script with the same arguments."""
    import shutil
    import tempfile
    from radian_amd.launch import run_ranks
    scratch = tempfile.mkdtemp(prefix="radian_bench_")
    try:
        cmd = list(worker_cmd) if worker_cmd is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
        rcs, out0 = run_ranks(world, cmd, env_extra={"RD_BENCH_RDV": scratch}, capture_rank0=True)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    if any(rcs):
        sys.stderr.write(out0)
        print(f"[bench] multi-GPU run failed: rank exit codes {rcs}", file=sys.stderr)
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if len(lines) != 1:
        sys.stderr.write(out0)
        print(f"[bench] expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps steps each (after --warmup steps each); the line reports the median region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windowed", action="store_true", help="evaluate all 512 windows per step like the reference (default: streamed forward)")
    ap.add_argument("--precision", choices=["fp32", "f16x3", "bf16x3"], default="fp32",
                    help="matrix-product arithmetic of the forward: exact fp32 MFMA (default), split-f16 products, or the "
                         "three-term bf16 split (every fp32 operand reconstructed exactly, six bf16 MFMAs per product)")
    ap.add_argument("--preheat-ms", type=float, default=500.0,
                    help="untimed steps for this long before the W warm-up steps (GPU clocks after idle); 0 = none")
    ap.add_argument("--decode-math", choices=["glibc", "fast"], default="glibc",
                    help="arithmetic of the beam search in the timed region (default: the library's default, glibc's operation sequence)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (other precision modes, global+LM, raw end to end)")
    ap.add_argument("--check", action="store_true",
                    help="untimed cross-checks on batch 0: streamed labels == windowed labels == the oracle's; probabilities within 1e-4 of the oracle's")
    ap.add_argument("--decode-group", type=int, default=8, help="batches per beam-search launch in the two-stream pipeline")
    ap.add_argument("--lanes", type=int, default=2, help="forward streams the pipelined batches rotate over (1..4)")
    ap.add_argument("--conv-fuse", type=int, default=1, choices=[0, 1],
                    help="block 0's first conv inside its second conv's kernel (1, default) or as a kernel of its own (rd_set_conv_fuse; same bits): A/B runs")
    ap.add_argument("--decode-form", default="auto", choices=["auto", "one", "two", "waves", "lanes"],
                    help="launch shape of the beam search (rd_set_decode_form; no effect on results): A/B runs")
    ap.add_argument("--conv-shape", type=int, default=0, choices=[0, 1],
                    help="fp32 conv workgroup shape: 0 = 128 x 256 tiles, two workgroups per CU (product); 1 = 256 x 256, one per CU (measurement)")
    ap.add_argument("--cpu-reads", type=int, default=0, help="reads in the CPU-baseline sample (0: sized to the usable core count)")
    ap.add_argument("--nrank-files-per-rank", type=int, default=4,
                    help="N > 1: multi-read fast5 files (4096 reads x 4096 samples each) per rank in the files -> FASTA leg through the multi-GPU "
                         "route (secondary_e2e_fast5_to_fasta; ~3 s of basecalling per rank at the default); 0 = no such leg")
    ap.add_argument("--nrank-legs", default="chunk,global",
                    help="N > 1: which files -> FASTA legs run through the multi-GPU route: chunk (configs[2]), global (configs[3]'s geometry: global decode "
                         "with the 12-mer LM), or both (default)")
    ap.add_argument("--e2e-reads", type=int, default=32768,
                    help="reads of the raw end-to-end secondary leg (the other driver legs take a half or a sixteenth of it): jobs of a few "
                         "seconds each, so that the fill and drain of the beam-search groups are a few per cent of a leg, as in a real run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))   # this process is the launcher; it never touches a GPU
    args.gpus = world
    # host budget before the first GPU call (radian_amd/hostbudget.py): each rank on its own slice of the usable cores, NUMA-local to its GPU
    from radian_amd import hostbudget
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    host_budget = hostbudget.apply(local_rank, local_world,
                                   devices=[int(os.environ["RD_BENCH_DEVICE"])] * local_world if "RD_BENCH_DEVICE" in os.environ else None)

    from radian_amd import Backend, weights, synthetic
    from radian_amd.backend import RD_TIMER_CONV, RD_TIMER_DECODE, RD_TIMER_HEAD

    secondaries = world == 1 and not args.no_secondary and not args.windowed and args.precision == "fp32"
    stitch_pool = None   # (the pipelined driver stitches natively on host threads: no worker processes to start)

    from radian_amd.backend import device_for_rank
    device = int(os.environ["RD_BENCH_DEVICE"]) if "RD_BENCH_DEVICE" in os.environ else device_for_rank(local_rank)   # (override: rehearsals on a 1-GPU box)
    be = Backend(device)
    comm_kind = "single"
    comm = None
    rccl_nranks = 1
    per_rank_ms = None
    json_fd = None
    if world > 1:
        from radian_amd import dist
        # RCCL prints (version banner, warnings) on the C-level stdout; stdout must carry the one JSON line only: for the
        # whole run file descriptor 1 points at stderr, and the line goes to the saved descriptor at the end.
        os.environ.setdefault("NCCL_DEBUG", "WARN")    # a transport problem on a node this never ran on should say what it is
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        # the one collective of the job: rank 0 loads + repacks the weights, RCCL broadcasts the 8.8 MB device image
        # over xGMI; the 128-byte RCCL id goes through a file keyed by the launcher's pid (no PyTorch in this process).
        # All ranks agree on the transport before anyone uses it (dist.connect: RCCL on every rank, or the file
        # transport on every rank -- never a mix, which would leave one side inside ncclBroadcast forever).
        uid_file = dist.uid_path(directory=os.environ.get("RD_BENCH_RDV"))   # own launcher: its scratch; foreign: /tmp, by launcher pid
        try:
            comm, comm_kind = dist.connect(be, rank, world, uid_file)
            comm.bcast_artifacts(be, lambda b: b.load_weights(weights.synthetic_weights(seed=1234)))
        except dist.StartupFailed as e:
            print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)   # a helper thread is stuck inside a collective: no interpreter shutdown, the launcher stops the job
        rccl_nranks = comm.nranks_seen()   # collective on the file transport, local on RCCL (ncclCommCount)
    else:
        be.load_weights(weights.synthetic_weights(seed=1234))

    be.set_precision(args.precision)
    be.set_decode_math(args.decode_math)
    be.set_conv_shape(args.conv_shape)
    be.set_decode_form(args.decode_form)
    be.set_conv_fuse(args.conv_fuse)

    # ---- synthetic input, resident in HBM: 4 distinct batches of 64 reads per rank, cycled
    reads_per_batch = BATCH_WINDOWS // 8
    n_batches = 4
    batches = []
    for b in range(n_batches):
        reads = synthetic.synthetic_reads(reads_per_batch, READ_LEN, seed=1000 * rank + b)
        win, valid, _, _ = synthetic.reads_to_windows(reads, CHUNK, STEP)
        assert win.shape == (BATCH_WINDOWS, CHUNK)
        d = be.dev_alloc(win.nbytes)
        be.h2d(d, win)
        norm = np.stack([synthetic.mad_normalise(r, 4) for r in reads]).astype(np.float32)   # [64][4096] normalised reads
        dn = be.dev_alloc(norm.nbytes)
        be.h2d(dn, norm)
        batches.append((d, valid, win, dn))
    labels = np.zeros((BATCH_WINDOWS, CHUNK), dtype=np.uint8)
    lens = np.zeros(BATCH_WINDOWS, dtype=np.int32)
    # output buffers of the two-stream pipeline (a batch's labels land two submits later / at flush)
    be.pipe_config(args.decode_group)
    be.pipe_set_lanes(args.lanes)
    out = [(np.zeros((BATCH_WINDOWS, CHUNK), dtype=np.uint8), np.full(BATCH_WINDOWS, -1, dtype=np.int32))
           for _ in range(2 * args.decode_group)]

    read_off = np.arange(reads_per_batch + 1, dtype=np.int64) * READ_LEN

    def step(i):
        """unpipelined: forward -> decode -> labels on the host, one stream (used for the per-kernel timing pass)"""
        if args.windowed:
            be.basecall_chunk_resident(batches[i % n_batches][0], BATCH_WINDOWS, CHUNK, batches[i % n_batches][1], BEAM, labels, lens)
        else:
            be.basecall_reads_chunk_resident(batches[i % n_batches][3], read_off, reads_per_batch, CHUNK, STEP, BEAM, labels, lens)

    def submit(i):
        """pipelined, reads-level: the batch's 64 normalised reads are windowed on the device and every time step is
        computed once (bit-identical to evaluating all 512 windows; tests/test_gpu_reads.py)"""
        lab, ln = out[i % len(out)]
        be.pipe_submit_reads(batches[i % n_batches][3], read_off, reads_per_batch, CHUNK, STEP, BEAM, lab, ln)

    def submit_windowed(i):
        """pipelined, window-level: all 512 windows through the model, as the reference does"""
        d, valid = batches[i % n_batches][0], batches[i % n_batches][1]
        lab, ln = out[i % len(out)]
        be.pipe_submit(d, BATCH_WINDOWS, CHUNK, valid, BEAM, lab, ln)

    def timed(fn, preheat=True):
        nonlocal per_rank_ms
        # pre-heat: on a box whose GPU has been idle, the first ~0.2 s of work runs slower (measured: the same 20-step region 232 ms
        # right after start-up, 198 ms from the second region on; tools/submit_diag.py), which is a third of a 0.2-s timed region.
        # Untimed steps for args.preheat_ms, then the contract's W warm-up steps, then the K timed steps.
        t_pre = time.perf_counter()
        i = 0
        while preheat and (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
            fn(i)
            i += 1
            if i % args.decode_group == 0:
                be.pipe_flush()
        be.pipe_flush()
        for i in range(args.warmup):
            fn(i)
        be.pipe_flush()
        be.sync()
        if world > 1:
            comm.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn(i)
        be.pipe_flush()   # every batch's labels are on the host when the clock stops
        be.sync()
        el = time.perf_counter() - t0
        if world > 1:
            comm.barrier()
            per_rank_ms = [x / args.steps * 1e3 for x in comm.allgather(el)]   # every rank's own clock over the same K steps
            el = float(comm.allreduce_max([el])[0])
        return el

    if args.check:
        # untimed: the reads-level (streamed) path must reproduce the window-level labels exactly ...
        la, na = np.zeros_like(labels), np.zeros_like(lens)
        be.basecall_chunk_resident(batches[0][0], BATCH_WINDOWS, CHUNK, batches[0][1], BEAM, la, na)
        step(0)
        assert np.array_equal(na, lens) and all(np.array_equal(la[i, :na[i]], labels[i, :na[i]]) for i in range(BATCH_WINDOWS)), \
            "streamed != windowed labels"
        # ... and the timed entry point (pipelined submit) must reproduce the oracle's labels on the GPU's probabilities
        if args.precision == "fp32":
            def piped():
                (submit_windowed if args.windowed else submit)(0)
                be.pipe_flush()
                return out[0]
            check_against_oracle(be, batches[0], piped)
    note(f"rank {rank}: inputs resident, timing {args.regions} regions of {args.steps} steps")
    # The contract's region -- W warm-up steps, then EXACTLY K timed steps between barrier + synchronisation -- run args.regions times
    # (default 3); the line reports the region with the MEDIAN time (its own steps / ms_per_step) and every region's value in value_runs:
    # a 0.2-s region on a GPU that other work has just left carries a per cent of noise, and one sample says nothing about it.
    regions, regions_per_rank = [], []
    for k in range(max(1, args.regions)):
        regions.append(timed(submit_windowed if args.windowed else submit, preheat=(k == 0)))
        regions_per_rank.append(per_rank_ms)
    elapsed = sorted(regions)[(len(regions) - 1) // 2]
    headline_per_rank_ms = regions_per_rank[regions.index(elapsed)]      # every rank's own clock over the REPORTED region
    value_runs = [world * args.steps * reads_per_batch * READ_LEN / e for e in regions]
    note(f"rank {rank}: headline {world * args.steps * reads_per_batch * READ_LEN / elapsed / 1e6:.2f} M samples/s (regions: "
         + ", ".join(f"{v / 1e6:.2f}" for v in value_runs) + ")")
    for lab, ln in out[: max(1, min(len(out), args.steps))]:
        assert ln.min() >= 0 and ln.max() <= CHUNK and ln.sum() > 0

    samples_per_step = reads_per_batch * READ_LEN  # input samples basecalled per step per GPU
    value = world * args.steps * samples_per_step / elapsed
    # ---- roofline of the dominant kernel (dilated conv on fp32 MFMA), HIP events on the launch stream
    roof = None
    if rank == 0:
        # Conv / head launches: a pipelined region on ONE forward lane directly after the timed region (same clock / thermal
        # state): launches of consecutive batches run back to back on the lane's stream, the beam search of the previous group on
        # the decode stream -- the timed region's regime minus the second lane, whose launches would overlap these and make a
        # launch's bracketed duration count shared time.  (Until round 2's last day this pass ran unpipelined steps -- forward,
        # beam search, labels to the host, then the next step: after each host round trip the first launches run 5 % slower,
        # 0.91-0.93 ms against 0.87 in the rocprofv3 trace of the same process, which is not what the pipeline does.)
        fn_pipe = submit_windowed if args.windowed else submit
        n_prof = 8
        be.pipe_flush()
        be.pipe_set_lanes(1)
        for i in range(4):
            fn_pipe(i)
        be.pipe_flush()
        be.timer_enable(RD_TIMER_CONV, 11 * n_prof)
        be.timer_enable(RD_TIMER_HEAD, n_prof)
        for i in range(n_prof):
            fn_pipe(i)
        be.pipe_flush()
        be.sync()
        tc = be.timer_read(RD_TIMER_CONV)
        th = be.timer_read(RD_TIMER_HEAD)
        l_ms, l_fl, l_tag = be.timer_read_launches(RD_TIMER_CONV)
        h_ms, h_fl, _ = be.timer_read_launches(RD_TIMER_HEAD)
        be.timer_enable(RD_TIMER_CONV, 0)
        be.timer_enable(RD_TIMER_HEAD, 0)
        be.pipe_set_lanes(args.lanes)
        # Beam search: unpipelined single-batch steps (512 sequences alone on the chip: the latency-bound figure)
        n_dec = 4
        step(0)
        be.timer_enable(RD_TIMER_DECODE, n_dec)
        for i in range(n_dec):
            step(i)
        be.sync()
        td = be.timer_read(RD_TIMER_DECODE)
        be.timer_enable(RD_TIMER_DECODE, 0)
        # the same single-stream steps with the beam search's faster arithmetic (rd_set_decode_math 0; the timed region and the
        # figures above run the default: glibc's operation sequence, scores bit-identical to the reference's)
        other = "fast" if args.decode_math == "glibc" else "glibc"
        be.set_decode_math(other)
        step(0)
        be.timer_enable(RD_TIMER_DECODE, n_dec)
        for i in range(n_dec):
            step(i)
        be.sync()
        tdg = be.timer_read(RD_TIMER_DECODE)
        be.timer_enable(RD_TIMER_DECODE, 0)
        be.set_decode_math(args.decode_math)
        # algorithmic FLOPs of the timed launches as accounted by the library: 393 216 per evaluated time step; the
        # streamed forward evaluates fewer rows in the early layers (a head only holds the rows its layer changes)
        flop_per_launch = tc["flops"] / max(1, tc["launches"])
        avg_s = tc["total_ms"] / max(1, tc["launches"]) * 1e-3
        achieved = flop_per_launch / avg_s / 1e12
        peak = {"fp32": FP32_MFMA_PEAK_TFLOPS, "f16x3": F16_MFMA_PEAK_TFLOPS / 3.0, "bf16x3": F16_MFMA_PEAK_TFLOPS / 6.0}[args.precision]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tsrc = None
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("conv_hbm_bytes_per_launch")
                tsrc = f"profiles/traffic.json ({tj.get('source', 'offline rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE')}); not measured in this run"
            except Exception:
                traffic = None
        conv_flops_per_step = tc["flops"] / n_prof   # the same batches and geometry as every step of the timed region
        # the same HIP-event pass split by kernel variant (epilogue): launches, average duration, rows per launch, fraction of the peak.
        # `frac` above is the LAUNCH-WEIGHTED figure over all of them (total FLOPs / total time), not the best variant's.
        by_variant = {}
        for tag, name in ((0, "relu"), (1, "res_ident"), (2, "res_match")):
            sel = l_tag == tag
            if sel.any():
                t_ms, t_fl = float(l_ms[sel].sum()), float(l_fl[sel].sum())
                by_variant[name] = {"launches": int(sel.sum()), "avg_ms": t_ms / int(sel.sum()), "rows_per_launch": t_fl / int(sel.sum()) / FLOP_PER_CONV_ROW,
                                    "tflops": t_fl / (t_ms * 1e-3) / 1e12, "frac": t_fl / (t_ms * 1e-3) / 1e12 / peak}
        if len(h_ms):
            by_variant["head"] = {"launches": int(len(h_ms)), "avg_ms": float(h_ms.mean()), "rows_per_launch": float(h_fl.mean()) / (2.0 * (256 * 128 + 128 * 5)),
                                  "tflops": float(h_fl.sum()) / (float(h_ms.sum()) * 1e-3) / 1e12,
                                  "frac": float(h_fl.sum()) / (float(h_ms.sum()) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  "note": "Dense(128) on fp32 MFMA + ReLU + Dense(5) + softmax on the vector unit; FLOPs of both, priced against the fp32 MFMA peak"}
        roof = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic, "traffic_source": tsrc,
            "kernel": {"fp32": "tcn_gemm_kernel<4,3,*> (dilated causal conv 256->256, k=3, v_mfma_f32_32x32x2_f32)",
                       "f16x3": "tcn_gemm_split_kernel<4,3,*> (same conv, 3 x v_mfma_f32_32x32x16_f16 per fp32 product; peak = 2516/3 TFLOP/s)",
                       "bf16x3": "tcn_gemm_bf3_kernel<4,3,*> (same conv, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product; peak = 2516/6 TFLOP/s)"}[args.precision],
            "flop_per_launch": flop_per_launch, "avg_launch_ms": avg_s * 1e3, "launches_timed": tc["launches"],
            "frac_is": "launch-weighted over every conv launch of the pass: total algorithmic FLOPs / total bracketed time / peak (by_variant splits it)",
            "by_variant": by_variant,
            # the fraction that refers to the TIMED REGION: all conv FLOPs the region issued / its wall time / the peak
            # (includes the head, the C_in=1 layer, beam search tails and launch gaps as lost time)
            "pipeline_frac": conv_flops_per_step * args.steps / elapsed / 1e12 / peak,
            "pipeline_conv_tflops": conv_flops_per_step * args.steps / elapsed / 1e12,
            "conv_ms_per_step": tc["total_ms"] / n_prof, "decode_ms_per_step": td["total_ms"] / n_dec,
            "head_ms_per_step": th["total_ms"] / n_prof,
            "timing": "frac/achieved: HIP events around every conv launch of a pipelined region on ONE forward lane directly after the "
                      "timed region (launches back to back on the lane's stream, the previous group's beam search on the decode stream: "
                      "the kernel's own duration; = profiles/*_kernel_stats.csv, taken with --lanes 1). In the timed region the launches "
                      "of consecutive batches overlap on two lanes, so a launch's bracketed duration there also counts the time it "
                      "shares the chip (profiles/*_kernel_stats_default_2lanes.csv); pipeline_frac is the timed region's own figure. "
                      "decode_*: unpipelined single-batch steps (512 sequences alone on the chip)",
            "decode_timesteps_per_s": float(sum(batches[i % n_batches][1].sum() for i in range(n_dec))) / max(1e-9, td["total_ms"] * 1e-3),
            # SURVEY 8d asks for both figures of the beam search: time steps/s (above) and the HBM fraction its 20 B per
            # time step amount to -- the kernel is issue / latency bound (a T-long serial chain per sequence), not HBM bound:
            # secondary_decode_only_peaky carries the yardstick that fits (instruction-issue share)
            "decode_hbm_frac": float(sum(batches[i % n_batches][1].sum() for i in range(n_dec))) / max(1e-9, td["total_ms"] * 1e-3) * 20.0 / 8.0e12,
            "decode_timesteps_per_s_" + other + "_math": float(sum(batches[i % n_batches][1].sum() for i in range(n_dec))) / max(1e-9, tdg["total_ms"] * 1e-3),
        }
    # secondaries, reported beside the headline (one GPU, fp32 headline only): the same job in the other matrix-product
    # modes; configs[3]'s global + LM geometry; the raw-reads end-to-end driver loop
    sec = {}
    if world == 1 and args.precision == "fp32" and not args.no_secondary and not args.windowed:
        for mode, key, desc, acc in (
                ("bf16x3", "secondary_bf16x3",
                 "three-term bf16 split: hi+mid+lo reconstructs every finite fp32 operand exactly; six v_mfma_f32_32x32x16_bf16 per "
                 "product (terms below 2^-24 relative dropped), fp32 accumulate",
                 "tests/test_gpu_forward.py::test_forward_bf16x3_*"),
                ("f16x3", "secondary_f16x3",
                 "f16x3 split products (3 x v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate; 22-bit operands)",
                 "max |softmax - float64-accumulated reference|: 6.6e-6 / 6.5e-5 (peaky head) vs 1.1e-5 / 7.1e-5 for the fp32-MFMA mode "
                 "(tests/test_gpu_forward.py)")):
            note(key)
            try:
                be.set_precision(mode)
            except Exception:
                continue
            try:
                el2 = timed(submit)
            finally:
                be.set_precision("fp32")
            sec[key] = {"precision": desc, "value": world * args.steps * samples_per_step / el2, "unit": "samples/s",
                        "ms_per_step": el2 / args.steps * 1e3, "accuracy": acc}
    if world == 1 and args.precision == "fp32" and not args.no_secondary and not args.windowed:
        # BASELINE configs[1] "forward only": the headline's batches through the signal model alone (rd_forward_reads_resident: the
        # streamed evaluation, no beam search), on the same rotating forward lanes.  Beside the headline it splits the step into its
        # forward and what the beam search adds (co-running hides none of the search's SIMD time).
        note("secondary_forward_only")
        try:
            k = [0]

            def fwd(i):
                be.forward_reads_resident(batches[i % n_batches][3], read_off, reads_per_batch, CHUNK, STEP, "chunk", lane=k[0] % args.lanes)
                k[0] += 1
            el_f = timed(fwd)
            sec["secondary_forward_only"] = {
                "value": world * args.steps * samples_per_step / el_f, "unit": "samples/s", "ms_per_step": el_f / args.steps * 1e3,
                "config": "BASELINE configs[1]: the signal model's forward alone on the headline's batches (64 reads x 4096, chunk 1024 / "
                          f"step 512, every time step once + the window heads), {args.lanes} forward lanes, no beam search",
                "beam_search_adds_ms_per_step": elapsed / args.steps * 1e3 - el_f / args.steps * 1e3}
        except Exception as e:
            print(f"[bench] secondary_forward_only failed: {e}", file=sys.stderr)
    if world == 1 and args.precision == "fp32" and not args.no_secondary and not args.windowed:
        # The literal configs[2] job: ALL 512 windows x 1024 rows through the model, as the reference evaluates them (basecall.py:88-93),
        # rd_pipe_submit on the same lanes and groups.  Bit-identical labels to the headline (bench.py --check); 1.78x the model rows.
        note("secondary_windowed")
        try:
            el_w = timed(submit_windowed)
            flop_row = 11 * FLOP_PER_CONV_ROW + 2 * 3 * 256 + 2 * 256 + 2 * (256 * 128 + 128 * 5)     # 4 394 240: every layer of the model
            bound = FP32_MFMA_PEAK_TFLOPS * 1e12 / (flop_row * CHUNK / STEP)                           # 17.9 M input samples/s at step 512
            sec["secondary_windowed"] = {
                "value": world * args.steps * samples_per_step / el_w, "unit": "samples/s", "ms_per_step": el_w / args.steps * 1e3,
                "model_rows_per_step": BATCH_WINDOWS * CHUNK, "flop_per_row": flop_row, "tflops": BATCH_WINDOWS * CHUNK * flop_row * args.steps / el_w / 1e12,
                "bound_samples_per_s": bound, "frac": world * args.steps * samples_per_step / el_w / bound,
                "config": "BASELINE configs[2] as the reference computes it: every window's 1024 rows through all layers (512 x 1024 = 524 288 model rows per "
                          "step against the headline's 375 040), forward fp32 MFMA + beam search W=10 + labels to the host, rd_pipe_submit, "
                          f"{args.lanes} lanes, groups of {args.decode_group}; frac = whole-step model FLOPs (4 394 240 per row, SURVEY 8d) / time / 157.3 TFLOP/s = "
                          "value / 17.9 M samples/s: the same FLOP basis as SURVEY's roofline, beam search and head included as lost time"}
        except Exception as e:
            print(f"[bench] secondary_windowed failed: {e}", file=sys.stderr)
        # SURVEY 8d's decode-only benchmark: peaky rows (logits x 4, blank + 2), no model in the loop -- the beam search alone at full
        # occupancy (4096 windows x 1024 rows resident in HBM), W = 6 / 10 / 25, the library's default arithmetic.  The kernel is
        # issue / latency bound (a T-long serial chain per sequence): time steps/s and the share of the chip's instruction-issue slots
        # its instructions take are the yardsticks, not HBM bytes.
        note("secondary_decode_only_peaky")
        try:
            n_dw = 4096
            pk = synthetic.peaky_probs(n_dw, CHUNK, seed=7)
            d_pk = be.dev_alloc(pk.nbytes)
            be.h2d(d_pk, pk)
            del pk
            v_pk = np.full(n_dw, CHUNK, dtype=np.int32)
            lab_pk, len_pk = np.zeros((n_dw, CHUNK), dtype=np.uint8), np.zeros(n_dw, dtype=np.int32)
            instr = {}
            ipath = os.path.join(ROOT, "profiles", "decode_peaky_instr.json")
            if os.path.exists(ipath):
                try:
                    instr = json.load(open(ipath))
                except Exception:
                    instr = {}
            leg = {"rows": "softmax(4 * N(0,1) + 2 on the blank), float32, 4096 windows x 1024 rows resident in HBM (synthetic.peaky_probs seed 7)",
                   "arithmetic": args.decode_math, "by_width": {},
                   "valu_issue_frac_is": "vector-ALU issue cycles / SIMD cycles of the chip: VALU instructions per time step (offline rocprofv3 --pmc pass of this leg, "
                                         "profiles/decode_peaky_instr.json) x 4 cycles (a wave64 instruction on a 16-lane SIMD) x time steps/s / (1024 SIMDs x 2.4 GHz); "
                                         "the rest of a step is LDS round trips, scalar work and waits inside one wave's serial chain; null when that file is absent"}
            for Wd in (6, 10, 25):
                be.decode_resident(d_pk, n_dw, CHUNK, v_pk, Wd, lab_pk, len_pk)
                n_l = 3
                be.timer_enable(RD_TIMER_DECODE, n_l)
                for _ in range(n_l):
                    be.decode_resident(d_pk, n_dw, CHUNK, v_pk, Wd, lab_pk, len_pk)
                be.sync()
                tdp = be.timer_read(RD_TIMER_DECODE)
                be.timer_enable(RD_TIMER_DECODE, 0)
                sps = n_dw * CHUNK * tdp["launches"] / (tdp["total_ms"] * 1e-3)
                per = {k: (instr.get(k + "_per_step") or {}).get(str(Wd)) for k in ("valu", "salu", "lds")}
                leg["by_width"][str(Wd)] = {"timesteps_per_s": sps, "ms_per_launch": tdp["total_ms"] / max(1, tdp["launches"]),
                                            "mean_bases_per_window": float(len_pk.mean()), "instructions_per_step": per if per["valu"] else None,
                                            "valu_issue_frac": (sps * per["valu"] * 4.0 / (1024 * 2.4e9)) if per["valu"] else None}
            be.dev_free(d_pk)
            sec["secondary_decode_only_peaky"] = leg
        except Exception as e:
            print(f"[bench] secondary_decode_only_peaky failed: {e}", file=sys.stderr)
    if secondaries:
        norms = [np.stack([synthetic.mad_normalise(r, 4) for r in synthetic.synthetic_reads(reads_per_batch, READ_LEN, seed=1000 * rank + b)]).astype(np.float32)
                 for b in range(n_batches)]
        table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11)
        soft = soft_head_weights()
        for key, kw in (("secondary_global_lm", dict(W=BEAM, table_order=11, context_len=11, hashed=False, logits="f32",
                                                     desc="BASELINE configs[3] geometry, one GPU: beam 10, k=11 LM table (4^11 x 4 f64, Dirichlet(0.3) seed 0), "
                                                          "sig/rna thresholds 0.5/0.5")),
                        ("secondary_global_lm_soft_head", dict(W=BEAM, table_order=11, context_len=11, hashed=False, logits="f32", weights_flat=soft,
                                                               desc="the same with the soft head (last Dense kernel x 0.05: ~1000 bases per read, the LM "
                                                                    "gate fires on most steps)")),
                        ("secondary_cfg5_w25_ctx256_f16", dict(W=25, table_order=11, context_len=256, hashed=True, logits="f16",
                                                               desc="BASELINE configs[4] stress: beam 25, --context-len 256 (hashed synthetic LM over a 4^11-row "
                                                                    "table, 256-label ring per beam), f16 logits; no reference behaviour (SURVEY F7): parity vs the "
                                                                    "oracle's same definition, tests/test_gpu_cfg5.py"))):
            note(key)
            try:
                # (3 x the headline's steps: the first and the last group's searches run alone, a tenth of a 20-step region)
                sec[key] = global_leg(device, norms, read_off, reads_per_batch, max(200, 3 * args.steps), table=table, **kw)   # ~2 s each: in 60 steps the groups' fill and drain are 1.5 %
            except Exception as e:
                print(f"[bench] {key} failed: {e}", file=sys.stderr)
        # the headline step on the soft-head model (same timed region as `value`)
        note("secondary_soft_head")
        try:
            be.load_weights(soft)
            el_soft = timed(submit)
            sec["secondary_soft_head"] = {
                "value": world * args.steps * samples_per_step / el_soft, "unit": "samples/s", "ms_per_step": el_soft / args.steps * 1e3,
                "mean_bases_per_window": float(np.mean([ln.mean() for _, ln in out[: max(1, min(len(out), args.steps))]])),
                "config": "the headline's step and timed region with the soft head (last Dense kernel x 0.05): the headline's He-normal "
                          "rows are saturated (~5 bases per 1024-row window); here windows decode to ~200 bases with merges and re-entries "
                          "every step"}
        except Exception as e:
            print(f"[bench] secondary_soft_head failed: {e}", file=sys.stderr)
        finally:
            be.load_weights(weights.synthetic_weights(seed=1234))
        # ... and on a head between the two brackets: soft rows dominated by the blank, ~25 bases per window, the base density of dRNA signal
        note("secondary_drna_like_head")
        try:
            be.load_weights(soft_head_weights(0.05, 4.5))
            el_d = timed(submit)
            sec["secondary_drna_like_head"] = {
                "value": world * args.steps * samples_per_step / el_d, "unit": "samples/s", "ms_per_step": el_d / args.steps * 1e3,
                "mean_bases_per_window": float(np.mean([ln.mean() for _, ln in out[: max(1, min(len(out), args.steps))]])),
                "config": "the headline's step and timed region with the last Dense kernel x 0.05 and + 4.5 on the blank's bias: soft rows that the blank "
                          "dominates, a base every ~40 rows -- direct-RNA signal carries ~25 bases per 1024 samples; the headline's rows are saturated (~5 per "
                          "window), secondary_soft_head's decode to ~200 (VERDICT r5 weak 6: neither bracket is dRNA-like)"}
        except Exception as e:
            print(f"[bench] secondary_drna_like_head failed: {e}", file=sys.stderr)
        finally:
            be.load_weights(weights.synthetic_weights(seed=1234))
        if args.e2e_reads > 0:
            w0 = weights.synthetic_weights(seed=1234)
            chunk_cli = ["--decode-type", "chunk", "--step-size", str(STEP), "--chunk-len", str(CHUNK), "--beam-width", str(BEAM), "--rna-model", "None"]
            legs = (
                ("secondary_e2e_raw", chunk_cli, np.full(args.e2e_reads, READ_LEN, dtype=np.int64), w0, None,
                 f"BASELINE configs[2] end to end: {args.e2e_reads} uniform reads x 4096 (every batch has the same plan)"),
                ("secondary_e2e_raw_ragged", chunk_cli, ragged_lengths(args.e2e_reads // 2, 71), w0, None,
                 "the same job on RAGGED reads (seeded log-normal lengths, median 9000, 1.5 k .. 60 k samples): every device batch "
                 "builds and uploads its own plan"),
                ("secondary_e2e_raw_soft_head", chunk_cli, np.full(args.e2e_reads // 2, READ_LEN, dtype=np.int64), soft, None,
                 "configs[2] end to end with the soft head: ~200-base fragments per window through the stitch (difflib's placement rule)"),
                ("secondary_reference_defaults", ["--rna-threshold", "0.5"], ragged_lengths(args.e2e_reads // 2, 72), soft, (table, 11),
                 "the reference's own defaults (basecall.py:24-35): --decode-type global, step 128, beam 6, 12-mer LM (4^11-row table), "
                 "thresholds 0.5 / 0.5, on ragged reads with the soft head (a job of a few seconds: on a quarter of the reads the fill and "
                 "drain of the beam-search groups were a third of the run and the figure 23 M)"),
                ("secondary_long_reads", ["--rna-threshold", "0.5"], np.full(max(256, args.e2e_reads // 16), 100000, dtype=np.int64), soft, (table, 11),
                 "reference defaults on LONG reads (100 000 samples each: one read's beam search is a 0.2-s serial chain)"),
            )
            for key, cli, lens, wf, lm, desc in legs:
                note(key)
                try:
                    sec[key] = driver_leg(device, stitch_pool, cli, lens, 70002, wf, lm=lm, desc=desc)
                except Exception as e:
                    print(f"[bench] {key} failed: {e}", file=sys.stderr)
            # the same job from files to files
            for key, filters in (("secondary_e2e_fast5_to_fasta", ()), ("secondary_e2e_gzip_fast5_to_fasta", (("deflate", 1),))):
                note(key)
                try:
                    sec[key] = driver_leg(device, stitch_pool, chunk_cli, np.full(args.e2e_reads // 2, READ_LEN, dtype=np.int64), 70003, w0, desc=(
                        "BASELINE configs[2] from a fast5 directory to FASTA files"), files=filters)
                except Exception as e:
                    print(f"[bench] {key} failed: {e}", file=sys.stderr)
            # ... and at configs[3]'s geometry (global decode, beam 10, the 12-mer LM): the N = 1 point of the N-rank leg of the same name
            note("secondary_e2e_fast5_to_fasta_global_lm")
            try:
                sec["secondary_e2e_fast5_to_fasta_global_lm"] = driver_leg(
                    device, stitch_pool, ["--decode-type", "global", "--step-size", str(STEP), "--chunk-len", str(CHUNK), "--beam-width", str(BEAM), "--rna-model", "None"],
                    np.full(args.e2e_reads // 2, READ_LEN, dtype=np.int64), 70003, w0, lm=(table, 11),
                    desc="BASELINE configs[3]'s geometry (global decode, beam 10, 12-mer LM 0.5 / 0.5) from a fast5 directory to FASTA files", files=())
            except Exception as e:
                print(f"[bench] secondary_e2e_fast5_to_fasta_global_lm failed: {e}", file=sys.stderr)
        del table
    nrank_legs = [x for x in args.nrank_legs.split(",") if x in ("chunk", "global")]
    if world > 1 and args.precision == "fp32" and not args.no_secondary and not args.windowed and args.nrank_files_per_rank > 0 and nrank_legs:
        # The one leg of the N > 1 line: the job from a fast5 directory to FASTA files through the product's multi-GPU
        # route, every rank on its own core slice -- the host feed SURVEY 8e names as the scaling limit.  Every rank takes part; the
        # ranks meet through files (own launcher: its rendezvous directory; foreign launcher: a directory named after its pid).
        note(f"rank {rank}: secondary_e2e_fast5_to_fasta at {world} ranks")
        try:
            from radian_amd import dist as _dist
            leg_dir = os.environ.get("RD_BENCH_RDV") or (_dist.uid_path() + "_legs")
            os.makedirs(leg_dir, exist_ok=True)
            leg = None
            if "chunk" in nrank_legs:
                leg = files_leg(rank, world, be, host_budget, leg_dir,
                                ["--decode-type", "chunk", "--step-size", str(STEP), "--chunk-len", str(CHUNK), "--beam-width", str(BEAM), "--rna-model", "None"],
                                device=device, files_per_rank=args.nrank_files_per_rank, comm_info={"startup_comm": comm_kind, "rccl_nranks": rccl_nranks})
            if rank == 0 and "chunk" in nrank_legs:
                sec["secondary_e2e_fast5_to_fasta"] = leg
                note(f"  {leg['value'] / 1e6:.2f} M samples/s over {world} ranks ({leg['seconds']:.2f} s; merged FASTA after {leg['seconds_to_merged_fasta']:.2f} s)"
                     if leg and "value" in leg else f"  skipped: {leg}")
            # ... and BASELINE configs[3]'s geometry, the one its scaling curve is quoted on: --decode-type global (assembly + one LM-gated beam search per
            # read), beam 10, the 12-mer RNA model (4^11-row table, Dirichlet(0.3) seed 0, built by every rank), thresholds 0.5 / 0.5 -- same files route
            if "global" in nrank_legs:
                note(f"rank {rank}: secondary_e2e_fast5_to_fasta_global_lm at {world} ranks")
                table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11)
                leg_g = files_leg(rank, world, be, host_budget, leg_dir,
                                  ["--decode-type", "global", "--step-size", str(STEP), "--chunk-len", str(CHUNK), "--beam-width", str(BEAM), "--rna-model", "None"],
                                  device=device, files_per_rank=args.nrank_files_per_rank, comm_info={"startup_comm": comm_kind, "rccl_nranks": rccl_nranks},
                                  lm=(table, 11), tag="global", what="BASELINE configs[3]'s geometry (global decode, beam 10, 12-mer LM 0.5 / 0.5)")
                del table
                if rank == 0:
                    sec["secondary_e2e_fast5_to_fasta_global_lm"] = leg_g
                    note(f"  {leg_g['value'] / 1e6:.2f} M samples/s over {world} ranks ({leg_g['seconds']:.2f} s)" if leg_g and "value" in leg_g else f"  skipped: {leg_g}")
            if rank == 0 and "RD_BENCH_RDV" not in os.environ:
                try:
                    os.rmdir(leg_dir)
                except OSError:
                    pass
        except BaseException as e:   # noqa: BLE001 -- never at the headline's cost
            if rank == 0:
                for name, key in (("chunk", "secondary_e2e_fast5_to_fasta"), ("global", "secondary_e2e_fast5_to_fasta_global_lm")):
                    if name in nrank_legs and key not in sec:
                        sec[key] = {"skipped": f"{type(e).__name__}: {e}", "n_ranks": world}
    if stitch_pool is not None:
        stitch_pool.shutdown()
    halo = 252
    # probability rows produced per step: every window row (windowed) or every time step once + 7 window heads (streamed)
    rows_streamed = BATCH_WINDOWS * CHUNK if args.windowed else reads_per_batch * (READ_LEN + 7 * halo)

    cpu = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cores = effective_cores()
            nr = args.cpu_reads or max(2, min(reads_per_batch, 4 * cores))   # ~10-20 s of CPU work, all usable cores busy
            note(f"cpu_baseline: oracle on {nr} reads, {cores} threads")
            try:
                cpu = cpu_baseline(batches[0][2], batches[0][1], nr, cores, max(1, min(8, nr)))
            except Exception as e:   # the oracle is test infrastructure: its absence must not cost the GPU line
                print(f"[bench] cpu_baseline unavailable: {e}", file=sys.stderr)
                cpu = None

    if rank == 0:
        out = {
            "metric": "signal samples/s basecalled (chunk=1024, beam=10)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "value_runs": value_runs, "value_is": f"the median of {len(value_runs)} timed regions of {args.steps} steps each (value_runs: every region, in order)",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16x3": "f16x3 split (f32 accumulate)", "bf16x3": "bf16x3 split of f32 operands (f32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: synthetic Gaussian int16 reads x 4096 samples (round(N(500,80))), "
                            "MAD-normalised, chunk=1024 step=512 -> 8 windows/read; step = 512 windows (64 reads): "
                            "TCN forward fp32 + chunk-mode CTC beam search W=10 over every window (LM unused in chunk "
                            "mode, reference basecall.py:110-121) + labels to host; random He-normal weights seed 1234 -- whose "
                            "softmax rows are SATURATED (~5 bases per 1024-row window: the beam search's easy case); "
                            "secondary_soft_head / secondary_e2e_raw_soft_head run the same job on soft rows (~200 bases per window)",
                "preheat_ms": args.preheat_ms,
                "timed_region": "starts with MAD-normalised float32 reads resident in HBM; ends with every window's labels on the host. "
                                "Excludes H2D of the raw signal, mad_normalise and the host string stitch -- those are inside "
                                "secondary_e2e_raw",
                "forward": ("windowed: all 512 windows x 1024 rows through the model, as the reference does" if args.windowed else
                            "streamed: each read's time steps are evaluated once (4096 + 7*252 rows per read instead of "
                            "8*1024); probabilities and labels bit-identical to the windowed evaluation "
                            "(tests/test_gpu_reads.py; bench.py --check; --windowed runs the windowed job)"),
                "model_rows_per_step": rows_streamed,
                "chunk_len": CHUNK, "step_size": STEP, "batch_windows": BATCH_WINDOWS, "beam_width": BEAM,
                "decode_type": "chunk", "samples_per_step_per_gpu": samples_per_step, "sharding": "reads per rank, no data-path collective", "startup_comm": comm_kind,
                "launcher": "bench.py (own)" if "RD_BENCH_RDV" in os.environ else ("foreign (RANK/WORLD_SIZE were set)" if world > 1 else "none"),
                "pipelining": f"{args.lanes} forward streams taking the steps' batches in turn (independent kernel chains fill each other's last, partial round of workgroups) + 1 decode stream: beam search + label copy-out of a group of {args.decode_group} batches overlaps the next group's forwards; all labels on host at stop",
            },
            "roofline": roof,
            # multi-GPU evidence: the transport every rank agreed on, the communicator size as RCCL reports it
            # (ncclCommCount; on the file transport: ranks that answered an exchange), every rank's own ms per step
            "startup_comm": comm_kind, "rccl_nranks": rccl_nranks, "ms_per_step_per_rank": headline_per_rank_ms,
            "host_budget_rank0": {"cores": len(host_budget["cpus"]), "split": host_budget["how"], "bound": host_budget["bound"], "numa_node": host_budget["numa_node"]},
        }
        out.update(sec)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        if json_fd is None:
            print(json.dumps(out))
        else:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
        if "RD_BENCH_RDV" not in os.environ:
            from radian_amd import dist
            dist.leave(rank, world, dist.uid_path())
    for b in batches:
        be.dev_free(b[0])
        be.dev_free(b[3])
    be.close()


if __name__ == "__main__":
    main()
