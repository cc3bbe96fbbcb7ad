#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: input signal samples/s basecalled (chunk=1024, beam=10).

Workload (BASELINE.json configs[2], SURVEY.md section 8d cfg 3): synthetic Gaussian int16 reads of 4096
samples, MAD-normalised, chunk 1024 / step 512 -> 8 windows per read; one STEP = one batch of 512 windows
(64 reads, the north_star batch) through the whole hot path: TCN forward (fp32 MFMA) -> per-window CTC
prefix beam search W=10 (chunk mode: LM unused, reference basecall.py:110-121) -> labels back on the host.
Inputs are resident in HBM when the timed region starts.  One process per GPU; ranks shard reads with no
data-path collective (weak scaling); the only RCCL traffic is the start-up broadcast of the weights.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

No PyTorch in the measured process: rendezvous for the RCCL unique id goes through a file keyed by
MASTER_PORT, barrier and max-over-ranks through RCCL (rd_rccl_*).
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


def cpu_baseline(windows, valid, n_reads):
    """The oracle (CPU restatement, kind 'port') on a bounded sample of the same workload, all host cores."""
    from oracle import oracle as orc
    from radian_amd import weights
    orc.build()
    w = weights.synthetic_weights(seed=1234)
    cores = orc.num_threads()
    t0 = time.perf_counter()
    probs = orc.tcn_forward(w, windows, nthreads=cores)
    t1 = time.perf_counter()
    n, T = windows.shape
    off = np.arange(n, dtype=np.int64) * T
    orc.beam_search_batch(probs.reshape(-1, 5), off, valid, BEAM, nthreads=cores)
    t2 = time.perf_counter()
    samples = n_reads * READ_LEN
    return {
        "value": samples / (t2 - t0), "unit": "samples/s", "cores": int(cores), "kind": "port",
        "sample": f"{n_reads} reads x {READ_LEN} samples ({n} windows) of the same workload; oracle forward "
                  f"{t1 - t0:.2f}s + beam search {t2 - t1:.2f}s, OpenMP over {cores} threads",
        "forward_s": t1 - t0, "decode_s": t2 - t1,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windowed", action="store_true", help="evaluate all 512 windows per step like the reference (default: streamed forward)")
    ap.add_argument("--precision", choices=["fp32", "f16x3"], default="fp32",
                    help="matrix-product arithmetic of the forward: exact fp32 MFMA (default) or split-f16 products (fp32-equivalent accuracy)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary timed run in the other precision mode")
    ap.add_argument("--check", action="store_true", help="untimed cross-check: streamed labels == windowed labels on one batch")
    ap.add_argument("--decode-group", type=int, default=8, help="batches per beam-search launch in the two-stream pipeline")
    ap.add_argument("--lanes", type=int, default=2, help="forward streams the pipelined batches rotate over (1..4)")
    ap.add_argument("--cpu-reads", type=int, default=0, help="reads in the CPU-baseline sample (0: sized to the host core count)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    from radian_amd import Backend, weights, synthetic
    synthetic.mad_normalise = __import__('radian_amd.preprocess', fromlist=['mad_normalise']).mad_normalise
    from radian_amd.backend import RD_TIMER_CONV, RD_TIMER_DECODE, RD_TIMER_HEAD

    be = Backend(int(os.environ.get("RD_BENCH_DEVICE", local_rank)))   # override only for rehearsals on a 1-GPU box
    comm_kind = "single"
    comm = None
    if world > 1:
        from radian_amd import dist
        # RCCL prints a version banner on the C-level stdout; stdout must carry the one JSON line only
        sys.stdout.flush()
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            # the one collective of the job: rank 0 loads + repacks the weights, RCCL broadcasts the 8.8 MB device image
            # over xGMI; the 128-byte RCCL id goes through a file keyed by the launcher's pid (no PyTorch in this process)
            comm = dist.RcclComm(be, rank, world, dist.uid_path())
            comm.bcast_artifacts(be, lambda b: b.load_weights(weights.synthetic_weights(seed=1234)))
            comm_kind = "rccl"
        except Exception as e:  # keep the scaling run alive: file-based barrier, every rank loads its own weights
            print(f"[bench rank {rank}] RCCL start-up failed ({e}); falling back to file-based barrier", file=sys.stderr)
            comm = dist.FileComm(rank, world, dist.uid_path() + ".fc")
            be.load_weights(weights.synthetic_weights(seed=1234))
            comm_kind = "file-fallback"
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd1, 1)
            os.close(saved_fd1)
    else:
        be.load_weights(weights.synthetic_weights(seed=1234))

    be.set_precision(args.precision)

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

    def timed(fn):
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
            el = float(comm.allreduce_max([el])[0])
        return el

    if args.check:
        # untimed: the reads-level (streamed) path must reproduce the window-level labels exactly
        la, na = np.zeros_like(labels), np.zeros_like(lens)
        be.basecall_chunk_resident(batches[0][0], BATCH_WINDOWS, CHUNK, batches[0][1], BEAM, la, na)
        step(0)
        assert np.array_equal(na, lens) and all(np.array_equal(la[i, :na[i]], labels[i, :na[i]]) for i in range(BATCH_WINDOWS)), \
            "streamed != windowed labels"
    elapsed = timed(submit_windowed if args.windowed else submit)
    for lab, ln in out[: max(1, min(len(out), args.steps))]:
        assert ln.min() >= 0 and ln.max() <= CHUNK and ln.sum() > 0

    samples_per_step = reads_per_batch * READ_LEN  # input samples basecalled per step per GPU
    value = world * args.steps * samples_per_step / elapsed
    # secondary, reported beside the headline: the same job with split-f16 ("f16x3") matrix products -- every fp32 operand
    # as an f16 hi+lo pair, three f16 MFMAs per product, fp32 accumulate; same softmax error vs a float64-accumulated
    # reference as the fp32-MFMA mode (tests/test_gpu_forward.py::test_forward_split_f16x3_accuracy)
    secondary = None
    if args.precision == "fp32" and not args.no_secondary and not args.windowed:
        try:
            be.set_precision("f16x3")
            el2 = timed(submit)
        finally:
            be.set_precision("fp32")
        secondary = {"precision": "f16x3 split products (3 x v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate)",
                     "value": world * args.steps * samples_per_step / el2, "unit": "samples/s",
                     "ms_per_step": el2 / args.steps * 1e3,
                     "accuracy": "max |softmax - float64-accumulated reference|: 6.6e-6 / 6.5e-5 (peaky head) vs 1.1e-5 / 7.1e-5 for "
                                 "the fp32-MFMA mode (tests/test_gpu_forward.py)"}
    halo = 252
    # probability rows produced per step: every window row (windowed) or every time step once + 7 window heads (streamed)
    rows_streamed = BATCH_WINDOWS * CHUNK if args.windowed else reads_per_batch * (READ_LEN + 7 * halo)

    # ---- roofline of the dominant kernel (dilated conv on fp32 MFMA), HIP events on the launch stream
    roof = None
    cpu = None
    if rank == 0:
        n_prof = 2
        be.timer_enable(RD_TIMER_CONV, 11 * n_prof)
        be.timer_enable(RD_TIMER_DECODE, n_prof)
        be.timer_enable(RD_TIMER_HEAD, n_prof)
        for i in range(n_prof):
            step(i)
        be.sync()
        tc = be.timer_read(RD_TIMER_CONV)
        td = be.timer_read(RD_TIMER_DECODE)
        th = be.timer_read(RD_TIMER_HEAD)
        be.timer_enable(RD_TIMER_CONV, 0)
        be.timer_enable(RD_TIMER_DECODE, 0)
        be.timer_enable(RD_TIMER_HEAD, 0)
        # algorithmic FLOPs of the timed launches as accounted by the library: 393 216 per evaluated time step; the
        # streamed forward evaluates fewer rows in the early layers (a head only holds the rows its layer changes)
        flop_per_launch = tc["flops"] / max(1, tc["launches"])
        avg_s = tc["total_ms"] / max(1, tc["launches"]) * 1e-3
        achieved = flop_per_launch / avg_s / 1e12
        peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else F16_MFMA_PEAK_TFLOPS / 3.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("conv_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic,
            "kernel": ("tcn_gemm_kernel<4,3,*> (dilated causal conv 256->256, k=3, v_mfma_f32_32x32x2_f32)" if args.precision == "fp32" else
                       "tcn_gemm_split_kernel<4,3,*> (same conv, 3 x v_mfma_f32_32x32x16_f16 per fp32 product; peak = 2516/3 TFLOP/s)"),
            "flop_per_launch": flop_per_launch, "avg_launch_ms": avg_s * 1e3, "launches_timed": tc["launches"],
            "conv_ms_per_step": tc["total_ms"] / n_prof, "decode_ms_per_step": td["total_ms"] / n_prof,
            "head_ms_per_step": th["total_ms"] / n_prof,
            "timing": "HIP events around every launch of a forward + beam search run on ONE stream after the timed region "
                      "(launches back to back: the kernel's own duration; = profiles/*_kernel_stats.csv, taken with --lanes 1). "
                      "In the timed region the launches of consecutive batches overlap on two lanes, so a launch's bracketed "
                      "duration there also counts the time it shares the chip (profiles/*_kernel_stats_default_2lanes.csv)",
            "decode_timesteps_per_s": float(sum(b[1].sum() for b in batches[:n_prof])) / max(1e-9, td["total_ms"] * 1e-3),
        }
        if world == 1 and not args.no_cpu_baseline:
            ncores = os.cpu_count() or 1
            nr = args.cpu_reads or max(2, min(reads_per_batch, ncores // 4))   # ~10-30 s of CPU work, all cores busy
            try:
                cpu = cpu_baseline(batches[0][2][: nr * 8], batches[0][1][: nr * 8], nr)
            except Exception as e:   # the oracle is test infrastructure: its absence must not cost the GPU line
                print(f"[bench] cpu_baseline unavailable: {e}", file=sys.stderr)
                cpu = None

    if rank == 0:
        out = {
            "metric": "signal samples/s basecalled (chunk=1024, beam=10)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16x3 split (f32 accumulate)", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: synthetic Gaussian int16 reads x 4096 samples (round(N(500,80))), "
                            "MAD-normalised, chunk=1024 step=512 -> 8 windows/read; step = 512 windows (64 reads): "
                            "TCN forward fp32 + chunk-mode CTC beam search W=10 over every window (LM unused in chunk "
                            "mode, reference basecall.py:110-121) + labels to host; random He-normal weights seed 1234",
                "forward": ("windowed: all 512 windows x 1024 rows through the model, as the reference does" if args.windowed else
                            "streamed: each read's time steps are evaluated once (4096 + 7*252 rows per read instead of "
                            "8*1024); probabilities and labels bit-identical to the windowed evaluation "
                            "(tests/test_gpu_reads.py; bench.py --check; --windowed runs the windowed job)"),
                "model_rows_per_step": rows_streamed,
                "chunk_len": CHUNK, "step_size": STEP, "batch_windows": BATCH_WINDOWS, "beam_width": BEAM,
                "decode_type": "chunk", "samples_per_step_per_gpu": samples_per_step, "sharding": "reads per rank, no data-path collective", "startup_comm": comm_kind,
                "pipelining": f"{args.lanes} forward streams taking the steps' batches in turn (independent kernel chains fill each other's last, partial round of workgroups) + 1 decode stream: beam search + label copy-out of a group of {args.decode_group} batches overlaps the next group's forwards; all labels on host at stop",
            },
            "roofline": roof,
        }
        if secondary is not None:
            out["secondary_f16x3"] = secondary
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    for b in batches:
        be.dev_free(b[0])
        be.dev_free(b[3])
    be.close()


if __name__ == "__main__":
    main()
