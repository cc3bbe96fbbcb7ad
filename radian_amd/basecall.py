"""Drop-in for radian/basecall.py: same command line, fast5 in, FASTA out -- the loop body runs on the MI355X.

    python -m radian_amd.basecall fast5_dir fasta_dir [--chunk-len 1024] [--step-size 128] [--batch-size 32]
           [--outlier-clip 4] [--rna-model models/rnamodel_12mer_pc.json | None] [--sig-model models/sig2seq.h5]
           [--sig-config models/sig2seq.yaml] [--beam-width 6] [--decode-type {global,chunk}] [--sig-threshold 0.5]
           [--rna-threshold 0.5] [--context-len 11] [--local]

Flags, defaults, read order, FASTA naming/rotation and the stdout lines follow radian/basecall.py:19-141.
What differs (results do not): reads are batched ACROSS reads for the GPU (the forward is batch independent,
SURVEY F10), so `--batch-size` no longer sizes anything (`--gpu-batch-windows` bounds a device batch); windows are formed on the
device and every time step is evaluated once (bit-identical to evaluating all windows);
`--rna-model None` disables the LM in global mode instead of crashing at decode.py:83; an RNA model that lacks some contexts
loads, and a read whose beam search reaches an absent one raises the reference's KeyError when the driver gets to it -- which is
also what a --context-len that differs from the model's key length does (every context is absent; lazy like decode.py:83, for
--context-len up to 13; longer ones raise at load); `--sig-config none` stands for sig2seq.yaml's dilations without a file (any other
path must exist, as in utilities.get_config; so must --rna-model in both decode types, basecall.py:48-50); `--sig-model` also
accepts `synthetic[:seed]` (seeded He-normal weights: the reference's sig2seq.h5 is not distributed with
the source tree) and packed `.rdnw` blobs; extra flags `--device`, `--gpus`.
One deliberate difference on failure with `--gpus N`: a single-GPU run (like the reference) leaves the reads before a failing one in
reads-*.fasta; the multi-GPU merger (launch.StreamMerger) writes under a hidden directory and removes it when any rank fails -- the
sparse-model KeyError included -- because ranks run ahead of each other and "the reads before the failing one" would be a set with
holes.  A failed multi-GPU job therefore leaves NO reads-*.fasta rather than a valid-looking partial one.
"""
import argparse
import os
import sys
from time import time

import numpy as np

from . import fast5, lm as lm_mod, weights as weights_mod
from .sequence_assembly import consensus_sequence, labels_to_str


def build_parser():
    parser = argparse.ArgumentParser(description="Basecall a nanopore dRNA sequencing run.")
    parser.add_argument("fast5_dir", help="Directory of single/multi fast5 files.")
    parser.add_argument("fasta_dir", help="Directory to output fasta files.")
    parser.add_argument("--local", action="store_true")
    parser.add_argument("--chunk-len", default=1024, type=int)
    parser.add_argument("--step-size", default=128, type=int)
    parser.add_argument("--batch-size", default=32, type=int)
    parser.add_argument("--outlier-clip", default=4, type=int)
    parser.add_argument("--rna-model", default="models/rnamodel_12mer_pc.json")
    parser.add_argument("--sig-model", default="models/sig2seq.h5")
    parser.add_argument("--sig-config", default="models/sig2seq.yaml")
    parser.add_argument("--beam-width", default=6, type=int)
    parser.add_argument("--decode-type", choices=["global", "chunk"], default="global")
    parser.add_argument("--sig-threshold", default=0.5, type=float)
    parser.add_argument("--rna-threshold", default=0.5, type=float)
    parser.add_argument("--context-len", default=11, type=int)
    # extensions
    parser.add_argument("--device", default=0, type=int, help="GPU index (single-process run)")
    parser.add_argument("--gpus", default=1, type=int, help="shard reads over this many GPUs (one process each)")
    parser.add_argument("--device-contexts", default=None, type=int,
                        help="independent device contexts per GPU, taking the batches in turn.  Default 1: a context pipelines by "
                             "itself (forwards of consecutive batches on rotating streams, the beam search of a group of batches on a "
                             "further stream under the next group's forwards); 2 with --no-pipeline")
    parser.add_argument("--no-pipeline", action="store_true",
                        help="one blocking device call per batch (rd_basecall_raw_*) instead of the in-context pipeline "
                             "(rd_pipe_submit_raw_*); overlap then comes from --device-contexts 2 only")
    parser.add_argument("--precision", default="fp32", choices=["fp32", "f16x3", "bf16x3"],
                        help="matrix products of the signal model: exact fp32 MFMA (default); split-f16 products (22-bit operands, fp32 "
                             "accumulation, about 2x faster; DESIGN.md 4.7); three-term bf16 split (every fp32 operand exact, six bf16 "
                             "MFMAs per product, about 1.4x faster; DESIGN.md 4.9)")
    parser.add_argument("--logits", default="f32", choices=["f32", "f16"],
                        help="storage of the softmax rows between the model and the decoder on the device: float32 (the reference's) or "
                             "float16 (not a reference option; BASELINE configs[4])")
    parser.add_argument("--decode-math", default="glibc", choices=["glibc", "fast"],
                        help="arithmetic of the beam search's log / logaddexp: glibc 2.35's operation sequence (default: scores bit-identical "
                             "to the reference's on an x86-64 FMA host, including exactly tied labelings) or this library's faster routines "
                             "(within an ulp of libm's: labelings identical unless two labelings tie within a few ulp; 10-35 %% faster beam "
                             "search)")
    parser.add_argument("--decode-partition", default=-1, type=int,
                        help="global mode, pipelined: CUs of each XCD kept free of forward workgroups for the beam search (a read's "
                             "search is one serial chain; beside conv waves a step runs ~8x slower).  -1: by beam width (4, 8 above W = 25, 12 above W = 64, 16 above W = 128: multiples of four -- a masked queue is dealt over four shader engines), 0: off -- on steady exact-fp32 jobs at beam <= 25 the search "
                             "then runs on every CU beside the forward and the job 3-8 %% faster (measured, profiles/r06_policy_probe.txt), but long-read jobs with an RNA "
                             "model and the f16x3 / bf16x3 modes lose: not the default")
    parser.add_argument("--lm-hashed-context", action="store_true",
                        help="global mode: accept a --context-len longer than the RNA model's k-mers (up to 256) by addressing the model's "
                             "table with a hash of the context (a synthetic long-context LM: no reference behaviour -- the reference raises "
                             "KeyError at decode.py:83)")
    parser.add_argument("--stitch-workers", default=None, type=int,
                        help="chunk mode: host threads of the native fragment stitch (default: the usable cores - 2, at most 16); with "
                             "--no-pipeline: worker processes of the pure-Python stitch (default min(4, cores / 4); 0: on the driver's thread)")
    parser.add_argument("--cpu-affinity", default="auto", choices=["auto", "none"],
                        help="--gpus N: auto = every rank binds itself, before its first GPU call, to an even share of the usable cores -- NUMA-local "
                             "to its GPU when /sys tells -- and sizes its threads from that share (radian_amd/hostbudget.py); none = stay where the "
                             "launcher put the process")
    parser.add_argument("--queue-block", default=256, type=int,
                        help="--gpus N: reads per claim of the per-node work queue (0: static round-robin by read index)")
    parser.add_argument("--gpu-batch-windows", default=None, type=int,
                        help="device batch size across reads, in windows (chunk mode) or chunk_len-row units (global mode).  Default: 4096 "
                             "(about 1.3 GB of activations per forward stream in fp32); with --no-pipeline a global-mode batch that holds "
                             "long reads grows (up to 16384 units, about 50 GB per context) until its forward covers the longest read's "
                             "beam search, which is one serial chain per read -- the pipeline groups batches on the device for that instead")
    return parser


def load_dilations(sig_config):
    """models/sig2seq.yaml `model.tcn` -> per-block dilations; checks the fixed geometry (sig2seq.yaml:34-49).  A path that does not
    exist raises FileNotFoundError like utilities.get_config's open() (utilities.py:16-18, basecall.py:60); `--sig-config none` is this
    CLI's explicit way of asking for sig2seq.yaml's dilations without a file."""
    if not sig_config or sig_config.lower() == "none":
        return weights_mod.DEFAULT_DILATIONS
    import yaml
    with open(sig_config) as f:
        cfg = yaml.safe_load(f)
    m = cfg["model"]
    t = m["tcn"]
    if (t["nb_filters"], t["kernel_size"], m["relu_units"], m["softmax_units"]) != (256, 3, 128, 5):
        raise ValueError("sig-config geometry differs from sig2seq.yaml's (256 filters, k=3, 128 relu units, 5 classes)")
    if t.get("use_skip_connections") or t.get("use_batch_norm") or t.get("dropout_rate", 0.0) not in (0, 0.0) \
            or t.get("padding") != "causal" or t.get("activation") != "relu":
        raise ValueError("sig-config uses TCN options this backend does not implement (skip connections / batch norm / dropout / non-causal)")
    return tuple(int(d) for d in t["dilations"]) * int(t.get("nb_stacks", 1))


def load_sig_model(spec, dilations):
    """Flat float32 parameters in load_weights order (radian/model.py:42-45)."""
    if spec.startswith("synthetic"):
        seed = int(spec.split(":", 1)[1]) if ":" in spec else 1234
        return weights_mod.synthetic_weights(seed=seed, dilations=dilations)
    if spec.endswith(".rdnw"):
        with open(spec, "rb") as f:
            flat, dil = weights_mod.unpack_blob(f.read())
        if tuple(dil) != tuple(dilations):
            raise ValueError(f"{spec}: dilations {dil} differ from the config's {dilations}")
        return flat
    from .h5weights import read_keras_weights
    return read_keras_weights(spec, dilations)


class FastaWriter:
    """reads-{n}.fasta with a new file after every 1000 reads (basecall.py:65-67,129-141)."""

    def __init__(self, fasta_dir):
        self.dir = fasta_dir
        self.n = 0
        self.i = 0
        self.f = open(f"{fasta_dir}/reads-{self.n}.fasta", "w")

    def write(self, read_id, sequence):
        self.write_final(read_id, sequence[::-1])  # reversed to be 5' to 3' (basecall.py:129)

    def write_final(self, read_id, sequence):
        """a record whose sequence is already 5' to 3' (the multi-GPU ranks reverse before they hand over)"""
        self.f.write(f">{read_id}\n{sequence}\n")
        self.i += 1
        if self.i == 1000:
            self.f.close()
            self.n += 1
            self.f = open(f"{self.dir}/reads-{self.n}.fasta", "w")
            self.i = 0

    def close(self):
        self.f.close()


STATUS_MESSAGES = {1: "MAD is zero, issue with signal.", 2: "Signal must not be empty to normalise"}  # preprocess.py:25-26,47-48


def report_skipped(read_id, status):
    """What basecall.py:79-82 prints when mad_normalise raises ValueError."""
    print((STATUS_MESSAGES[int(status)],))
    print(f"{read_id} signal issue, skipping this read.")


def device_batch(be, batch, args, use_lm):
    """batch: list of (read_id, raw int16 signal).  Raw reads go to the device: MAD normalisation, windowing, forward
    (every time step evaluated once), assembly and beam search run there.  Returns (label arrays, status per read):
    chunk mode -> per read the list of per-window fragments, global mode -> per read one labeling."""
    raws = [raw for _, raw in batch]
    if args.decode_type == "global":
        return be.basecall_raw_global(raws, args.outlier_clip, args.chunk_len, args.step_size, args.beam_width, use_lm,
                                      args.sig_threshold, args.rna_threshold)
    return be.basecall_raw_chunk(raws, args.outlier_clip, args.chunk_len, args.step_size, args.beam_width)


def _stitch_reads(frag_lists):
    """worker-process task: fragments (label arrays) of a few reads -> their consensus strings"""
    return [consensus_sequence([labels_to_str(f) for f in frags]) for frags in frag_lists]


def make_stitch_pool(n_workers):
    """Process pool for the chunk-mode string stitch in pure Python (difflib, the reference's own code path; one interpreter
    cannot keep up with one GPU) -- the --no-pipeline driver's host stage; the pipelined driver stitches natively
    (sequence_assembly.consensus_batch).  Must be created BEFORE the process touches the GPU: the workers are spawned fresh
    interpreters that never load the HIP library."""
    if n_workers is None:
        n_workers = min(4, max(1, (os.cpu_count() or 2) // 4))
    if n_workers <= 0:
        return None
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    pool = ProcessPoolExecutor(max_workers=n_workers, mp_context=multiprocessing.get_context("spawn"))
    try:
        for f in [pool.submit(_stitch_reads, []) for _ in range(n_workers)]:   # start every worker now
            f.result()
    except Exception as e:   # e.g. a __main__ that cannot be re-imported by the workers: stitch on the host thread instead
        print(f"stitch workers unavailable ({type(e).__name__}: {e}); stitching on the driver thread", file=sys.stderr)
        pool.shutdown(wait=False, cancel_futures=True)
        return None
    return pool


def host_finish(labels, args):
    """labels of one read -> its (un-reversed) sequence; chunk mode stitches the fragments (basecall.py:122-123)."""
    if args.decode_type == "global":
        return labels_to_str(labels)
    return consensus_sequence([labels_to_str(f) for f in labels])


def _owned_reads(args, reads, shard, queue, sources):
    """(key, read) pairs this process basecalls, in input order.  key orders the merged output: the read's index in the
    enumeration, or (file index, read index in the file) with file-level sources."""
    rank, world = shard
    if sources is not None:
        if queue is not None:          # dist.FileReadQueue: claim blocks of reads file by file; open only claimed files
            last = None
            for fi, lo, hi in queue.claims(sources):
                if last is not None and last != fi:
                    sources[last].close()
                last = fi
                for ri, read in sources[fi].reads(lo, hi):
                    yield (fi, ri), read
            if last is not None:
                sources[last].close()
        else:                          # static: whole files round-robin over the ranks
            for fi, src in enumerate(sources):
                if fi % world != rank:
                    continue
                for ri, read in src.reads(0, src.n_reads()):
                    yield (fi, ri), read
                src.close()
        return
    if reads is None:
        reads = fast5.iter_directory(args.fast5_dir)
    for idx, read in enumerate(reads):
        if (not queue.owns(idx)) if queue is not None else (idx % world != rank):
            continue
        yield idx, read


def _read_ahead(pairs, max_reads=8192, max_samples=48 << 20, block=128):
    """(key, read) pairs -> (key, read_id, raw signal) with the HDF5 work -- group walks, dataset reads, the work queue's
    claims -- done on a reader thread that runs ahead of the driver loop by at most max_reads reads / max_samples samples.
    libhdf5 is called through ctypes, which releases the interpreter lock for the call, so parsing (~60 us of a ~100-us read in
    libhdf5: tools/host_feed_bench.py) overlaps the loop's batching and the finishing of earlier batches; every libhdf5 call of
    the process stays on this one thread.  Reads change hands in blocks of `block`: handing over every read makes the two
    threads fight for the interpreter lock at every ctypes call (measured: half the rate of no thread at all).  Order is
    preserved; an exception of the producer is re-raised by the consumer at the position where it happened (every read before it is delivered first)."""
    import queue as queue_mod
    import threading
    q = queue_mod.Queue()
    room = threading.Condition()
    state = {"reads": 0, "samples": 0, "stop": False}

    def produce():
        blk, n_s = [], 0
        try:
            for key, read in pairs:
                raw = np.asarray(read.get_raw_data())
                blk.append((key, read.read_id, raw))
                n_s += raw.shape[0]
                if len(blk) >= block or n_s >= max_samples // 4:
                    with room:
                        while not state["stop"] and state["reads"] > 0 and (state["reads"] + len(blk) > max_reads or state["samples"] + n_s > max_samples):
                            room.wait(0.1)
                        if state["stop"]:
                            return
                        state["reads"] += len(blk)
                        state["samples"] += n_s
                    q.put((blk, n_s))
                    blk, n_s = [], 0
            if blk:
                q.put((blk, n_s))
            q.put(None)
        except BaseException as e:      # (handed to the consumer -- behind the reads that were read before it: the reference has
            if blk:                     # basecalled and written every read ahead of the failing one, basecall.py:70-141)
                q.put((blk, n_s))
            q.put(e)
        finally:
            close = getattr(pairs, "close", None)
            if close is not None:
                try:
                    close()             # (a generator: its finally blocks close the open fast5 file on this thread)
                except Exception:
                    pass

    t = threading.Thread(target=produce, name="radian-read-ahead", daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            blk, n_s = item
            with room:
                state["reads"] -= len(blk)
                state["samples"] -= n_s
                room.notify()
            yield from blk
    finally:
        with room:
            state["stop"] = True
            room.notify_all()
        t.join(timeout=30.0)


def run(args, be, reads=None, writer=None, shard=(0, 1), stitch_pool=None, queue=None, sources=None, on_result=None, stats=None):
    """The driver loop (basecall.py:69-141) over `reads` (default: every read under args.fast5_dir), or over file-level
    `sources` (fast5.Fast5Source per file: the multi-GPU launcher's form).
    shard=(rank, world): this process handles reads whose index % world == rank -- or, with a queue shared by the ranks
    of the node (dist.WorkQueue over read indices / dist.FileReadQueue over files then reads), the blocks it claims --
    and returns [(key, read_id, sequence)] when neither a writer nor on_result takes the records (an empty list otherwise: a
    consumed record is not kept); on_result(key, read_id, sequence), if given, receives each result as it is finished (in order).
    stats: a dict that receives what this process took on -- {"reads", "samples"} (per-rank rates of a multi-GPU job).
    `be` is one Backend or a list of Backends on the same GPU (independent rd_ctx / HIP streams): batches go to them
    round robin on one thread each, so the MFMA-bound forward of one batch overlaps the latency-bound beam search of
    the previous one.  Reading/batching (HDF5 through ctypes), the device calls and the host work on the results
    (string stitch, FASTA write) run concurrently; output order is the input order."""
    from concurrent.futures import ThreadPoolExecutor
    backends = list(be) if isinstance(be, (list, tuple)) else [be]
    # in-context pipeline (rd_pipe_submit_raw_*): a device call only QUEUES the batch and returns a ticket; results are
    # collected a few batches later, in order
    pipelined = not getattr(args, "no_pipeline", False) and all(hasattr(b, "pipe_submit_raw") for b in backends)
    if args.step_size <= 0:
        raise ValueError("Step size must be > 0")            # preprocess.py:5-8
    if args.step_size > args.chunk_len:
        raise ValueError("Step size must be <= window size")
    use_lm = getattr(args, "_lm_loaded", False) and args.decode_type == "global"
    rank, world = shard
    # a record that a writer or on_result has consumed is written and forgotten, as the reference does (basecall.py:129): the list
    # only grows when it is the one way out (a whole run's sequences are O(total bases): ~1 GB per million 1-kb reads)
    collect_results = writer is None and on_result is None
    results = []
    batch, batch_idx, n_win = [], [], 0
    dev_pools = [ThreadPoolExecutor(max_workers=1) for _ in backends]   # an rd_ctx is not thread-safe: one thread each
    host_pool = ThreadPoolExecutor(max_workers=1)                        # host post-processing, strictly in order
    in_flight, finishing = [], []
    n_submitted = 0

    failed = []     # the host stage stopped at a read (the reference's KeyError): nothing after that read may come out

    def finish(b, b_idx, labels, status, dur):
        if failed:
            return
        try:
            _finish(b, b_idx, labels, status, dur)
        except BaseException as e:
            failed.append(e)
            raise

    def _finish(b, b_idx, labels, status, dur):
        seqs = None
        if isinstance(labels, tuple):
            # pipelined chunk mode: the label matrix as the device left it -> consensus strings of every read in one native
            # call (simple_assembly + argmax, sequence_assembly.py:19-48 / basecall.py:122-123, on host threads)
            from .sequence_assembly import consensus_batch
            lab2d, lens, nw = labels
            allseq = consensus_batch(lab2d, lens, nw, threads=args.stitch_workers)
            seqs = iter([sq for sq, st in zip(allseq, status) if st == 0])
            labels = [None] * len(b)
        elif stitch_pool is not None and args.decode_type == "chunk":
            # the reads of the batch in slices, one task each; results come back in order
            ok = [lab for lab, st in zip(labels, status) if st == 0]
            per = max(1, -(-len(ok) // (4 * stitch_pool._max_workers)))
            futs = [stitch_pool.submit(_stitch_reads, ok[i:i + per]) for i in range(0, len(ok), per)]
            seqs = iter([sq for f in futs for sq in f.result()])
        for (rid, _), idx, lab, st in zip(b, b_idx, labels, status):
            if st != 0:
                report_skipped(rid, st)
                continue
            if lab is None and seqs is None:
                # sparse RNA model: this read's beam search kept a labeling whose context the model does not hold -- the
                # reference dies here with KeyError (decode.py:83, uncaught in basecall.py:70-141); the reads before it are out
                raise KeyError(f"read {rid}: the RNA model holds no entry for a context of the beam search (radian/decode.py:83)")
            seq = next(seqs) if seqs is not None else host_finish(lab, args)
            if writer is not None:
                writer.write(rid, seq)
            if on_result is not None:
                on_result(idx, rid, seq)
            if collect_results:
                results.append((idx, rid, seq))
            print(f"Basecalled read {rid} in {dur:.2f} sec.")

    def on_device(backend, b):
        t0 = time()
        if pipelined:
            return backend.pipe_submit_raw(args.decode_type, [raw for _, raw in b], args.outlier_clip, args.chunk_len, args.step_size,
                                           args.beam_width, use_lm, args.sig_threshold, args.rna_threshold), t0
        labels, status = device_batch(backend, b, args, use_lm)
        return labels, status, (time() - t0) / len(b)

    def collect(ticket, t0, n):
        if args.decode_type == "chunk":                 # blocks until the batch's group has been decoded
            lab2d, lens, nw, status = ticket.result_raw()
            return (lab2d, lens, nw), status, (time() - t0) / n
        labels, status = ticket.result()
        return labels, status, (time() - t0) / n

    def retire(keep):
        """hand the oldest device results to the host stage (in submission order) until `keep` batches are in flight"""
        while len(in_flight) > keep:
            fut, b, b_idx, k = in_flight.pop(0)
            res = fut.result()                          # re-raises device-side errors
            if pipelined:                               # (the context's calls stay on its own thread)
                res = dev_pools[k].submit(collect, res[0], res[1], len(b)).result()
            labels, status, dur = res
            finishing.append(host_pool.submit(finish, b, b_idx, labels, status, dur))
            while len(finishing) > 2:
                finishing.pop(0).result()               # re-raises host-side errors

    # Global mode: a read's beam search is a serial chain of one time step per sample (~1.7 us at beam <= 12) that
    # only overlaps the OTHER context's forward (~29 M rows/s): a batch should hold ~128 forward rows per chain step of its longest
    # read (measured: 768 reads x 100 k samples, 19 M samples/s at 4096, 25 M at 16384; tools/cli_long.py).
    longest = 0
    chain_rows = 128 if args.beam_width <= 12 else 200 if args.beam_width <= 25 else 300

    # A unit (chunk_len rows) costs three activation tensors of 1 KiB per row (1.5 KiB with bf16x3) per forward stream, plus
    # probabilities / assembled matrix / trie; the automatic batch size never asks for more than a third of the device memory
    # that is free now (a smaller-HBM part, other processes on the GPU) -- an explicit --gpu-batch-windows is taken as given.
    mem_cap = None
    if hasattr(backends[0], "mem_info"):
        try:
            free, _total = backends[0].mem_info()
            per_unit = args.chunk_len * ((1536 if getattr(args, "precision", "fp32") == "bf16x3" else 1024) * 3 + 80 + 24 * max(1, args.beam_width))
            streams = len(backends) * (2 if pipelined else 1)
            mem_cap = max(64, int(free / 3 / (per_unit * streams)))
        except Exception:
            mem_cap = None

    def batch_limit(longest_read):
        n = _batch_limit(longest_read)
        return n if mem_cap is None or args.gpu_batch_windows is not None else min(n, mem_cap)

    def _batch_limit(longest_read):
        if args.gpu_batch_windows is not None:
            return args.gpu_batch_windows
        if args.decode_type != "global" or pipelined:   # (the pipeline covers a long read's chain by grouping batches)
            return 4096
        return max(4096, min(16384, longest_read * chain_rows // args.chunk_len))

    def flush():
        nonlocal batch, batch_idx, n_win, n_submitted
        if not batch:
            return
        k = n_submitted % len(backends)
        n_submitted += 1
        in_flight.append((dev_pools[k].submit(on_device, backends[k], batch), batch, batch_idx, k))
        # pipelined: enough batches in flight that a global-mode group can gather the rows that cover its longest read's
        # chain (the host side of a batch is a few MB)
        retire(len(backends) * ((24 if args.decode_type == "global" else 8) if pipelined else 1))
        batch, batch_idx, n_win = [], [], 0

    def drain():
        retire(0)
        for f in finishing:
            f.result()
        finishing.clear()

    try:
        for idx, read_id, raw in _read_ahead(_owned_reads(args, reads, shard, queue, sources)):
            n = raw.shape[0]
            if stats is not None:
                stats["reads"] = stats.get("reads", 0) + 1
                stats["samples"] = stats.get("samples", 0) + int(n)
            if n == 0:
                flush()                  # keep the reference's message order
                drain()
                report_skipped(read_id, 2)
                continue
            if args.decode_type == "chunk":
                nw = (0 if n < args.chunk_len else (n - args.chunk_len) // args.step_size + 1) + 1   # windows decoded
            else:
                nw = -(-n // args.chunk_len)   # global: the device evaluates N rows and decodes one sequence per read
            if batch and n_win + nw > batch_limit(max(longest, n)):
                flush()
                longest = 0
            batch.append((read_id, raw))
            batch_idx.append(idx)
            n_win += nw
            longest = max(longest, n)
        flush()
        drain()
    finally:
        for p in dev_pools:
            p.shutdown(wait=True)
        host_pool.shutdown(wait=True)
    return results


ARTIFACT_CACHE = "artifacts.npz"


def save_artifacts(art, directory):
    """the parsed artefacts for the ranks of a multi-GPU job (launch.run_multi_gpu): parsing the real RNA model -- 4^11 keys,
    ~400 MB of JSON -- takes ~25 s (tools/lm_load_bench.py); the launcher has done it to validate the arguments, so the rank that
    feeds the broadcast reads the arrays back instead of parsing again"""
    np.savez(os.path.join(directory, ARTIFACT_CACHE), dilations=np.asarray(art["dilations"], dtype=np.int64), weights=art["weights"],
             lm_table=art["lm_table"] if art["lm_table"] is not None else np.zeros((0, 4)), lm_k=art["lm_k"],
             lm_hashed_order=art.get("lm_hashed_order", 0), lm_absent=art.get("lm_absent", 0))


def load_artifacts(args, cache_dir=None):
    """Host-only half of basecall.py:47-62: parse / validate the signal model and (global mode) the RNA model.
    Returns {"dilations", "weights", "lm_table", "lm_k"}; raises what the reference would before any GPU is touched: FileNotFoundError
    for a --rna-model (in EITHER decode type: basecall.py:48-50 opens it unconditionally) or --sig-config file that does not exist.
    A --context-len that differs from the model's key length is NOT an error here: the reference fails lazily, with KeyError at
    decode.py:83 on the first read whose beam search keeps a labeling of context-len labels, after basecalling the shorter reads before
    it -- art["lm_absent"] asks for exactly that model (Backend.load_lm_absent); beyond 13 labels, where no dense image exists, the
    KeyError comes at load instead.
    cache_dir: where the launcher left its parse of the same arguments (save_artifacts)."""
    if cache_dir is not None and os.path.exists(os.path.join(cache_dir, ARTIFACT_CACHE)):
        with np.load(os.path.join(cache_dir, ARTIFACT_CACHE)) as z:
            art = {"dilations": tuple(int(d) for d in z["dilations"]), "weights": z["weights"],
                   "lm_table": z["lm_table"] if z["lm_table"].shape[0] else None, "lm_k": int(z["lm_k"])}
            if int(z["lm_hashed_order"]):
                art["lm_hashed_order"] = int(z["lm_hashed_order"])
            if "lm_absent" in z.files and int(z["lm_absent"]):
                art["lm_absent"] = int(z["lm_absent"])
        return art
    dil = load_dilations(args.sig_config)
    art = {"dilations": dil, "weights": load_sig_model(args.sig_model, dil), "lm_table": None, "lm_k": 0}
    if args.rna_model != "None":
        if not os.path.exists(args.rna_model):
            raise FileNotFoundError(2, "No such file or directory", args.rna_model)   # basecall.py:49, whatever --decode-type says
        table, k = lm_mod.load_json(args.rna_model)
        if args.decode_type == "global":
            if k != args.context_len and getattr(args, "lm_hashed_context", False):
                if not (k < args.context_len <= 256):   # (a shorter context has a dense table of its own: not what the flag is for)
                    raise ValueError(f"--lm-hashed-context: --context-len must be longer than the RNA model's {k}-label contexts, at most 256")
                if args.beam_width > 64:                # (hashed contexts exist in the wave-per-sequence kernels only; said here, before any read is touched)
                    raise ValueError("--lm-hashed-context: --beam-width must be at most 64 (wider beams run on the general kernel, which has no hashed contexts)")
                art["lm_table"], art["lm_k"], art["lm_hashed_order"] = table, args.context_len, k
                return art
            if k != args.context_len:
                if not 1 <= args.context_len <= 13:
                    raise KeyError(f"--context-len {args.context_len} does not match the RNA model's context length {k} "
                                   "(the reference fails with KeyError at decode.py:83)")
                art["lm_absent"], art["lm_k"] = args.context_len, args.context_len
                return art
            art["lm_table"], art["lm_k"] = table, k
    return art


def apply_artifacts(args, be, art, clone_from=None):
    """Device half: hand the parsed artefacts to a Backend -- or, for a further context of the same process, take the device
    images of the Backend that already holds them (clone_from)."""
    be.set_precision(getattr(args, "precision", "fp32"))
    be.set_logits(getattr(args, "logits", "f32"))
    be.set_decode_math(getattr(args, "decode_math", "glibc"))
    if hasattr(be, "set_decode_partition"):
        be.set_decode_partition(getattr(args, "decode_partition", -1))
    if clone_from is not None:
        be.clone_artifacts_from(clone_from)
        return
    be.load_weights(art["weights"], art["dilations"])
    args._lm_loaded = False
    if art.get("lm_absent"):
        be.load_lm_absent(art["lm_absent"])
        args._lm_loaded = True
    elif art["lm_table"] is not None:
        if art.get("lm_hashed_order"):
            be.load_lm_hashed(art["lm_table"], art["lm_hashed_order"], art["lm_k"])
        else:
            be.load_lm(art["lm_table"], art["lm_k"])
        args._lm_loaded = True


def n_contexts(args):
    """device contexts per GPU: 1 when a context pipelines by itself, 2 with --no-pipeline, or what --device-contexts says"""
    if args.device_contexts is not None:
        return max(1, args.device_contexts)
    return 2 if args.no_pipeline else 1


def setup_backend(args, be):
    """Load weights and (when given) the RNA model into a Backend; mirrors basecall.py:47-62."""
    apply_artifacts(args, be, load_artifacts(args))


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.gpus > 1:
        from .launch import run_multi_gpu
        return run_multi_gpu(args, argv if argv is not None else sys.argv[1:])
    pool = make_stitch_pool(args.stitch_workers) if args.decode_type == "chunk" and args.no_pipeline else None   # before the GPU is touched
    from .backend import Backend
    art = load_artifacts(args)
    bes = [Backend(args.device) for _ in range(n_contexts(args))]
    for i, b in enumerate(bes):
        apply_artifacts(args, b, art, clone_from=bes[0] if i else None)
    del art
    writer = FastaWriter(args.fasta_dir)
    try:
        run(args, bes, writer=writer, stitch_pool=pool)
    finally:
        writer.close()  # basecall.py:141
        for b in bes:
            b.close()
        if pool is not None:
            pool.shutdown()


if __name__ == "__main__":
    main()
