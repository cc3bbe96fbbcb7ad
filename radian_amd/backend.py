"""Python face of the C ABI: one Backend == one rd_ctx == one GPU rank."""
import ctypes

import numpy as np

from . import _lib
from .weights import pack_blob, DEFAULT_DILATIONS

RD_TIMER_CONV, RD_TIMER_DECODE, RD_TIMER_HEAD, RD_TIMER_IN = 0, 1, 2, 3


class RadianHipError(RuntimeError):
    pass


def lib_path():
    return _lib.LIB_PATH


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


MISSING_CONTEXT = -1   # RD_LEN_MISSING_CONTEXT (include/radian_hip.h)


def _labels_of(labels, off, n):
    """one sequence's labels out of the flat label buffer -- or None where the library reports RD_LEN_MISSING_CONTEXT: the
    read's beam search looked up a context that the sparse RNA model does not hold (the reference raises KeyError there,
    radian/decode.py:83; radian_amd.basecall does when it reaches the read)"""
    if n == MISSING_CONTEXT:
        return None
    return labels[off: off + n].copy()


def device_count():
    """visible HIP devices (rd_device_count)"""
    n = ctypes.c_int(0)
    L = _lib.load()
    if L.rd_device_count(ctypes.byref(n)) != 0:
        raise RadianHipError(L.rd_last_error().decode())
    return n.value


def device_for_rank(local_rank):
    """The device of a rank of a one-process-per-GPU job: its local rank -- or, when the launcher has narrowed every process's
    view (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES with fewer devices than local ranks), the local rank modulo what is visible."""
    n = device_count()
    if n < 1:
        raise RadianHipError("no HIP device is visible to this process")
    return local_rank if local_rank < n else local_rank % n


class PipeTicket:
    """One batch queued on a Backend's reads-level pipeline (Backend.pipe_submit_raw).  The arrays the library writes into
    live here until the batch is delivered."""

    def __init__(self, be, decode_type, off, n):
        self.be, self.decode_type, self.off, self.n = be, decode_type, off, n
        self.status = np.zeros(n, dtype=np.int32)
        self.labels = self.lens = self.label_off = self.nw = None
        self.seq = 0

    def done(self):
        return self.be.pipe_progress(0) >= self.seq

    def wait(self):
        if self.be.pipe_progress(self.seq) < self.seq:
            raise RadianHipError("pipeline did not deliver the awaited batch")

    def result_raw(self):
        """chunk mode, blocks until delivered: (label matrix uint8 [n_windows, chunk_len], lengths int32 [n_windows], windows
        per read, status) -- what radian_amd.sequence_assembly.consensus_batch takes, without a Python object per window"""
        self.wait()
        return self.labels, self.lens, self.nw, self.status

    def result(self):
        """(labels, status) as basecall_raw_global / basecall_raw_chunk return them; blocks until delivered"""
        self.wait()
        if self.decode_type == "global":
            return [_labels_of(self.labels, self.off[r], self.lens[r]) for r in range(self.n)], self.status
        out, w = [], 0
        for n in self.nw:
            out.append([self.labels[w + i, : self.lens[w + i]].copy() for i in range(n)])
            w += n
        return out, self.status


class Backend:
    """Owns an rd_ctx.  All compute goes to the HIP library; nothing here falls back to the CPU."""

    def __init__(self, device_id=0):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        self._h = None
        self._check(self._L.rd_create(int(device_id), ctypes.byref(h)))
        self._h = h
        self.device_id = device_id
        self.lm_k = None
        # undelivered PipeTickets by submit number: the library writes a batch's labels / lengths / status into the ticket's
        # arrays when it DELIVERS the batch (rd_pipe_progress / rd_pipe_flush, possibly on behalf of another ticket's wait),
        # so the arrays must outlive a ticket the caller dropped
        self._tickets = {}

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise RadianHipError(f"[rd error {rc}] " + self._L.rd_last_error().decode("utf-8", "replace"))

    def close(self):
        if self._h is not None:
            self._L.rd_destroy(self._h)      # (does not deliver: nothing is written into ticket arrays after this)
            self._h = None
        self._tickets.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self._L.rd_sync(self._h))

    def set_precision(self, mode):
        """'fp32' (default, exact fp32 MFMA), 'f16x3' (split-f16 matrix products, 22-bit operands) or 'bf16x3' (three-term
        bf16 split: every fp32 operand exact, six bf16 MFMAs per product)."""
        code = {"fp32": 0, "f16x3": 1, "bf16x3": 2}[mode] if isinstance(mode, str) else int(mode)
        self._check(self._L.rd_set_precision(self._h, code))

    def set_conv_shape(self, shape):
        """0 (default): 128-row tiles, two 256-thread workgroups per CU; 1: 256-row tiles, one 512-thread workgroup per CU (fp32 mode)."""
        self._check(self._L.rd_set_conv_shape(self._h, int(shape)))

    def split3(self, values):
        """Device-side three-term bf16 split of float32 values -> uint16 [3, n] bit patterns (hi, mid, lo)."""
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        out = np.zeros((3, v.size), dtype=np.uint16)
        self._check(self._L.rd_split3(self._h, _p(v), v.size, _p(out)))
        return out

    @property
    def max_beam_width(self):
        return self._L.rd_decode_max_width()

    # ------------------------------------------------------------------ artefacts
    def load_weights(self, flat, dilations=DEFAULT_DILATIONS):
        """model.load_weights (radian/model.py:44): flat float32 parameters in Keras order."""
        blob = pack_blob(flat, dilations)
        self.load_weights_blob(blob)

    def load_weights_blob(self, blob):
        buf = ctypes.create_string_buffer(blob, len(blob))
        self._check(self._L.rd_load_weights(self._h, ctypes.cast(buf, ctypes.c_void_p), len(blob)))

    def load_lm(self, table, k):
        """LM table [4^k,4] float64 (radian/basecall.py:48-57); rows of NaN mark contexts a sparse model does not hold
        (lm.table_from_dict); None unloads."""
        if table is None:
            self._check(self._L.rd_load_lm(self._h, None, 0))
            self.lm_k = None
            return
        table = np.ascontiguousarray(table, dtype=np.float64)
        if table.shape != (4 ** k, 4):
            raise ValueError(f"LM table must be [4^{k},4], got {table.shape}")
        self._check(self._L.rd_load_lm(self._h, _p(table), int(k)))
        self.lm_k = k

    def load_lm_absent(self, k):
        """--context-len k with an RNA model whose keys have another length (rd_load_lm_absent): every lookup is the reference's KeyError."""
        self._check(self._L.rd_load_lm_absent(self._h, int(k)))
        self.lm_k = k

    def load_lm_hashed(self, table, table_order, context_len):
        """Synthetic LM for contexts longer than a dense table can index (rd_load_lm_hashed): table [4^table_order, 4],
        row = hash of the last `context_len` labels (<= 256)."""
        table = np.ascontiguousarray(table, dtype=np.float64)
        if table.shape != (4 ** table_order, 4):
            raise ValueError(f"LM table must be [4^{table_order},4], got {table.shape}")
        self._check(self._L.rd_load_lm_hashed(self._h, _p(table), int(table_order), int(context_len)))
        self.lm_k = context_len

    def set_logits(self, mode):
        """'f32' (default) or 'f16': storage of the softmax rows on the reads-level paths (rd_set_logits)."""
        self._check(self._L.rd_set_logits(self._h, {"f32": 0, "f16": 1}[mode] if isinstance(mode, str) else int(mode)))

    def set_trie_budget(self, nbytes):
        """workspace one beam-search launch may ask for (rd_set_trie_budget; 0 = the default 24 GiB): launches beyond it run as several
        runs of sequences sharing the workspace.  No effect on results -- tests set it small to exercise the cut."""
        self._check(self._L.rd_set_trie_budget(self._h, int(nbytes)))

    def set_conv_fuse(self, on):
        """block 0's first conv inside its second conv's kernel (default) or as a kernel of its own (rd_set_conv_fuse); same bits"""
        self._check(self._L.rd_set_conv_fuse(self._h, 1 if on else 0))

    def set_decode_form(self, form):
        """'auto' (default); 'waves' / 'lanes': launch shape for widths above 12; 'two' / 'one': widths up to 12 always / never as
        two sequences per wave; 'queue': every launch through the work-queue kernel with few resident workgroups (rd_set_decode_form)."""
        self._check(self._L.rd_set_decode_form(self._h, {"auto": 0, "waves": 1, "lanes": 2, "two": 3, "one": 4, "queue": 5}[form] if isinstance(form, str) else int(form)))

    def set_decode_partition(self, cus_per_xcd):
        """CUs per XCD reserved for the beam search of the global-mode reads pipeline (rd_set_decode_partition): -1 = by beam
        width (default), 0 = off."""
        self._check(self._L.rd_set_decode_partition(self._h, int(cus_per_xcd)))

    def set_decode_math(self, mode):
        """'glibc' (default: scores bit-identical to the reference's) or 'fast': arithmetic of the beam search's log / logaddexp
        (rd_set_decode_math)."""
        self._check(self._L.rd_set_decode_math(self._h, {"fast": 0, "glibc": 1}[mode] if isinstance(mode, str) else int(mode)))

    # ------------------------------------------------------------------ seams (host arrays)
    def forward(self, windows):
        """sig_model.predict (radian/basecall.py:91,93): [n,T] -> [n,T,5] float32."""
        windows = np.ascontiguousarray(windows, dtype=np.float32)
        if windows.ndim != 2:
            raise ValueError("windows must be [n_windows, chunk_len]")
        n, T = windows.shape
        probs = np.empty((n, T, 5), dtype=np.float32)
        self._check(self._L.rd_forward(self._h, _p(windows), n, T, _p(probs)))
        return probs

    def assemble(self, probs, pad, step):
        """assemble_matrices after the pad trim (radian/basecall.py:96,100).  probs [nW,T,5] of one read."""
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        nW, T, _ = probs.shape
        cap = (nW - 1) * step + T
        out = np.empty((max(cap, 1), 5), dtype=np.float64)
        n_rows = ctypes.c_int64(0)
        is64 = ctypes.c_int(0)
        self._check(self._L.rd_assemble(self._h, _p(probs), nW, T, int(pad), int(step), _p(out), out.shape[0],
                                        ctypes.byref(n_rows), ctypes.byref(is64)))
        out = out[: n_rows.value]
        return out if is64.value else out.astype(np.float32)

    def forward_reads(self, signals, chunk_len, step):
        """sig_model.predict over the windows of whole normalised reads (radian/basecall.py:83-93) through the streamed
        evaluation (rd_forward_reads): -> per read an array [nW, chunk_len, 5] float32 (rows the pad trim drops are zero)"""
        sigs = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in signals]
        off = np.zeros(len(sigs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([s_.shape[0] for s_ in sigs])
        nws = [self.count_windows(int(s_.shape[0]), chunk_len, step) for s_ in sigs]
        tot = int(sum(nws))
        flat = np.concatenate(sigs) if sigs else np.zeros(0, dtype=np.float32)
        out = np.zeros((tot, chunk_len, 5), dtype=np.float32)
        n = ctypes.c_int64(0)
        self._check(self._L.rd_forward_reads(self._h, _p(flat), _p(off), len(sigs), int(chunk_len), int(step), _p(out), tot, ctypes.byref(n)))
        assert n.value == tot
        res, w = [], 0
        for k in nws:
            res.append(out[w:w + k])
            w += k
        return res

    def decode_batch(self, mats, seq_off, seq_len, beam_width, use_lm=False, s_threshold=0.0, r_threshold=0.0,
                     with_scores=False):
        """beam_search over a batch of sequences given as concatenated rows (radian/decode.py:100-212).
        Returns a list of uint8 label arrays (and the winners' log pr_total when with_scores)."""
        mats = np.ascontiguousarray(mats)
        if mats.dtype not in (np.float32, np.float64):
            raise TypeError("probabilities must be float32 or float64")
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        n = int(seq_len.shape[0])
        label_off = np.zeros(n, dtype=np.int64)
        if n:
            label_off[1:] = np.cumsum(seq_len[:-1].astype(np.int64))
        labels = np.zeros(int(seq_len.astype(np.int64).sum()) + 1, dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        scores = np.zeros(n, dtype=np.float64) if with_scores else None
        self._check(self._L.rd_decode_batch(self._h, _p(mats), 1 if mats.dtype == np.float64 else 0, _p(seq_off), _p(seq_len), n,
                                            int(beam_width), 1 if use_lm else 0, float(s_threshold), float(r_threshold),
                                            _p(labels), _p(label_off), _p(lens), _p(scores)))
        out = [_labels_of(labels, label_off[i], lens[i]) for i in range(n)]
        return (out, scores) if with_scores else out

    def decode(self, mat, beam_width, use_lm=False, s_threshold=0.0, r_threshold=0.0):
        mat = np.ascontiguousarray(mat)
        return self.decode_batch(mat.reshape(-1, 5), [0], [mat.shape[0]], beam_width, use_lm, s_threshold, r_threshold)[0]

    # ------------------------------------------------------------------ fused paths
    def basecall_chunk(self, windows, valid_len, beam_width):
        """forward + LM-free per-window beam search (radian/basecall.py:86-96,110-121)."""
        windows = np.ascontiguousarray(windows, dtype=np.float32)
        n, T = windows.shape
        valid_len = np.ascontiguousarray(valid_len, dtype=np.int32)
        labels = np.zeros((n, T), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        self._check(self._L.rd_basecall_chunk(self._h, _p(windows), n, T, _p(valid_len), int(beam_width), _p(labels), _p(lens)))
        return [labels[i, : lens[i]].copy() for i in range(n)]

    def basecall_global(self, windows, read_win_off, pads, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        """forward + per-read assembly + one beam search per read (radian/basecall.py:86-109)."""
        windows = np.ascontiguousarray(windows, dtype=np.float32)
        nW, T = windows.shape
        read_win_off = np.ascontiguousarray(read_win_off, dtype=np.int32)
        pads = np.ascontiguousarray(pads, dtype=np.int32)
        n_reads = pads.shape[0]
        caps = np.array([(read_win_off[r + 1] - read_win_off[r] - 1) * step + T - pads[r] for r in range(n_reads)], dtype=np.int64)
        caps = np.maximum(caps, 0)
        label_off = np.zeros(n_reads, dtype=np.int64)
        label_off[1:] = np.cumsum(caps[:-1])
        labels = np.zeros(int(caps.sum()) + 1, dtype=np.uint8)
        lens = np.zeros(n_reads, dtype=np.int32)
        self._check(self._L.rd_basecall_global(self._h, _p(windows), T, int(step), _p(read_win_off), _p(pads), n_reads,
                                               int(beam_width), 1 if use_lm else 0, float(s_threshold), float(r_threshold),
                                               _p(labels), _p(label_off), _p(lens)))
        return [_labels_of(labels, label_off[r], lens[r]) for r in range(n_reads)]

    # ------------------------------------------------------------------ reads-level fused paths
    def count_windows(self, n_samples, chunk_len, step):
        n = self._L.rd_count_windows(int(n_samples), int(chunk_len), int(step))
        if n < 0:
            raise ValueError("bad window geometry")
        return n

    @staticmethod
    def _pack_reads(signals):
        sig = [np.ascontiguousarray(x, dtype=np.float32).ravel() for x in signals]
        off = np.zeros(len(sig) + 1, dtype=np.int64)
        off[1:] = np.cumsum([x.shape[0] for x in sig])
        return (np.concatenate(sig) if sig else np.zeros(0, np.float32)), off

    def basecall_reads_chunk(self, signals, chunk_len, step, beam_width):
        """Chunk mode over whole (normalised) reads: returns, per read, the list of per-window label arrays
        (radian/basecall.py:83-121 without the host stitch).  Each time step is computed once (streamed forward)."""
        flat, off = self._pack_reads(signals)
        nw = [self.count_windows(off[r + 1] - off[r], chunk_len, step) for r in range(len(signals))]
        tot = int(sum(nw))
        labels = np.zeros((tot, chunk_len), dtype=np.uint8)
        lens = np.zeros(tot, dtype=np.int32)
        self._check(self._L.rd_basecall_reads_chunk(self._h, _p(flat), _p(off), len(signals), int(chunk_len), int(step),
                                                    int(beam_width), _p(labels), _p(lens)))
        out, w = [], 0
        for n in nw:
            out.append([labels[w + i, : lens[w + i]].copy() for i in range(n)])
            w += n
        return out

    def basecall_reads_global(self, signals, chunk_len, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        """Global mode over whole (normalised) reads (radian/basecall.py:83-109): one label array per read."""
        flat, off = self._pack_reads(signals)
        n = len(signals)
        labels = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        label_off = np.ascontiguousarray(off[:-1])
        self._check(self._L.rd_basecall_reads_global(self._h, _p(flat), _p(off), n, int(chunk_len), int(step), int(beam_width),
                                                     1 if use_lm else 0, float(s_threshold), float(r_threshold), _p(labels),
                                                     _p(label_off), _p(lens)))
        return [_labels_of(labels, off[r], lens[r]) for r in range(n)]

    # ------------------------------------------------------------------ raw int16 reads (normalisation on the device)
    STATUS_MESSAGES = {1: "MAD is zero, issue with signal.", 2: "Signal must not be empty to normalise"}  # preprocess.py:25-26,47-48

    @staticmethod
    def _pack_raw(raws):
        sig = [np.ascontiguousarray(x, dtype=np.int16).ravel() for x in raws]
        off = np.zeros(len(sig) + 1, dtype=np.int64)
        off[1:] = np.cumsum([x.shape[0] for x in sig])
        flat = np.concatenate(sig) if sig else np.zeros(0, np.int16)
        if flat.size == 0:
            flat = np.zeros(1, np.int16)
        return flat, off

    def normalise_reads(self, raws, outlier_clip):
        """float32(mad_normalise(raw, clip)) per read (radian/preprocess.py:24-49) + status per read (0 ok, 1 MAD zero, 2 empty)."""
        flat, off = self._pack_raw(raws)
        out = np.zeros(max(1, int(off[-1])), dtype=np.float32)
        status = np.zeros(len(raws), dtype=np.int32)
        self._check(self._L.rd_normalise_reads(self._h, _p(flat), _p(off), len(raws), int(outlier_clip), _p(out), _p(status)))
        return [out[off[r]:off[r + 1]].copy() for r in range(len(raws))], status

    def basecall_raw_chunk(self, raws, outlier_clip, chunk_len, step, beam_width):
        """raw int16 reads -> (per read list of per-window label arrays, status per read)."""
        flat, off = self._pack_raw(raws)
        nw = [self.count_windows(off[r + 1] - off[r], chunk_len, step) for r in range(len(raws))]
        tot = int(sum(nw))
        labels = np.zeros((tot, chunk_len), dtype=np.uint8)
        lens = np.zeros(tot, dtype=np.int32)
        status = np.zeros(len(raws), dtype=np.int32)
        self._check(self._L.rd_basecall_raw_chunk(self._h, _p(flat), _p(off), len(raws), int(outlier_clip), int(chunk_len), int(step),
                                                  int(beam_width), _p(labels), _p(lens), _p(status)))
        out, w = [], 0
        for n in nw:
            out.append([labels[w + i, : lens[w + i]].copy() for i in range(n)])
            w += n
        return out, status

    def basecall_raw_global(self, raws, outlier_clip, chunk_len, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        flat, off = self._pack_raw(raws)
        n = len(raws)
        labels = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        label_off = np.ascontiguousarray(off[:-1])
        self._check(self._L.rd_basecall_raw_global(self._h, _p(flat), _p(off), n, int(outlier_clip), int(chunk_len), int(step),
                                                   int(beam_width), 1 if use_lm else 0, float(s_threshold), float(r_threshold),
                                                   _p(labels), _p(label_off), _p(lens), _p(status)))
        return [_labels_of(labels, off[r], lens[r]) for r in range(n)], status

    # ------------------------------------------------------------------ device-resident (bench)
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._L.rd_dev_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return p

    def mem_info(self):
        """(free, total) bytes of the context's device"""
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self._L.rd_mem_info(self._h, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def dev_free(self, p):
        self._check(self._L.rd_dev_free(self._h, p))

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self._L.rd_memcpy_h2d(self._h, d_ptr, _p(arr), arr.nbytes))

    def d2h(self, arr, d_ptr):
        self._check(self._L.rd_memcpy_d2h(self._h, _p(arr), d_ptr, arr.nbytes))

    def forward_resident(self, d_windows, n, T, d_probs=None):
        self._check(self._L.rd_forward_resident(self._h, d_windows, n, T, d_probs))

    def forward_reads_resident(self, d_signal, read_off, n_reads, chunk_len, step, decode_type="chunk", lane=0):
        """forward only over whole reads resident in HBM (rd_forward_reads_resident): asynchronous on `lane`; -> rows evaluated"""
        rows = ctypes.c_int64(0)
        self._check(self._L.rd_forward_reads_resident(self._h, d_signal, _p(read_off), int(n_reads), int(chunk_len), int(step),
                                                      {"chunk": 0, "global": 1}[decode_type], int(lane), ctypes.byref(rows)))
        return rows.value

    def basecall_chunk_resident(self, d_windows, n, T, valid_len, beam_width, labels, lens):
        self._check(self._L.rd_basecall_chunk_resident(self._h, d_windows, n, T, _p(valid_len), int(beam_width), _p(labels), _p(lens)))

    def decode_resident(self, d_probs, n, T, valid_len, beam_width, labels, lens):
        self._check(self._L.rd_decode_resident(self._h, d_probs, n, T, _p(valid_len), int(beam_width), _p(labels), _p(lens)))

    def pipe_submit(self, d_windows, n, T, valid_len, beam_width, labels, lens):
        """Two-stream pipeline (rd_pipe_submit): labels/lens are filled two submits later or at pipe_flush()."""
        self._check(self._L.rd_pipe_submit(self._h, d_windows, n, T, _p(valid_len), int(beam_width), _p(labels), _p(lens)))

    def pipe_submit_reads(self, d_signal, read_off, n_reads, chunk_len, step, beam_width, labels, lens):
        """Pipelined chunk-mode batch of whole reads resident in HBM (rd_pipe_submit_reads)."""
        self._check(self._L.rd_pipe_submit_reads(self._h, d_signal, _p(read_off), int(n_reads), int(chunk_len), int(step),
                                                 int(beam_width), _p(labels), _p(lens)))

    def basecall_reads_chunk_resident(self, d_signal, read_off, n_reads, chunk_len, step, beam_width, labels, lens):
        self._check(self._L.rd_basecall_reads_chunk_resident(self._h, d_signal, _p(read_off), int(n_reads), int(chunk_len),
                                                             int(step), int(beam_width), _p(labels), _p(lens)))

    def basecall_reads_global_resident(self, d_signal, read_off, n_reads, chunk_len, step, beam_width, use_lm, s_threshold,
                                       r_threshold, labels, label_off, lens):
        """Global mode over normalised reads resident in HBM (rd_basecall_reads_global_resident)."""
        self._check(self._L.rd_basecall_reads_global_resident(self._h, d_signal, _p(read_off), int(n_reads), int(chunk_len),
                                                              int(step), int(beam_width), 1 if use_lm else 0, float(s_threshold),
                                                              float(r_threshold), _p(labels), _p(label_off), _p(lens)))

    # ------------------------------------------------------------------ reads-level pipeline (one context, pipe_reads.hip)
    def pipe_submit_reads_global(self, d_signal, read_off, n_reads, chunk_len, step, beam_width, use_lm, s_threshold, r_threshold,
                                 labels, label_off, lens):
        """Pipelined global-mode batch of normalised reads resident in HBM (rd_pipe_submit_reads_global); labels / lens are
        filled once pipe_progress reports the batch delivered (or at pipe_flush)."""
        self._check(self._L.rd_pipe_submit_reads_global(self._h, d_signal, _p(read_off), int(n_reads), int(chunk_len), int(step),
                                                        int(beam_width), 1 if use_lm else 0, float(s_threshold), float(r_threshold),
                                                        _p(labels), _p(label_off), _p(lens)))

    def pipe_progress(self, wait_for=0):
        """Deliver finished groups of the reads-level pipeline; blocks until `wait_for` submits (counted over the context's
        life) are delivered when wait_for > 0.  -> number of submits delivered so far."""
        n = ctypes.c_int64(0)
        self._check(self._L.rd_pipe_progress(self._h, int(wait_for), ctypes.byref(n)))
        self._release_tickets(n.value)
        return n.value

    def _release_tickets(self, delivered):
        for seq in [q for q in self._tickets if q <= delivered]:
            del self._tickets[seq]

    def pipe_submit_raw(self, decode_type, raws, outlier_clip, chunk_len, step, beam_width, use_lm=False, s_threshold=0.0,
                        r_threshold=0.0):
        """Pipelined form of basecall_raw_global / basecall_raw_chunk: queue a batch of raw int16 reads and return a
        PipeTicket; ticket.result() gives what the unpipelined call returns once the batch is delivered."""
        flat, off = self._pack_raw(raws)
        n = len(raws)
        t = PipeTicket(self, decode_type, off, n)
        if decode_type == "global":
            t.labels = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
            t.lens = np.zeros(n, dtype=np.int32)
            t.label_off = np.ascontiguousarray(off[:-1])
            self._check(self._L.rd_pipe_submit_raw_global(self._h, _p(flat), _p(off), n, int(outlier_clip), int(chunk_len), int(step),
                                                          int(beam_width), 1 if use_lm else 0, float(s_threshold), float(r_threshold),
                                                          _p(t.labels), _p(t.label_off), _p(t.lens), _p(t.status)))
        else:
            t.nw = [self.count_windows(off[r + 1] - off[r], chunk_len, step) for r in range(n)]
            tot = int(sum(t.nw))
            t.labels = np.zeros((tot, chunk_len), dtype=np.uint8)
            t.lens = np.zeros(tot, dtype=np.int32)
            self._check(self._L.rd_pipe_submit_raw_chunk(self._h, _p(flat), _p(off), n, int(outlier_clip), int(chunk_len), int(step),
                                                         int(beam_width), _p(t.labels), _p(t.lens), _p(t.status)))
        t.seq = self.pipe_submitted()
        self._tickets[t.seq] = t
        return t

    def pipe_submitted(self):
        """batches submitted to the reads-level pipeline so far = the pipe_progress count at which the latest one is delivered"""
        n = ctypes.c_int64(0)
        self._check(self._L.rd_pipe_submitted(self._h, ctypes.byref(n)))
        return n.value

    def pipe_stats(self):
        """counters of the reads-level pipeline (rd_pipe_stats)"""
        v = (ctypes.c_int64 * 5)()
        self._check(self._L.rd_pipe_stats(self._h, v, 5))
        return dict(zip(("submitted", "delivered", "launches", "queue_launches", "limit_closes"), (int(x) for x in v)))

    def pipe_policy(self, beam_width, on_partition, use_lm=False):
        """what the context has measured for its global-mode group policy (rd_pipe_policy_read): ns per forward row, us per time
        step of a group's longest chain (0.0: not measured yet) and the rule in force, forward rows per chain step.  on_partition:
        1..3 = waves per SIMD of the decode partition, 0 = the whole chip"""
        ns, us, rows = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        self._check(self._L.rd_pipe_policy_read(self._h, int(beam_width), int(on_partition), 1 if use_lm else 0, ctypes.byref(ns),
                                                ctypes.byref(us), ctypes.byref(rows)))
        return {"ns_per_row": ns.value, "us_per_step": us.value, "rows_per_step": rows.value}

    def pipe_config(self, group_batches):
        self._check(self._L.rd_pipe_config(self._h, int(group_batches)))

    def pipe_set_lanes(self, lanes):
        """forward streams the submitted batches rotate over (1..4, default 2)"""
        self._check(self._L.rd_pipe_set_lanes(self._h, int(lanes)))

    def pipe_flush(self):
        self._check(self._L.rd_pipe_flush(self._h))
        self._release_tickets(self.pipe_submitted())

    def timer_enable(self, which, max_launches):
        self._check(self._L.rd_timer_enable(self._h, which, max_launches))

    def timer_read(self, which):
        ms = ctypes.c_double()
        n = ctypes.c_int()
        fl = ctypes.c_double()
        by = ctypes.c_double()
        self._check(self._L.rd_timer_read(self._h, which, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
        return {"total_ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}

    def timer_read_launches(self, which, cap=4096):
        """the recorded launches one by one (rd_timer_read_launches): (ms float32[n], flops float64[n], tag int32[n])"""
        ms = np.zeros(cap, dtype=np.float32)
        fl = np.zeros(cap, dtype=np.float64)
        tag = np.zeros(cap, dtype=np.int32)
        n = ctypes.c_int()
        self._check(self._L.rd_timer_read_launches(self._h, which, int(cap), _p(ms), _p(fl), _p(tag), ctypes.byref(n)))
        return ms[: n.value], fl[: n.value], tag[: n.value]

    # ------------------------------------------------------------------ multi-GPU start-up
    def rccl_probe(self):
        """librccl loads in this process (no communicator, no bootstrap thread)."""
        self._check(self._L.rd_rccl_probe())

    def rccl_unique_id(self):
        buf = (ctypes.c_uint8 * 128)()
        self._check(self._L.rd_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def rccl_init(self, rank, nranks, uid):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self._L.rd_rccl_init(self._h, rank, nranks, ctypes.cast(buf, ctypes.c_void_p)))

    def rccl_finalize(self):
        self._check(self._L.rd_rccl_finalize(self._h))

    def clone_artifacts_from(self, src):
        """Take the loaded weights (every packing) and LM table of another Backend of this process by a device copy
        (rd_clone_artifacts): the receiver's code of the multi-GPU broadcast, without parsing / repacking again."""
        self._check(self._L.rd_clone_artifacts(self._h, src._h))

    def rccl_bcast_model(self, root=0):
        self._check(self._L.rd_rccl_bcast_model(self._h, root))

    def rccl_allreduce_max(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._L.rd_rccl_allreduce_max(self._h, _p(a), a.size))
        return a

    def rccl_barrier(self):
        self._check(self._L.rd_rccl_barrier(self._h))

    def rccl_comm_count(self):
        """ranks in the communicator as RCCL reports them (ncclCommCount)"""
        n = ctypes.c_int(0)
        self._check(self._L.rd_rccl_comm_count(self._h, ctypes.byref(n)))
        return n.value
