/*
 * radian_hip.h -- C ABI of libradian_hip.so: the MI355X (gfx950) backend for RADIAN's
 * inference + decode hot path.
 *
 * The reference (comprna/radian) has no FFI; the seam is five in-process Python calls made by
 * radian/basecall.py.  Each entry point below names the reference call it replaces (file:line under
 * the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds to basecall.py.
 * This header lists what a RADIAN maintainer binds and nothing else: the measurement switches, kernel timers and pipeline
 * read-outs that bench.py, tools/ and tests/ use live in radian_hip_diag.h (same library; none of them changes a result).
 *
 * Conventions
 *   - every function returns 0 on success or a negative RD_ERR_* code; rd_last_error() then returns
 *     a thread-local human-readable message.  Nothing is thrown, nothing calls back into the host.
 *   - all pointers are caller-owned HOST memory unless the parameter name starts with d_ (device
 *     memory obtained from rd_dev_alloc).  Plain pointers and sizes only; no framework types.
 *   - one rd_ctx per process and per GPU rank; calls on one context are not thread-safe.
 *   - class order of probability rows is A, C, G, T, blank (radian/models/sig2seq.yaml:2;
 *     radian/decode.py:124); labels are 0..3 = A,C,G,T and are NOT reversed (basecall.py:129
 *     reverses the string on the host).
 *   - there is no CPU fallback: without a GPU rd_create fails.
 */
#ifndef RADIAN_HIP_H
#define RADIAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_ERR_ARG (-1)   /* bad argument */
#define RD_ERR_HIP (-2)   /* HIP runtime error */
#define RD_ERR_STATE (-3) /* call order (e.g. forward before weights) */
#define RD_ERR_NOMEM (-4) /* device allocation failed */
#define RD_ERR_RCCL (-5)  /* RCCL error / librccl not loadable */
#define RD_ERR_FORMAT (-6) /* rd_lm_json_* / rd_fast5_*: the input is not of the one shape the fast reader handles (no verdict: use the full parser / libhdf5) */
#define RD_ERR_IO (-7)     /* rd_fast5_open: the file cannot be opened or mapped */
/* Not an error code: the value of label_len[i] for a sequence whose beam search looked up a context that a SPARSE RNA model
 * does not hold (rd_load_lm, rows of NaN).  The reference raises KeyError at radian/decode.py:83 on such a read; the caller
 * does the same when it reaches that read (radian_amd/basecall.py).  The sequence's labels are not written. */
#define RD_LEN_MISSING_CONTEXT (-1)

typedef struct rd_ctx rd_ctx;

/* ---- library ------------------------------------------------------------------------------- */
const char* rd_last_error(void);
int rd_version(void);                 /* ABI version, currently 1 */
int rd_device_count(int* n);          /* number of visible HIP devices */
int rd_decode_max_width(void);        /* largest supported --beam-width (1024; radian/decode.py:145 slices with any width) */
int rd_decode_lane_width(void);       /* widths up to this (256) run on the wave-per-sequence kernels, wider ones on the general kernel */

/* ---- context ------------------------------------------------------------------------------- */
int rd_create(int device_id, rd_ctx** out);
int rd_destroy(rd_ctx* ctx);
int rd_sync(rd_ctx* ctx);             /* wait for the context's stream */
/* Matrix-product arithmetic of the forward (sig_model.predict, radian/basecall.py:88-93; the reference computes in fp32):
 *   0 (default) exact fp32 MFMA;
 *   1 split-f16 "f16x3": every fp32 operand carried as an f16 hi+lo pair (22 significant bits), products
 *     hi*hi + hi*lo + lo*hi accumulated in fp32 on the f16 matrix pipe;
 *   2 three-term bf16 split "bf16x3": every fp32 operand carried EXACTLY as hi+mid+lo bf16 (3 x 8 significant bits,
 *     fp32's exponent range), the six cross products down to 2^-16 relative accumulated in fp32 on the bf16 matrix pipe;
 *     the dropped terms are below one fp32 rounding of the product. */
int rd_set_precision(rd_ctx* ctx, int mode);

/* ---- model artefacts ----------------------------------------------------------------------- */
/* Replaces model.load_weights(checkpoint) -- radian/model.py:42-45.
 * blob = rd_weights_header followed by float32 tensors in Keras load_weights order and layouts:
 *   block0: conv0.kernel[K][1][C], conv0.bias[C], conv1.kernel[K][C][C], conv1.bias[C],
 *           matching.kernel[1][1][C], matching.bias[C];
 *   block i>0: conv0.kernel[K][C][C], bias[C], conv1.kernel[K][C][C], bias[C];
 *   dense.kernel[C][H], dense.bias[H]; dense_1.kernel[H][5], dense_1.bias[5].
 * Geometry is fixed to radian/models/sig2seq.yaml:34-49 (C=256, K=3, H=128, 5 classes); the number
 * of blocks and their dilations come from the header. */
typedef struct rd_weights_header {
    uint32_t magic;      /* 'RDNW' = 0x574e4452 */
    uint32_t version;    /* 1 */
    uint32_t nb_filters; /* 256 */
    uint32_t kernel_size;/* 3 */
    uint32_t relu_units; /* 128 */
    uint32_t n_classes;  /* 5 */
    uint32_t n_blocks;   /* len(dilations) * nb_stacks, <= 16 */
    uint32_t dilations[16];
    uint32_t n_floats;   /* number of float32 values that follow */
} rd_weights_header;
int rd_load_weights(rd_ctx* ctx, const void* blob, size_t nbytes);

/* Replaces the RNA-model dict built at radian/basecall.py:48-57 and read at decode.py:83.
 * table[ctx][4] doubles, ctx = base-4 number of the k context labels, oldest label most
 * significant (the JSON key string read left to right).  k = --context-len, 1..13.
 * A row of NaNs marks a context the model does NOT hold (a sparse JSON): the reference's dict lookup raises KeyError
 * when -- and only when -- the search of a read keeps a labeling that ends in such a context (decode.py:83, looked up
 * for every kept labeling of >= k labels at every time step, whatever the gate says); here that read's label_len comes
 * back as RD_LEN_MISSING_CONTEXT and every other read is unaffected.
 * Passing table == NULL unloads the LM. */
int rd_load_lm(rd_ctx* ctx, const double* table, int k);
/* --context-len k (1..13) with an RNA model whose keys have another length: the reference's `model[context]` (decode.py:83) then
 * raises KeyError for EVERY context, i.e. on the first read whose search keeps a labeling of >= k labels before its last time step,
 * after having basecalled the shorter reads before it (basecall.py:70-141).  Loads that model: every read that reaches such a
 * labeling comes back as RD_LEN_MISSING_CONTEXT, every other read decodes as without an LM (no lookup ever succeeds). */
int rd_load_lm_absent(rd_ctx* ctx, int k);
/* Long contexts (--context-len up to 256; BASELINE configs[4]).  NO reference behaviour: the reference needs one dict
 * entry per context (decode.py:83), impossible beyond a dozen labels.  A synthetic LM for such contexts is a dense
 * table[4^table_order][4] addressed by a hash of the context: row = H(l_0..l_{k-1}) & (4^table_order - 1),
 * H = sum l_i * B^(k-1-i) mod 2^32, B = 0x9E3779B1; gate and mixing as decode.py:79-96.  The decoder keeps the hash
 * incrementally per beam with a 256-label ring (the label leaving the window).  Parity: the oracle's same definition. */
int rd_load_lm_hashed(rd_ctx* ctx, const double* table, int table_order, int context_len);
/* Storage type of the softmax rows between the head kernel and the decoder on the reads-level paths
 * (rd_basecall_reads_*, rd_basecall_raw_*, rd_pipe_submit_reads): 0 float32 (the reference's, default), 1 float16
 * (10 B per time step; rounded to nearest by the head kernel, widened exactly by the decoder / assembly).  Not a
 * reference option (BASELINE configs[4] "fp16 logits"): labels equal the oracle's on the same f16-rounded rows. */
int rd_set_logits(rd_ctx* ctx, int mode);
/* Decode partition of the global-mode reads pipeline (rd_pipe_submit_reads_global / rd_pipe_submit_raw_global; no effect on
 * results, no reference counterpart): cus_per_xcd CUs of each of the 8 XCDs are kept free of forward workgroups (the
 * pipeline's forward streams are CU-masked to the others) and run the beam search.  A read's search is one serial chain of
 * a time step per sample; a beam-search wave that shares its SIMD with conv waves issuing MFMAs back to back gets about one
 * instruction issue per MFMA (17 us per step measured instead of 2).  -1 (default) = by beam width (4 CUs per XCD, 8 above
 * W = 25, 12 above W = 64, 16 above W = 128: 12.5 ... 50 % of the chip; a masked queue's CUs are dealt over the four shader engines of an XCD and the forward
 * runs at the pace of the engine left with the fewest, so only multiples of four are worth setting), 0 = off (groups then
 * grow until their forward rows cover the slow chain). */
int rd_set_decode_partition(rd_ctx* ctx, int cus_per_xcd);
/* Arithmetic of the beam search's log / logaddexp (decode.py:16-17,172-201 call math.log and np.logaddexp, i.e. the host's
 * libm): 1 (default) = the operation sequence of glibc 2.35's x86-64 FMA build (exp, log, log1p restated in
 * csrc/glibc_math.h): scores and labelings bit-identical to the reference's on such a host, including labelings that are
 * equiprobable in exact arithmetic; 0 = this library's faster routines (<= 1 ulp from glibc's, so scores agree to a few ulp
 * and labelings are identical unless two labelings tie within that distance; the beam search runs 10 % (peaked rows,
 * thousands of sequences) to 35 % (flat rows, few sequences) faster). */
int rd_set_decode_math(rd_ctx* ctx, int mode);

/* ---- the five seams, host-pointer form ----------------------------------------------------- */
/* sig_model.predict(windows) -- radian/basecall.py:91,93.
 * windows [n_windows][chunk_len] float32 (MAD-normalised) -> probs [n_windows][chunk_len][5] float32. */
int rd_forward(rd_ctx* ctx, const float* windows, int n_windows, int chunk_len, float* probs);

/* assemble_matrices(matrices, step_size) after matrices[-1] = matrices[-1][:-pad]
 * -- radian/basecall.py:96,100; radian/matrix_assembly.py:6-53.
 * probs [n_windows][chunk_len][5] float32 of ONE read; out receives [*n_rows][5] float64
 * (capacity out_cap rows); *is_f64 reports the reference's result dtype (float64 when any time step
 * is covered by more than one window, else float32 -- the values are exact either way). */
int rd_assemble(rd_ctx* ctx, const float* probs, int n_windows, int chunk_len, int pad, int step, double* out,
                int64_t out_cap, int64_t* n_rows, int* is_f64);

/* beam_search(mat, 'ACGT', beam_width, lm, s_threshold, r_threshold, len_context, cache)
 * -- radian/basecall.py:102-109 (global) and :113-120 (chunk); radian/decode.py:100-212.
 * A batch of independent sequences over concatenated probability rows:
 *   probs      rows [*][5], float32 (prob_is_f64=0) or float64 (1)
 *   seq_off[i] first row of sequence i, seq_len[i] its number of rows (0 allowed)
 *   use_lm     0: lm=None (chunk mode); 1: use the table from rd_load_lm with thresholds s_thr/r_thr
 *   labels_out receives sequence i's labels at labels_out + label_off[i] (capacity >= seq_len[i]),
 *   label_len[i] its length; best_score (nullable) the winner's log pr_total. */
int rd_decode_batch(rd_ctx* ctx, const void* probs, int prob_is_f64, const int64_t* seq_off, const int32_t* seq_len,
                    int n_seq, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                    const int64_t* label_off, int32_t* label_len, double* best_score);

/* ---- fused paths (probabilities never leave HBM) ------------------------------------------- */
/* chunk mode, radian/basecall.py:86-96,110-121 for a batch of windows that may span many reads:
 * forward, then an LM-free beam search of each window over its first valid_len[i] rows
 * (chunk_len, or chunk_len - pad for the last window of a read).  Labels of window i are written
 * at labels_out + i*chunk_len.  simple_assembly (basecall.py:122-123) stays on the host. */
int rd_basecall_chunk(rd_ctx* ctx, const float* windows, int n_windows, int chunk_len, const int32_t* valid_len,
                      int beam_width, uint8_t* labels_out, int32_t* label_len);

/* global mode, radian/basecall.py:86-109 for a batch of reads: forward over all windows, per-read
 * assembly, one LM-gated beam search per read.
 *   read_win_off[r] first window of read r (n_reads+1 entries), pad[r] the zero padding of its last
 *   window, step the window step; labels of read r at labels_out + label_off[r]
 *   (capacity >= assembled length = (nW_r-1)*step + chunk_len - pad[r]). */
int rd_basecall_global(rd_ctx* ctx, const float* windows, int chunk_len, int step, const int32_t* read_win_off,
                       const int32_t* pad, int n_reads, int beam_width, int use_lm, double s_thr, double r_thr,
                       uint8_t* labels_out, const int64_t* label_off, int32_t* label_len);

/* ---- reads-level fused paths: windowing happens on the device and each time step is computed once ------------
 * The loop body of radian/basecall.py:83-121 for a batch of whole reads.  signal = the reads' MAD-normalised samples
 * (basecall.py:78) packed back to back, read r at [read_off[r], read_off[r+1]) (n_reads+1 offsets, read_off[0] = 0).
 * Windows are those of get_windows(signal, chunk_len, step) (preprocess.py:4-22).  The causal TCN is evaluated once
 * over each read ("stream"); a window's rows past the receptive field are the stream's rows and only its first
 * RF-1 rows are computed separately, so the probabilities -- and the labels -- are bit-identical to the windowed
 * computation at a fraction of the work.
 *   chunk : labels of window w (counted across the batch, rd_count_windows per read) at labels_out + w*chunk_len,
 *           label_len[w]; simple_assembly stays on the host.
 *   global: labels of read r at labels_out + label_off[r] (capacity >= its number of samples), label_len[r]. */
int rd_count_windows(int64_t n_samples, int chunk_len, int step);   /* windows get_windows makes of one read */
int rd_basecall_reads_chunk(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len,
                            int step, int beam_width, uint8_t* labels_out, int32_t* label_len);
int rd_basecall_reads_global(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len,
                             int step, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                             const int64_t* label_off, int32_t* label_len);

/* ---- the step before the hot path on the device: mad_normalise -- radian/preprocess.py:24-49, basecall.py:78 ----
 * raw = unscaled int16 DAQ samples of the reads packed back to back (read r at [read_off[r], read_off[r+1])).
 * status[r]: 0 ok; 1 "MAD is zero, issue with signal." (output zeros); 2 "Signal must not be empty to normalise" --
 * the two ValueErrors after which basecall.py:77-82 skips the read.  Output = float32(mad_normalise(raw, clip)),
 * bit-identical to NumPy (exact order statistics, IEEE float64 arithmetic, np.vectorize's int64 quirk included). */
int rd_normalise_reads(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                       float* norm_out /* nullable */, int32_t* status);
/* normalise + reads-level chunk / global basecall in one call; reads with status 1 are computed on zeros and must be
 * dropped by the caller, empty reads are rejected (filter them first). */
int rd_basecall_raw_chunk(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                          int chunk_len, int step, int beam_width, uint8_t* labels_out, int32_t* label_len, int32_t* status);
int rd_basecall_raw_global(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                           int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr,
                           uint8_t* labels_out, const int64_t* label_off, int32_t* label_len, int32_t* status);

/* ---- device-resident form (inputs already in HBM; used by bench.py and by pipelined hosts) -- */
int rd_dev_alloc(rd_ctx* ctx, size_t bytes, void** d_ptr);
int rd_dev_free(rd_ctx* ctx, void* d_ptr);
int rd_mem_info(rd_ctx* ctx, size_t* free_bytes, size_t* total_bytes);   /* hipMemGetInfo of the context's device (batch sizing) */
int rd_memcpy_h2d(rd_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int rd_memcpy_d2h(rd_ctx* ctx, void* dst, const void* d_src, size_t bytes);
/* same contract as rd_forward / rd_basecall_chunk with d_windows resident; d_probs may be NULL
 * (internal workspace).  Asynchronous on the context stream except for the label copy-out. */
int rd_forward_resident(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len, float* d_probs);
/* sig_model.predict alone (radian/basecall.py:88-93; BASELINE configs[1] "forward only") for a batch of whole normalised
 * reads resident in HBM: the streamed evaluation of rd_basecall_reads_chunk (decode_type 0: one row per time step + the
 * rows of every window's zero-padded head) or rd_basecall_reads_global (1) without the decode, asynchronous on forward
 * lane `lane` (0..3; lanes are independent streams with their own activations).  Rows go to the context's probability
 * workspace -- ONE workspace per context: calls on several lanes overwrite each other's rows (the entry point exists for the
 * forward's timing, BASELINE configs[1]; rd_forward_reads returns rows); *total_rows (nullable) = their number.  rd_sync
 * waits for every lane. */
int rd_forward_reads_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len,
                              int step, int decode_type, int lane, int64_t* total_rows);
/* The same forward with the reference's shapes: `signal` holds the normalised reads back to back (host), probs_out receives
 * [*n_windows][chunk_len][5] float32 -- what `sig_model.predict(windows)` returns for get_windows() of every read in turn
 * (radian/basecall.py:83-93), window rows the pad trim at basecall.py:96 drops are zero.  Every time step is evaluated once
 * (plus each later window's first 252 rows, which see that window's own zero padding) and the rows are gathered into
 * windows on the device: bit-identical to rd_forward on the same windows.  Blocking. */
int rd_forward_reads(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len, int step,
                     float* probs_out, int64_t windows_cap, int64_t* n_windows);
int rd_basecall_chunk_resident(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len,
                               const int32_t* valid_len, int beam_width, uint8_t* labels_out, int32_t* label_len);
int rd_decode_resident(rd_ctx* ctx, const float* d_probs, int n_windows, int chunk_len, const int32_t* valid_len,
                       int beam_width, uint8_t* labels_out, int32_t* label_len);

/* Software pipeline over chunk-mode batches: the forwards of consecutive submitted batches rotate over a few
 * independent streams ("lanes", default 2, rd_pipe_set_lanes; each lane has its own activation tensors) so that one
 * batch's partially filled last round of workgroups overlaps the next batch's launches; the beam search + label
 * copy-out of a GROUP of submitted batches (default 4, rd_pipe_config) runs on a further, high-priority stream and
 * overlaps the forwards of the next group.  Same contract as rd_basecall_chunk_resident, except that
 * labels_out/label_len of a submitted batch are only valid after rd_pipe_flush (or once a later submit had to
 * recycle its slot); the caller keeps them alive until then (labels of window i at labels_out + i*chunk_len).
 * The input buffers of a submitted batch must stay untouched until then as well. */
int rd_pipe_config(rd_ctx* ctx, int group_batches);
int rd_pipe_set_lanes(rd_ctx* ctx, int lanes); /* 1..4; only while the pipeline is empty */
int rd_pipe_submit(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len, const int32_t* valid_len,
                   int beam_width, uint8_t* labels_out, int32_t* label_len);
int rd_pipe_flush(rd_ctx* ctx);
/* reads-level forms with the signal resident in HBM */
int rd_basecall_reads_chunk_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads,
                                     int chunk_len, int step, int beam_width, uint8_t* labels_out, int32_t* label_len);
int rd_basecall_reads_global_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads,
                                      int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr,
                                      uint8_t* labels_out, const int64_t* label_off, int32_t* label_len);
int rd_pipe_submit_reads(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len,
                         int step, int beam_width, uint8_t* labels_out, int32_t* label_len);

/* The same pipeline over batches of WHOLE READS, for both decode types and for raw input -- the loop body of
 * radian/basecall.py:77-121 (mad_normalise, get_windows, predict, assemble_matrices, beam_search) for a stream of read
 * batches inside ONE context: on the next forward lane [raw form: H2D of the samples out of the lane's pinned staging block,
 * MAD normalisation] -> streamed forward -> [global: per-read assembly]; on the decode stream, per GROUP of batches, the
 * beam search of every read (global; LM-gated with use_lm) or window (chunk) + labels to the host.  A group closes after
 * rd_pipe_config batches, or -- global mode -- as soon as its forward rows cover the beam search of its longest read (a
 * read's search is one serial chain of a time step per sample, which only the next group's forwards can hide).
 * Contracts as rd_basecall_reads_global / rd_basecall_raw_global / rd_basecall_raw_chunk, except:
 *   - labels_out / label_len / status of a submitted batch are valid once rd_pipe_progress reports it delivered (or after
 *     rd_pipe_flush); the caller keeps them alive until then.  raw / read_off / label_off may be reused when the call returns;
 *     d_signal must stay untouched until delivery;
 *   - empty reads are rejected (RD_ERR_ARG): basecall.py:77-82 skips them before this point.
 * Batches whose read lengths equal the previous batch's on the same lane reuse its tile descriptors; otherwise the plan
 * is rebuilt on the host and uploaded behind the lane's previous forward -- no stream is drained for it. */
int rd_pipe_submit_reads_global(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len,
                                int step, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                                const int64_t* label_off, int32_t* label_len);
int rd_pipe_submit_raw_global(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                              int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr,
                              uint8_t* labels_out, const int64_t* label_off, int32_t* label_len, int32_t* status);
int rd_pipe_submit_raw_chunk(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                             int chunk_len, int step, int beam_width, uint8_t* labels_out, int32_t* label_len, int32_t* status);
/* Deliver finished groups of the reads-level pipeline to their callers, in submission order.  Without blocking when
 * wait_for <= 0; otherwise returns once at least wait_for of the batches submitted so far (counted since the context was
 * created) have been delivered, closing the open group if what is awaited sits in it.  *delivered = that count. */
int rd_pipe_progress(rd_ctx* ctx, int64_t wait_for, int64_t* delivered);
/* Batches submitted to the reads-level pipeline so far: right after a submit, the number rd_pipe_progress must reach for
 * that batch to have been delivered. */
int rd_pipe_submitted(rd_ctx* ctx, int64_t* submitted);

/* RNA model file -> dense table, without a JSON object tree.  radian/basecall.py:48-57 (json.load, then every "ACGT..." key re-keyed as a
 * tuple of label indices).  The reference's default model has 4^11 keys in ~420 MB of text; these two calls scan the one shape such a
 * file has -- an object of k-character keys over ACGT, each with an array of four JSON numbers -- straight into the [4^k][4] float64 table
 * rd_load_lm takes (row = base-4 number of the context, first character most significant):
 *   rd_lm_json_probe: k = length of the first key (1..13);
 *   rd_lm_json_fill:  the caller has filled table with NaN; every key's row is written (a repeated key keeps its last value, like a Python
 *                     dict), rows of contexts the file does not hold stay NaN (sparse model: see rd_load_lm); *n_entries = pairs read,
 *                     *n_contexts = distinct contexts.
 * Anything else in the text (escapes, another alphabet, a key of another length, NaN / Infinity, nested values, trailing text) returns
 * RD_ERR_FORMAT and decides nothing: the caller falls back to a full JSON parser, whose errors are the reference's.  Numbers are
 * converted correctly rounded and locale-independently (the double Python's float() gives).  No GPU is touched, no context is needed. */
int rd_lm_json_probe(const char* buf, size_t n, int* k_out);
int rd_lm_json_fill(const char* buf, size_t n, int k, double* table, int64_t* n_entries, int64_t* n_contexts);

/* ---- the step before the hot path, on the HOST's cores: raw signals out of fast5 files, in batches --
 * radian/basecall.py:7,70-76: `get_fast5_file(path).get_reads()`, `read.read_id`, `read.get_raw_data()` (int16 DAQ values, unscaled).
 * A handle indexes one file's reads in ont_fast5_api's order -- multi-read: the root's `read_<id>` groups by name, signal at Raw/Signal;
 * single-read: the groups of /Raw/Reads, id = the group's `read_id` attribute (else its name) -- over a read-only mapping of the file, walking
 * the CLASSIC HDF5 layout (superblock 0/1, version-1 object headers, symbol-table or compact-link groups, contiguous / compact / chunked
 * int16 datasets, chunks raw or through HDF5's built-in deflate / shuffle / Fletcher-32 filters: what libhdf5's defaults write, what
 * radian/data/reads.fast5 is, and the gzip-compressed signals of pre-VBZ MinKNOW files) with every access bounds-checked.
 *   rd_fast5_open / rd_fast5_open_mem (the caller keeps buf alive and unchanged) / rd_fast5_close
 *   rd_fast5_count       reads of the file
 *   rd_fast5_lengths     samples of reads [lo, hi): sizes the block for ...
 *   rd_fast5_read_batch  reads [lo, hi) copied back to back into samples (capacity cap), offsets[hi - lo + 1] = where each starts (the
 *                        last entry = the total); ids (nullable): the read ids, NUL-terminated, id_stride bytes apart.
 * Anything else in the file (newer superblock / object headers, fractal-heap groups, any other filter such as VBZ, a chunk that fails to
 * inflate or fails its checksum, another sample type, an address outside the file) returns RD_ERR_FORMAT and decides nothing: the caller reads the file
 * through libhdf5, whose errors are then the verdict.  ~2 us per 4096-sample read on one core (libhdf5: ~63; deflated signals 60 M samples/s against 18 M); no GPU is touched, no
 * context is needed, a handle is used by one thread at a time. */
typedef struct rd_fast5 rd_fast5;
int rd_fast5_open(const char* path, rd_fast5** out);
int rd_fast5_open_mem(const void* buf, size_t n, rd_fast5** out);
void rd_fast5_close(rd_fast5* f);
int rd_fast5_count(const rd_fast5* f, int64_t* n_reads);
int rd_fast5_lengths(rd_fast5* f, int64_t lo, int64_t hi, int64_t* n_samples);
int rd_fast5_read_batch(rd_fast5* f, int64_t lo, int64_t hi, int16_t* samples, int64_t cap, int64_t* offsets, char* ids, int id_stride);

/* ---- the step after the hot path in chunk mode, on the HOST's cores: simple_assembly + argmax --
 * radian/sequence_assembly.py:19-48, radian/basecall.py:122-123.  labels / label_len as the chunk-mode entry points return
 * them (window w at labels + w * chunk_len); read r owns windows [read_win_off[r], read_win_off[r + 1]).  The consensus of
 * read r (labels 0..3, not reversed) goes to seq_out + seq_off[r] (capacity >= the sum of its windows' label_len),
 * seq_len[r] its length -- or -1 where the reference raises IndexError (its vote matrix grows at most once per fragment).
 * Fragment placement restates difflib.SequenceMatcher (autojunk included) exactly; n_threads host threads share the
 * reads.  No GPU is touched and no context is needed. */
int rd_stitch_chunk(const uint8_t* labels, const int32_t* label_len, int chunk_len, const int32_t* read_win_off, int n_reads,
                    uint8_t* seq_out, const int64_t* seq_off, int32_t* seq_len, int n_threads);

/* ---- multi-GPU start-up: one RCCL broadcast of weights + LM table over xGMI ------------------- */
/* librccl can be loaded in this process (dlopen + symbol lookup; creates nothing).  Ranks other than the one that draws the
 * unique id call this before the collective ncclCommInitRank, so that a rank without a usable librccl is known to everyone first. */
int rd_rccl_probe(void);
int rd_rccl_unique_id(uint8_t id_out[128]);                       /* rank 0, then shared out of band */
int rd_rccl_init(rd_ctx* ctx, int rank, int nranks, const uint8_t id[128]);
int rd_rccl_bcast_model(rd_ctx* ctx, int root);                   /* weights (+ LM when loaded on root) */
/* Hand the loaded artefacts of `src` (weights in every packing, LM table) to `dst`, another context of this process: what
 * rd_rccl_bcast_model does for another rank, with a device copy as the transport.  No reference counterpart. */
int rd_clone_artifacts(rd_ctx* dst, rd_ctx* src);
int rd_rccl_allreduce_max(rd_ctx* ctx, double* inout, int n);     /* host values, max over ranks */
int rd_rccl_barrier(rd_ctx* ctx);
/* Size of the communicator as RCCL itself reports it (ncclCommCount): the evidence that N ranks really joined. */
int rd_rccl_comm_count(rd_ctx* ctx, int* nranks);
int rd_rccl_finalize(rd_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RADIAN_HIP_H */
