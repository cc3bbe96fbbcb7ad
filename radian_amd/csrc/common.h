// common.h -- shared declarations for the radian_hip library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#define RD_OK 0
#define RD_ERR_ARG (-1)
#define RD_ERR_HIP (-2)
#define RD_ERR_STATE (-3)
#define RD_ERR_NOMEM (-4)
#define RD_ERR_RCCL (-5)

void rd_set_error(const char* fmt, ...);

#define RD_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            rd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
            return RD_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

#define RD_REQUIRE(cond, ...)                                                                          \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            rd_set_error(__VA_ARGS__);                                                                 \
            return RD_ERR_ARG;                                                                         \
        }                                                                                              \
    } while (0)

// A grow-only device buffer (workspace); never shrinks, freed with the context.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T> T* as() const { return (T*)p; }
};

// Model geometry: radian/models/sig2seq.yaml:34-49
constexpr int RD_C = 256;      // tcn.nb_filters
constexpr int RD_K = 3;        // tcn.kernel_size
constexpr int RD_H = 128;      // relu_units
constexpr int RD_NCLS = 5;     // softmax_units (A,C,G,T,blank)
constexpr int RD_MAX_BLOCKS = 16;

struct Model {
    bool loaded = false;
    int nblocks = 0;
    int dil[RD_MAX_BLOCKS] = {0};
    // device tensors (packed layouts, see forward.hip)
    float* sink = nullptr;      // 1024 floats, write-only scratch
    float* w_in = nullptr;      // block0.conv0 kernel [3][256]
    float* b_in = nullptr;      // [256]
    float* w_match = nullptr;   // [256]
    float* b_match = nullptr;   // [256]
    float* w_conv[2 * RD_MAX_BLOCKS] = {nullptr};  // packed [48 chunks][256 co][16 k]; index 2*blk+which (blk0.conv0 unused)
    float* b_conv[2 * RD_MAX_BLOCKS] = {nullptr};  // [256]
    float* w_d1 = nullptr;      // packed like conv, K=256: [8 chunks][128 co][32 k]
    float* b_d1 = nullptr;      // [128]
    // split-f16 images of the same tensors (forward.hip, f16x3 section): [chunk][co][hi 32 | lo 32] halves, scaled by 1/inv_scale
    void* ws_conv[2 * RD_MAX_BLOCKS] = {nullptr};
    float inv_scale[2 * RD_MAX_BLOCKS] = {0};
    void* ws_d1 = nullptr;
    float inv_scale_d1 = 0.f;
    // three-term bf16 images (forward.hip, bf16x3 section): [chunk][co][6 slots x 8 bf16], slots swizzled, unscaled
    void* w3_conv[2 * RD_MAX_BLOCKS] = {nullptr};
    void* w3_d1 = nullptr;
    float* w_d2 = nullptr;      // [128][5] (Keras layout)
    float* b_d2 = nullptr;      // [5]
    DevBuf storage;
};

struct LM {
    bool loaded = false;
    int k = 0;                      // context length (labels)
    int table_order = 0;            // the table has 4^table_order rows (= k unless hashed)
    int hashed = 0;                 // 1: long-context mode, row = hash(context) (decode.hip)
    double* table = nullptr;        // [4^table_order][4] device
    double* d_entropy = nullptr;    // [4^k] device: entropy of each context's distribution (glibc log, computed at load)
    int sparse = 0;                 // 1: some contexts are absent (rows of NaN at rd_load_lm): d_missing has their bits
    uint32_t* d_missing = nullptr;  // bit ctx: the model does not hold this context (decode.py:83 raises KeyError on it)
    uint32_t* gate_bits = nullptr;  // bit ctx: d_entropy[ctx] < gate_r_thr
    double gate_r_thr = 0.0;
    bool gate_valid = false;
    DevBuf storage, gate_storage;   // storage = table, d_entropy, d_missing (one RCCL broadcast)
};

// One 32-row SUB-TILE of the forward: rows [t0, t0+32) of a segment whose time step 0 is global row seg_row; a
// workgroup tile is four consecutive descriptors of a layer's list (any mix of segments; lists are padded with empty ones).
// A segment is a window, a whole read (the "stream"), or the first rows of a window (a "head").  Heads only hold the
// rows that see their window's own zero left-padding; every later row equals the read's stream row, so a head's
// reads of rows t >= alt_in (conv input) / alt_res (residual) are redirected to global row alt_row + t.
struct TileDesc {
    int64_t seg_row;   // global row (activations / probabilities) of the segment's time step 0
    int64_t src_row;   // index of the segment's sample 0 in the signal buffer
    int64_t alt_row;   // global row of the stream row that equals this segment's time step 0
    int32_t t0;        // first time step of this sub-tile
    int32_t seg_len;   // time steps this layer computes and stores for the segment
    int32_t in_len;    // time steps that exist as input (zero beyond)
    int32_t alt_in;    // conv-input rows t >= alt_in come from alt_row + t   (INT32_MAX: never)
    int32_t alt_res;   // residual rows  t >= alt_res come from alt_row + t
    int32_t pad_;
};

// per-layer tile lists: index 0 = block-0 first conv (C_in = 1); 2b = first conv of block b >= 1; 2b+1 = second conv of
// block b; 2*nblocks = dense head.  Uniform windows and streams use one list for every layer.
constexpr int RD_MAX_LAYERS = 2 * 16 + 1;
struct TileLists {
    const TileDesc* d[RD_MAX_LAYERS];
    int n[RD_MAX_LAYERS];          // workgroup tiles = descriptors / 4
    int64_t rows[RD_MAX_LAYERS];   // time steps the layer evaluates (for the FLOP / byte accounting of the timers)
};

struct KernelTimer {
    // HIP-event timing of one kernel family on the launch stream (bench.py roofline leg)
    bool enabled = false;
    std::vector<hipEvent_t> starts, stops;
    size_t used = 0;
    double flops = 0.0;  // algorithmic flops accumulated over recorded launches
    double bytes = 0.0;
    std::vector<double> each_flops;   // per recorded launch (rd_timer_read_launches: the roofline by kernel variant)
    std::vector<int> each_tag;        // conv: 0 relu (a block's first conv), 1 res_ident, 2 res_match (block 0's second conv); others 0
};

// One forward "lane": a stream with its own three activation tensors.  Lane 0 is the context's main stream; the
// pipelined entry points spread consecutive batches over several lanes so that independent kernel chains overlap (the
// partially filled last round of one chain's launch is filled by the other chain's workgroups).
constexpr int RD_MAX_LANES = 4;
struct FwdLane {
    hipStream_t st = nullptr;
    DevBuf act[3];
    hipEvent_t done = nullptr;   // recorded after the lane's latest forward
};

struct rd_ctx {
    int device = 0;
    int n_cu = 256;      // compute units of the device (launch-shape decisions)
    int precision = 0;   // 0: exact fp32 MFMA (default); 1: split-f16 (f16x3); 2: three-term bf16 split (bf16x3) matrix products
    int decode_form = 0; // rd_set_decode_form
    int decode_math = 1; // rd_set_decode_math (default: glibc's operation sequence -- scores bit-identical to the reference's)
    int conv_fuse = 1;   // rd_set_conv_fuse: 1 = block 0's first conv is computed inside its second (forward.hip, FIN), 0 = its own kernel
    int conv_shape = 0;  // rd_set_conv_shape: 0 = 128-row tiles, two 256-thread workgroups per CU; 1 = 256-row tiles, one 512-thread workgroup
    int logits_f16 = 0;  // 1: the reads-level paths keep the softmax rows as f16 in HBM (10 B per time step), the decoder widens them
    hipStream_t stream = nullptr;
    hipStream_t stream_hi = nullptr;   // high priority: the beam search of the unpipelined entry points (a few long-running waves
                                       // that must not queue behind another context's forward workgroups)
    hipEvent_t ev_chain = nullptr;
    Model model;
    LM lm;
    // workspaces
    DevBuf ws_tiles, ws_raw;
    int tiles_nW = -1, tiles_T = -1;   // shape the cached uniform tile descriptors were built for
    // lanes[0].st == stream; lanes >= 1 are created on first use.  Lanes [RD_MAX_LANES, 2 RD_MAX_LANES) are the PARTITIONED
    // twins of lanes [0, RD_MAX_LANES): their streams are CU-masked to everything but the decode partition (below)
    FwdLane lanes[2 * RD_MAX_LANES];
    // Decode partition (global-mode reads pipeline): part_cus CUs of every XCD are kept free of forward workgroups and run
    // the beam search.  A beam-search wave that shares a SIMD with conv waves issuing MFMAs back to back gets about one
    // instruction issue per MFMA (measured: 17 us per time step instead of 2), and a read's search is one serial chain.
    int part_mode = -1;   // rd_set_decode_partition: -1 = chosen by beam width, 0 = off, k = k CUs per XCD
    int part_cus = 0;     // CUs per XCD the existing masked streams were created for (0: none exist)
    DevBuf ws_in, ws_probs, ws_mat, ws_seq, ws_nodes_child, ws_nodes_back, ws_labels, ws_misc;
    DevBuf ws_queue;                // the work-queue counter of beam_search_queue_kernel (decode.hip)
    DevBuf ws_wide, ws_wide_slot;   // beam widths above 51 (decode_wide.hip): per-sequence scratch block, per-trie-node slot map
    int64_t trie_budget = (int64_t)24 << 30;   // bytes of beam-search workspace one launch may ask for (rd_plan_trie_runs; rd_set_trie_budget)
    // pinned host staging
    void* h_stage = nullptr;
    size_t h_stage_cap = 0;
    bool h_stage_busy = false;   // an async copy out of h_stage was queued and the stream has not been synchronised since
    KernelTimer timer_conv, timer_decode, timer_head, timer_in;
    void* rccl = nullptr;  // RcclState*
    void* pipe = nullptr;  // Pipe* (two-stream forward/decode software pipeline over chunk-mode batches, api.hip)
    void* rpipe = nullptr; // ReadsPipe* (the same scheme over batches of whole reads, global mode / raw input, pipe_reads.hip)
    int pipe_group = 4;    // batches per beam-search launch (rd_pipe_config)
    int pipe_lanes = 2;    // forward streams the submitted batches rotate over (rd_pipe_set_lanes)
    void* plan_cache[2] = {nullptr, nullptr};  // PlanCache* for chunk / global reads-level plans
};

// forward.hip
int rd_forward_dev(rd_ctx* ctx, const float* d_windows, int nW, int T, float* d_probs, int lane = 0);
int rd_forward_tiles_dev(rd_ctx* ctx, const float* d_signal, const TileLists& tiles, int64_t total_rows, void* d_probs, int lane = 0,
                         int probs_f16 = 0 /* 1: d_probs is _Float16 [rows][5] */);
int rd_lane_get(rd_ctx* ctx, int lane, FwdLane** out);   // creates the lane's stream on first use
constexpr int RD_XCDS = 8;                               // MI355X: 8 XCDs x 32 CUs; CU-mask bit i = CU i / 8 of XCD i % 8
int rd_part_set(rd_ctx* ctx, int cus_per_xcd);           // (re)size the decode partition: drops the masked lane streams of another size
// pooled CU-masked streams: on the first k CUs of every XCD (complement = false) or on all the others
int rd_masked_stream_acquire(int device, int cus_per_xcd, bool complement, hipStream_t* out);
void rd_masked_stream_release(hipStream_t st);   // waits for the stream, then hands it back (never destroyed)
int rd_sync_lanes(rd_ctx* ctx);                          // every forward stream idle
int rd_split3_dev(rd_ctx* ctx, const float* d_in, size_t n, uint16_t* d_out);   // fp32 -> [3][n] bf16 bit patterns (hi, mid, lo)
int rd_model_halo(const rd_ctx* ctx);  // receptive field - 1 = (K-1) * 2 * sum(dilations)
// decode.hip
int rd_decode_dev(rd_ctx* ctx, const void* d_probs, int ptype /* 0 f32, 1 f64, 2 f16 rows */, const int64_t* d_seq_off, const int32_t* d_seq_len,
                  const int64_t* d_node_off, const int64_t* d_label_off, int n_seq, int64_t total_nodes, int W, int use_lm,
                  double s_thr, double r_thr, uint8_t* d_labels, int32_t* d_label_len, double* d_best_score,
                  hipStream_t stream = nullptr /* default: ctx->stream */, const int64_t* d_seq_off2 = nullptr,
                  const int32_t* d_seq_split = nullptr, int n_cu_avail = 0 /* CUs the stream may use (0: all) */,
                  int queue_wave_slots = 0 /* > 0: n_seq exceeds what those CUs keep resident: that many waves' worth of workgroups take the sequences from a queue */);
// preprocess.hip
int rd_normalise_dev(rd_ctx* ctx, const int16_t* d_raw, const int64_t* d_read_off, int n_reads, int clip, float* d_out,
                     int32_t* d_status, hipStream_t stream = nullptr /* default: ctx->stream */);
// assemble.hip
// one read of a batched assembly: its forward rows start at row src_row of the probability buffer (streamed: row t of the
// read; windowed: window i at src_row + i*T), its N assembled float64 rows go to row out_row of the output
struct AsmRead {
    int64_t src_row, out_row;
    int32_t N, nW, pad, pad_;
};
int rd_assemble_batch_dev(hipStream_t st, const void* d_probs, const AsmRead* d_reads, int n_reads, int64_t max_n, int T, int step,
                          double* d_out, int streamed, int in_f16);
int rd_gather_windows_dev(hipStream_t st, const float* d_rows, const int64_t* d_off1, const int64_t* d_off2, const int32_t* d_split,
                          const int32_t* d_valid, int n_windows, int T, float* d_out);
int rd_assemble_dev(rd_ctx* ctx, const void* d_probs, int nW, int T, int pad, int step, double* d_out, int64_t N,
                    int streamed = 0 /* 1: d_probs is the streamed forward [N][5]; row t is taken from row t */,
                    int in_f16 = 0 /* 1: d_probs rows are _Float16 */);

// api.hip
int rd_pipe_drain_decode_internal(rd_ctx* ctx);   // wait for the chunk pipeline's beam searches in flight
// pipe_reads.hip
int rd_rpipe_flush(rd_ctx* ctx);
bool rd_rpipe_idle(const rd_ctx* ctx);
int rd_rpipe_drain_decode(rd_ctx* ctx);   // wait for the pipeline's beam searches in flight (they share the trie workspace)
void rd_rpipe_destroy(rd_ctx* ctx);

// The beam search packs (parent node id << 2) | label into a 32-bit back-pointer and gives a sequence at most 1 + W * rows
// trie nodes: a sequence must satisfy 1 + W * rows < 2^29 (W = 10: 53 M rows; W = 51: 10.5 M rows).  Host-side check of
// every entry point that knows its sequence lengths (RD_ERR_ARG beyond, instead of a silently wrong traceback).
inline bool rd_decode_len_ok(int W, int64_t rows) { return 1 + (int64_t)W * rows < ((int64_t)1 << 29); }

extern "C" int rd_decode_max_width(void);
constexpr int RD_LANE_MAX_W = 256;     // decode.hip: the wave-per-sequence kernels' widest beam (rd_decode_lane_width)
constexpr int RD_HASHED_MAX_W = 64;    // ... and the widest one with hashed long contexts (rd_load_lm_hashed)
// decode_wide.hip: beam widths above RD_LANE_MAX_W, up to RD_WIDE_MAX_W (40 W bytes of LDS for the ranking keys)
constexpr int RD_WIDE_MAX_W = 1024;
int rd_decode_wide_launch(rd_ctx* ctx, hipStream_t st, const void* decode_args, int ptype, int n_seq, int64_t total_nodes, bool lm);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// decode_wide.hip's per-sequence scratch block: beams [2][W] of 48 B | c_ptot, c_pb, c_pnb [5W] doubles | c_pj, c_src [5W] ints | mq [W] ints
inline size_t rd_wide_scratch_bytes(int W)
{
    const size_t b = (size_t)2 * W * 48 + (size_t)3 * 5 * W * 8 + (size_t)2 * 5 * W * 4 + (size_t)W * 4;
    return (b + 255) & ~(size_t)255;
}

// One beam-search launch needs (1 + W * rows) trie nodes per sequence -- 20 B each (child table + back pointer), 24 B above the lane
// kernels' width (the slot map of decode_wide.hip) plus that kernel's scratch block per sequence -- for ALL its sequences at once: W x the
// rows of the launch.  A pipeline group of 96 M rows at W = 100, or a few thousand windows at W = 1024, would ask for hundreds of GB and die
// with RD_ERR_NOMEM in the middle of a job.  So every caller that builds a launch's node offsets cuts its sequences -- in launch
// order -- into RUNS whose workspace fits the context's budget (at least one sequence per run); the runs go to rd_decode_dev one after the
// other on the same stream and share the workspace.  Node offsets restart at 0 in every run.  Results do not depend on the cut.
struct TrieRun {
    int k0, k1;          // sequences [k0, k1) of the launch order
    int64_t nodes;       // their trie nodes
};
template <typename LenOf>
inline void rd_plan_trie_runs(const rd_ctx* ctx, int W, int k_begin, int k_end, LenOf len_of, int64_t* node_off, std::vector<TrieRun>& runs)
{
    const int64_t per_node = W > RD_LANE_MAX_W ? 24 : 20;
    const int64_t per_seq = W > RD_LANE_MAX_W ? (int64_t)rd_wide_scratch_bytes(W) : 0;
    int64_t nodes = 0, bytes = 0;
    int k0 = k_begin;
    for (int k = k_begin; k < k_end; k++) {
        const int64_t nk = 1 + (int64_t)W * len_of(k), bk = nk * per_node + per_seq;
        if (k > k0 && bytes + bk > ctx->trie_budget) {
            runs.push_back({k0, k, nodes});
            k0 = k;
            nodes = bytes = 0;
        }
        node_off[k] = nodes;
        nodes += nk;
        bytes += bk;
    }
    if (k_end > k0) runs.push_back({k0, k_end, nodes});
}

// Hashed long contexts (rd_load_lm_hashed) exist in the lane kernels only: refused where the arguments are checked, not when a group
// of batches is launched after earlier reads were written
#define RD_REQUIRE_WIDTH_LM(ctx, W, use_lm)                                                                                     \
    RD_REQUIRE(!((use_lm) && (ctx)->lm.loaded && (ctx)->lm.hashed && (W) > RD_HASHED_MAX_W),                                     \
               "beam width %d with a hashed long-context RNA model (rd_load_lm_hashed): hashed contexts exist for widths up to 64", (W))
