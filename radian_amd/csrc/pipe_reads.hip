// pipe_reads.hip -- software pipeline over batches of WHOLE READS inside one rd_ctx (declared in include/radian_hip.h):
// rd_pipe_submit_reads_global / rd_pipe_submit_raw_global / rd_pipe_submit_raw_chunk / rd_pipe_progress.
//
// The loop body of radian/basecall.py:77-121 for a stream of read batches.  What runs where:
//   forward lane (rotating, rd_pipe_set_lanes; each lane owns its activation tensors, staging and tile descriptors):
//       [raw form: H2D of the int16 samples out of the lane's pinned staging block -> mad_normalise (preprocess.py:24-49)]
//       -> streamed TCN forward (every time step once) into the open GROUP's probability rows
//       -> [global mode: per-read assembly (matrix_assembly.py:6-53) of the batch into the group's float64 matrix]
//   decode stream (high priority): per closed group ONE metadata upload, the beam search of every read (global: LM-gated,
//       decode.py:100-212) or window (chunk) of the group, labels + lengths (+ normalisation status) to pinned host memory.
// A group closes when it holds rd_pipe_config batches -- or, in global mode, as soon as its forward work covers the beam
// search of its longest read: a read's search is one serial chain of one time step per sample (~2 us each), which only
// the NEXT group's forwards can hide, so a group needs ~100 forward rows per chain step and no more.
// Results are delivered in submission order: rd_pipe_progress hands over finished groups without blocking (or blocks
// until a given number of submits has been delivered); rd_pipe_flush delivers everything.
#include "common.h"
#include "plan.h"
#include "../../include/radian_hip.h"
#include "../../include/radian_hip_diag.h"   // measurement / diagnostic entry points defined in this file

#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

using namespace rdi;

namespace {

// ---- copies as kernels on the stream's OWN queue ----------------------------------------------------------------------------------------
// hipMemcpyAsync hands a copy to the runtime's copy path (SDMA engines, or its blit queue), which is shared by all streams and keeps its
// order: a copy that waits for a kernel -- the labels going back to the host behind a 70-ms beam search -- holds up every later copy in it,
// among them the next group's first host-to-device copies, and the forward lanes stand still for the length of the search.  Which engine a
// copy lands on varies, so a stream of small long-read batches ran at 27.9 or at 20 M samples/s, flipping within a run; with
// HSA_ENABLE_SDMA=0 (one shared blit queue) always at 17.3 (tools/policy_alt_run.py STREAM=long; DESIGN_LOG.md round 5).  The pipeline's
// copies therefore run as an ordinary kernel in the queue of the stream they belong to: pinned host memory is device-addressable, the
// volumes are small (0.5 MB of samples per batch in, a few MB of labels per group out), and nothing outside that stream waits behind them.
__global__ void copy_bytes_kernel(char* __restrict__ dst, const char* __restrict__ src, size_t n)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        for (size_t i = i0; i + 16 <= n; i += stride) *(uint4*)(dst + i) = *(const uint4*)(src + i);
        const size_t tail = n & ~(size_t)15;
        if (blockIdx.x == 0 && threadIdx.x < (n & 15)) dst[tail + threadIdx.x] = src[tail + threadIdx.x];
    } else {
        for (size_t i = i0; i < n; i += stride)
            for (size_t k = i; k < n && k < i + 16; k++) dst[k] = src[k];
    }
}
// (as_kernel = false: the runtime's copy path -- chunk-mode groups, whose searches last a millisecond: nothing waits behind them for long, and an
// SDMA copy of the group's 4 MB of labels costs the forward nothing, where the copy kernel's waves cost the headline 1 %)
int copy_on_stream(void* dst, const void* src, size_t n, hipStream_t st, bool as_kernel = true)
{
    if (n == 0) return RD_OK;
    if (!as_kernel) {
        RD_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDefault, st));
        return RD_OK;
    }
    const size_t chunks = (n + 15) / 16;
    const int blocks = (int)std::min<size_t>((chunks + 255) / 256, 1024);
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks), dim3(256), 0, st, (char*)dst, (const char*)src, n);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// forward rows a group must hold per beam-search step of its longest read before it may close early (global mode):
// forward ~29 M rows/s = 34 ns per row; a step costs 1.6-2.0 us (W <= 12), 2.7-3.1 us (W <= 25), 4-6 us beyond
// Two regimes for the beam search of a global-mode group (a read's search is one serial chain of a step per sample):
//  * FEW sequences (at most three waves per SIMD of the decode partition -- what its LDS keeps resident): on the
//    partition's CUs, which the group's forwards keep clear, a step costs 1.6-2.0 us (W <= 12), 2.7-3.1 us (W <= 25), 4-6 us
//    beyond -- the group is worth closing as soon as its forward rows (34 ns each) cover the longest chain at that pace.
//    Batches of few reads (long reads, small steps) are submitted to the partitioned forward lanes for this;
//  * MANY sequences: the partition's few SIMDs would be the bottleneck (a saturated SIMD does ~0.15-0.45 M steps/s), so the
//    search runs on the whole chip beside the next group's conv waves, where a wave gets about one instruction issue per
//    MFMA -- 17 us per step measured at W = 10 (profiles/r03a_global_pipe_trace.txt) -- but hundreds of them run at once:
//    the group keeps growing until its forward rows cover the longest chain at THAT pace, and its forwards use every CU
//    (batches of many reads go to the unpartitioned lanes).
// Measured (tools/repro_refdefaults.py: 16 384 ragged reads, reference defaults + 12-mer LM, soft head): all on the whole
// chip 25.0 M samples/s; the longest 288 reads of every group on the partition and the rest on the whole chip 18.8 M (the
// partition becomes the bottleneck) -- hence no split inside a group, and no partition under batches of many reads (forced there
// with two sequences per wave and longest-first order: 20.0 M at 3 or 4 CUs per XCD, 18.0 M at 6).
// These constants are what a context starts from (exact fp32, glibc arithmetic: 34 ns per forward row); the figures it measures
// in its own first groups replace them (Calib below).
constexpr double kDefaultNsPerRow = 34.0;
inline int64_t chain_rows_default(int W, bool on_partition)
{
    // (W <= 6: two sequences share a wave's instructions -- beam_search2_kernel -- so beside conv waves a sequence advances
    // twice as fast per issued instruction)
    if (!on_partition) return W <= 6 ? 300 : W <= 12 ? 560 : W <= 25 ? 900 : 1500;
    return W <= 12 ? 96 : W <= 25 ? 140 : 260;
}

// Self-calibration of the group policy (round 4).  Two figures decide when a global-mode group is worth closing: what a forward row
// costs (ns, all lanes together) and what a time step of the group's longest chain costs while the NEXT group's forwards run (us).
// Both depend on things the constants above cannot know -- the matrix-product mode (f16x3 / bf16x3 forwards are 1.3-2.2x faster), the
// beam width, the arithmetic of the beam search, whether the search has the decode partition to itself, the device -- so the context
// measures them with HIP events: the end of every submit's lane work (forward rate over a window of consecutive submits) and both
// ends of every group's beam search (elapsed / longest chain).  Until a figure exists the constant stands in, scaled by the measured
// forward rate when that is known.
struct Calib {
    static constexpr int NF = 16, WIN = 6, ND = 4;
    struct Fwd {
        hipEvent_t e0 = nullptr, e1 = nullptr;   // both ends of a submit's lane work
        int64_t rows = 0;
        int prec = 0;
    } f[NF];
    int64_t f_next = 0;        // submits recorded so far (sample i lives in f[i % NF])
    int64_t f_seen = 0;        // samples before this index have been used or dropped
    double ns_row[3] = {0.0, 0.0, 0.0};   // per matrix-product mode, all lanes together; 0: not measured yet: the smallest of the last NH windows
    static constexpr int NH = 5;
    double ns_hist[3][NH] = {{0.0}};      // the last NH windows' figures per mode (a ring)
    int ns_n[3] = {0, 0, 0};              // windows seen per mode
    struct Dec {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        int64_t longest = 0;
        uint32_t key = 0;
        bool pending = false;
    } d[ND];
    int64_t d_next = 0;
    std::vector<std::pair<uint32_t, double>> us_step;   // (key, us per time step of the longest chain), few entries

    // m = waves per SIMD of the decode partition that the measured group put there (1..3), 0 = the whole chip beside conv waves
    static uint32_t key_of(int W, int m, int math, int prec, int lm) { return (uint32_t)W | (uint32_t)m << 8 | (uint32_t)math << 11 | (uint32_t)prec << 12 | (uint32_t)lm << 14; }
    double* find(uint32_t key)
    {
        for (auto& e : us_step)
            if (e.first == key) return &e.second;
        return nullptr;
    }
};

// waves a group of n sequences puts on every SIMD of a decode partition of part_cus CUs per XCD (a sequence is half a wave at W <= 6,
// one wave up to 12, two up to 25, four beyond)
inline int part_waves_per_simd(int part_cus, int W, int64_t n_seq)
{
    const int64_t simds = (int64_t)RD_XCDS * part_cus * 4;
    const int64_t waves = W <= 6 ? (n_seq + 1) / 2 : W <= 12 ? n_seq : W <= 25 ? 2 * n_seq : W <= 64 ? 4 * n_seq : W <= 128 ? 5 * n_seq : 10 * n_seq;
    const int64_t m = (waves + simds - 1) / simds;
    return (int)(m < 1 ? 1 : m);
}

// completed measurements -> estimates (never blocks: an event that has not fired yet is looked at again later)
void calib_harvest(Calib& c)
{
    // Forward pace = the time during which AT LEAST ONE lane was working on a forward, over the rows those forwards produced: the
    // union of the (e0, e1) intervals of WIN consecutive submits.  Not the time between the completions of consecutive submits --
    // that contains every pause of the pipeline (the host waiting for a group slot because a beam search was not covered), reads
    // slow, shrinks the rule, covers less: a runaway seen on the reference-defaults job (80 ns per row "measured", 13 M samples/s
    // instead of 28 M) -- and not a lane's own duration over the lane count either, which reads 20-25 % fast whenever the lanes
    // overlap only partly (configs[4] leg: 27 ns "measured" against 36).
    while (c.f_next - c.f_seen >= Calib::WIN) {
        if (c.f_next - c.f_seen >= Calib::NF) {         // (older than the ring: overwritten, or about to be by the submit in progress)
            c.f_seen = c.f_next - Calib::NF + 1;
            continue;
        }
        const int64_t j0 = c.f_seen;
        if (hipEventQuery(c.f[(j0 + Calib::WIN - 1) % Calib::NF].e1) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        double a[Calib::WIN], b[Calib::WIN];
        int64_t rows = 0;
        bool ok = true;
        const Calib::Fwd& ref = c.f[j0 % Calib::NF];
        for (int k = 0; k < Calib::WIN && ok; k++) {
            const Calib::Fwd& f = c.f[(j0 + k) % Calib::NF];
            float ms0 = 0.f, ms1 = 0.f;
            ok = f.prec == ref.prec && f.rows > 0 && hipEventElapsedTime(&ms0, ref.e0, f.e0) == hipSuccess &&
                 hipEventElapsedTime(&ms1, ref.e0, f.e1) == hipSuccess && ms1 > ms0;
            a[k] = ms0;
            b[k] = ms1;
            rows += f.rows;
        }
        if (ok) {
            // busy time between the END of the window's first forward and the end of its last, for the rows of all but the first:
            // (counted from the first one's START the window would hold its ramp -- with two lanes, seven slots for six forwards)
            const double t0 = b[0];
            rows -= ref.rows;
            for (int k = 1; k < Calib::WIN; k++) a[k] = a[k] < t0 ? t0 : a[k];
            for (int i = 2; i < Calib::WIN; i++)          // by start time (a handful of intervals)
                for (int k = i; k > 1 && a[k] < a[k - 1]; k--) {
                    std::swap(a[k], a[k - 1]);
                    std::swap(b[k], b[k - 1]);
                }
            double busy = 0.0, lo = a[1], hi = b[1] > a[1] ? b[1] : a[1];
            for (int k = 2; k < Calib::WIN; k++) {
                if (b[k] <= a[k]) continue;               // (ended before the first one did: nothing inside the window)
                if (a[k] > hi) {
                    busy += hi - lo;
                    lo = a[k];
                    hi = b[k];
                } else if (b[k] > hi) {
                    hi = b[k];
                }
            }
            busy += hi - lo;
            if (rows > 0 && busy > 0.0) {
                // The SMALLEST of the last five windows, not a running mean: a lane's interval now and then holds a stall that is not the
                // forward's -- 80-200 ms inside one submit's lane work, while a long beam search runs or a second process uses the GPU
                // (tools/policy_probe.py; one window in ten read 220 ns per row against 36-50).  A mean follows such windows, the rule
                // hits its lower clamp, groups close uncovered, the host waits for their beam searches -- which is what makes stalls.
                // Stalls only ever lengthen an interval, so the smallest recent figure is the forward's own pace; a real slowdown
                // (another matrix-product mode has its own ring; a shared GPU) moves all five within five windows, and until then
                // the rule errs towards LARGER groups: more coverage, not less.
                const double ns = busy * 1e6 / (double)rows;
                const int pr = ref.prec;
                c.ns_hist[pr][c.ns_n[pr] % Calib::NH] = ns;
                c.ns_n[pr]++;
                double v[Calib::NH];
                const int n = c.ns_n[pr] < Calib::NH ? c.ns_n[pr] : Calib::NH;
                for (int i = 0; i < n; i++) v[i] = c.ns_hist[pr][i];
                c.ns_row[pr] = *std::min_element(v, v + n);
            }
        } else {
            (void)hipGetLastError();
        }
        c.f_seen += Calib::WIN / 2;      // (windows overlap by half)
    }
    for (auto& d : c.d) {
        if (!d.pending || hipEventQuery(d.e1) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        float ms = 0.f;
        if (d.longest > 0 && hipEventElapsedTime(&ms, d.e0, d.e1) == hipSuccess && ms > 0.f) {
            const double us = (double)ms * 1e3 / (double)d.longest;
            if (double* e = c.find(d.key)) *e = 0.5 * *e + 0.5 * us;
            else c.us_step.emplace_back(d.key, us);
        } else {
            (void)hipGetLastError();
        }
        d.pending = false;
    }
}

// Forward rows a group must hold per time step of its longest read before it closes.  m = waves per SIMD that the group's sequences
// put on the decode partition (part_waves_per_simd), 0 = the group decodes on the whole chip beside conv waves.
//  * On the partition the rule is measured: (us per step of a chain at THIS occupancy m) / (ns per forward row) + 20 % (with + 5 % a
//    64-read step of saturated rows closed a group of its own, 8.6 ms of search under 9.2 ms of forward, and every jitter stalled the
//    lanes: secondary_global_lm 28.3 -> 26.2 M samples/s; with + 30 % the soft-head leg took a third step per group: 28.2 -> 26.4 M).  The pace
//    is keyed by m because it depends on it (W = 25: 3.3 us at one wave per SIMD, 4.9 at three): a rule fed with "elapsed / longest"
//    of whatever group ran last feeds back on itself -- larger groups, slower chains, larger groups -- until the partition's
//    sequence limit flips the group onto the whole chip (configs[4] leg 26.3 -> 21-24 M samples/s; profiles/r04_policy_ab.txt).
//    Until a pace has been measured at an occupancy, the round-3 constant stands in as a pace (96 rows x 34 ns = 3.3 us, ...).
//  * Beside conv waves a group's search time is its total work over the chip's rate -- proportional to the group's own size --
//    so there is no pace to measure (following elapsed / longest ran away to the row cap: reference-defaults job 27.9 -> 13.4 M);
//    the constant for the beam width is scaled by the measured forward pace.
int64_t chain_rows(const rd_ctx* ctx, Calib& c, int W, int m, int use_lm)
{
    const bool on_partition = m > 0;
    const int64_t def = chain_rows_default(W, on_partition);
    const double ns = c.ns_row[ctx->precision];
    if (ns <= 0.0) return def;
    double rows = (double)def * kDefaultNsPerRow / ns;   // the constant, for the forward this context really runs
    if (on_partition) {
        const int mm = m > 3 ? 3 : m;
        const double* us = c.find(Calib::key_of(W, mm, ctx->decode_math, ctx->precision, use_lm));
        if (us) rows = *us * 1e3 / ns * 1.2;
    }
    const double lo = (double)def / 3.0, hi = (double)def * 8.0;
    rows = rows < lo ? lo : rows > hi ? hi : rows;
    return (int64_t)rows;
}
// sequences the partition decodes at chain pace: three waves per SIMD (a sequence is 1 / 2 / 4 waves for W <= 12 / 25 / 51;
// ~12 KiB of LDS per sequence keeps 13 resident per CU).  Measured at W = 10 on 64 SIMDs: 128 / 256 sequences 28.2 / 27.8 M
// samples/s, 512 sequences 22.1 M (tools/global_pipe_bench.py)
inline int part_seq_limit(int part_cus, int W)
{
    const int waves3 = RD_XCDS * part_cus * 4 * 3;      // three waves per SIMD
    return W <= 6 ? 2 * waves3 : W <= 12 ? waves3 : W <= 25 ? waves3 / 2 : W <= 64 ? waves3 / 4 : W <= 128 ? waves3 / 5 : waves3 / 10;   // (W <= 6: two sequences per wave; 65 ... 128: five waves, 43 KB of LDS: three per CU; 129 ... 256: ten waves, 80 KB: two per CU)
}
// sequences the work-queue search of an oversubscribed partition keeps resident: its workgroups are one sequence each (1 / 2 / 4 waves for
// W <= 12 / 25 / 51 -- no two-sequences-per-wave form), three waves per SIMD
inline int queue_resident_seqs(int part_cus, int W)
{
    const int waves3 = RD_XCDS * part_cus * 4 * 3;
    return W <= 12 ? waves3 : W <= 25 ? waves3 / 2 : W <= 64 ? waves3 / 4 : W <= 128 ? waves3 / 5 : waves3 / 10;
}
constexpr int64_t kGroupRowsCap = 96ll << 20;   // rows a group may gather while it waits for coverage (~6 GB of probabilities + matrix)
// CUs per XCD of the decode partition for a beam width (rd_set_decode_partition -1).  ALWAYS A MULTIPLE OF FOUR: a CU-masked
// queue's bits are dealt round over the four shader engines of an XCD (bit j of an XCD -> engine j mod 4), and the dispatcher
// hands every engine the same share of a launch's workgroups, so the forward runs at the pace of the engine with the FEWEST
// CUs left.  Measured (tools/global_pipe_bench.py 1 partK, 64-read steps, ms per step): K = 0: 8.72 | 1, 2, 3, 4: 9.15, 9.15,
// 9.16, 9.17 | 5, 6, 7, 8: 10.58, 10.62, 10.61, 10.63 -- the fourth CU of a group of four is free, the fifth costs as much as
// the eighth.  (Rounds 2-3 used 3 / 5 / 8; 5 was the worst choice on the table.)  32 CUs = 128 SIMDs step ~0.45 M
// sequence-steps/s each at W = 10 (twice the forward's ~29 M rows/s) and keep up at W = 25; wider beams take a second four.
// Round 6: widths 65 ... 128 (five waves per sequence, 13-15 us per step) take a third four: at 8 the partition's 153 resident sequences deliver ~11 M
// steps/s against a forward of 21 M rows/s; at 12, 230 sequences and 15-16 M against 18 M (tools/policy_probe.py fp32 100 0 <part>: short / long /
// alternating / ragged streams 10.7 / 9.6 / 11.5 / 9.2 M samples/s at 8, 15.3 / 8.6 / 16.2 / 13.0 at 12, 16.4 / 13.6 / 14.9 / 10.2 at 16).
inline int auto_part_cus(int W) { return W <= 25 ? 4 : W <= 64 ? 8 : W <= 128 ? 12 : 16; }

struct RSub {                 // one submitted batch inside a group
    int n_seq = 0, seq0 = 0;  // its decoded sequences (global: reads; chunk: windows) = [seq0, seq0 + n_seq) of the group
    int n_reads = 0, read0 = 0;
    uint8_t* user_labels = nullptr;
    int32_t* user_lens = nullptr;
    int32_t* user_status = nullptr;        // raw form: per read
    std::vector<int64_t> user_label_off;   // per sequence
};

struct RSeq {                 // one decoded sequence of a group
    int64_t off1, off2;       // source rows (rows [0, split) from off1, the rest from off2 - see DecodeArgs); global: off1 only
    int32_t split, len;
    int32_t is64;             // global: rows come from the group's float64 matrix
    int64_t label_off;        // inside the group's label buffer
};

struct RSlot {                // a group
    DevBuf probs, mat, meta, labels, status;
    void* h_meta = nullptr;
    size_t h_meta_cap = 0;
    void* h_out = nullptr;
    size_t h_out_cap = 0;
    hipEvent_t dec_done = nullptr;
    bool busy = false;        // beam search launched, results not yet delivered
    int64_t launch_seq = 0;   // order of the launches on the decode stream
    hipStream_t dec_stream = nullptr;   // where this group's beam search runs / ran
    int part = 0;             // CUs per XCD of the decode partition the group's forwards kept clear (0: none)
    // what makes a group homogeneous
    int mode = -1, W = 0, f16 = 0, use_lm = 0;
    double s_thr = 0.0, r_thr = 0.0;
    int64_t rows = 0, rows64 = 0, cap_rows = 0, labels_total = 0, longest = 0;
    int64_t steps_total = 0;  // time steps of all the group's sequences (global mode): the partition's total work
    bool oversub = false;     // global mode: more sequences than the partition keeps resident, by decision (see submit): stays on the partition
    int n_reads = 0;
    size_t status_off = 0;    // inside h_out, valid while busy
    std::vector<int> order;   // decode order of the sequences (global: float64 ones first), valid while busy
    int n64 = 0;
    std::vector<RSub> subs;
    std::vector<RSeq> seqs;
    unsigned lane_mask = 0;
    int64_t grow_hint = 0;    // rows to reserve next time the slot is empty (a group had to close for lack of room)
};

struct RLane {                // per forward lane: staging + descriptors of the lane's latest submit
    DevBuf tiles;             // tile descriptors | AsmRead records
    DevBuf raw;               // read offsets | status | int16 samples
    DevBuf sig;               // normalised signal
    void* h_stage = nullptr;
    size_t h_stage_cap = 0;
    hipEvent_t staged = nullptr;
    bool staged_pending = false;
    // the plan the descriptors on the device were built from
    bool valid = false;
    int chunk = -1, step = -1, mode = -1, nblocks = -1;
    int dil[RD_MAX_BLOCKS] = {0};
    std::vector<int64_t> lens;
    ReadsPlan plan;
    bool streamed = false;
    TileLists lists;
    size_t n_desc = 0;
};

struct ReadsPipe {
    hipStream_t s_dec = nullptr;
    hipStream_t s_part = nullptr;   // the decode partition's stream (global mode), masked to part_cus CUs of every XCD
    int part_cus = 0;
    static constexpr int NSLOT = 3;   // groups in existence: one gathering + up to two whose beam searches are queued on the decode stream
    RSlot slot[NSLOT];
    RLane lane[2 * RD_MAX_LANES];   // per PHYSICAL lane: a lane and its partitioned twin are different streams, so each has its own staging / signal / descriptors
    int cur = 0;
    int next_lane = 0;
    int64_t submitted = 0, delivered = 0, launches = 0;
    int64_t queue_launches = 0, limit_closes = 0;   // groups searched through the work queue (oversubscribed partition); groups closed at the partition's limit instead
    hipStream_t last_dec = nullptr;   // stream of the latest beam-search launch
    hipEvent_t ev_switch = nullptr;
    Calib calib;
};

void slot_reset(RSlot& s)
{
    s.subs.clear();
    s.seqs.clear();
    s.order.clear();
    s.rows = s.rows64 = s.labels_total = s.longest = 0;
    s.steps_total = 0;
    s.oversub = false;
    s.n_reads = 0;
    s.lane_mask = 0;
    s.mode = -1;
}

int rpipe_get(rd_ctx* ctx, ReadsPipe** out)
{
    if (!ctx->rpipe) {
        ReadsPipe* p = new ReadsPipe();
        ctx->rpipe = p;   // (owned by the context from here on: rd_rpipe_destroy frees whatever was created)
        int lo = 0, hi = 0;
        RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        RD_HIP(hipStreamCreateWithPriority(&p->s_dec, hipStreamNonBlocking, hi));
        for (int i = 0; i < ReadsPipe::NSLOT; i++) RD_HIP(hipEventCreateWithFlags(&p->slot[i].dec_done, hipEventDisableTiming));
        RD_HIP(hipEventCreateWithFlags(&p->ev_switch, hipEventDisableTiming));
        for (auto& f : p->calib.f) {                                       // (timing events: the policy's measurements)
            RD_HIP(hipEventCreate(&f.e0));
            RD_HIP(hipEventCreate(&f.e1));
        }
        for (auto& d : p->calib.d) {
            RD_HIP(hipEventCreate(&d.e0));
            RD_HIP(hipEventCreate(&d.e1));
        }
    }
    *out = (ReadsPipe*)ctx->rpipe;
    return RD_OK;
}

// deliver a finished (or awaited) group to its callers
int slot_collect(ReadsPipe* p, RSlot& s)
{
    if (!s.busy) return RD_OK;
    RD_HIP(hipEventSynchronize(s.dec_done));
    s.busy = false;
    const uint8_t* hl = (const uint8_t*)s.h_out;
    const size_t o_len = align_up((size_t)s.labels_total + 16, 256);
    const int32_t* hlen = (const int32_t*)((const char*)s.h_out + o_len);
    const int32_t* hstat = (const int32_t*)((const char*)s.h_out + s.status_off);
    int rc = RD_OK;
    // hlen is in decode order
    std::vector<int32_t> len_of(s.seqs.size());
    for (size_t k = 0; k < s.order.size(); k++) len_of[s.order[k]] = hlen[k];
    for (const RSub& sb : s.subs) {
        for (int i = 0; i < sb.n_seq; i++) {
            const RSeq& q = s.seqs[sb.seq0 + i];
            const int32_t n = len_of[sb.seq0 + i];
            if (n == RD_LEN_MISSING_CONTEXT) {   // sparse LM: the search reached an absent context (the caller raises KeyError)
                sb.user_lens[i] = n;
                continue;
            }
            if (n < 0 || n > q.len) {
                rd_set_error("pipeline: sequence %d produced an impossible label length %d (rows %d)", sb.seq0 + i, n, q.len);
                rc = RD_ERR_STATE;
                continue;
            }
            sb.user_lens[i] = n;
            if (n) memcpy(sb.user_labels + sb.user_label_off[i], hl + q.label_off, (size_t)n);
        }
        if (sb.user_status)
            for (int r = 0; r < sb.n_reads; r++) sb.user_status[r] = hstat[sb.read0 + r];
    }
    p->delivered += (int64_t)s.subs.size();
    slot_reset(s);
    return rc;
}

// beam search + copy-out of everything forwarded into the group so far
int slot_launch_decode(rd_ctx* ctx, ReadsPipe* p, RSlot& s)
{
    if (s.seqs.empty() || s.busy) return RD_OK;
    int rc;
    const size_t n = s.seqs.size();
    // decode order: global mode runs the float64 (assembled) reads first, then the float32 ones (reads that one window
    // covers: the reference decodes their float32 rows as they are, matrix_assembly.py:46-53)
    s.order.clear();
    for (size_t i = 0; i < n; i++)
        if (s.seqs[i].is64) s.order.push_back((int)i);
    s.n64 = (int)s.order.size();
    for (size_t i = 0; i < n; i++)
        if (!s.seqs[i].is64) s.order.push_back((int)i);
    // longest first inside each run: workgroups are dispatched in index order as slots free up, so the long chains start at once
    // and the short ones fill in behind them (a launch then takes max(longest chain, work / rate) instead of rounds x longest), and
    // the two sequences that share a wave at W <= 6 have similar lengths
    auto by_len = [&](int x, int y) { return s.seqs[x].len != s.seqs[y].len ? s.seqs[x].len > s.seqs[y].len : x < y; };
    std::sort(s.order.begin(), s.order.begin() + s.n64, by_len);
    std::sort(s.order.begin() + s.n64, s.order.end(), by_len);
    const size_t a8 = align_up(n * 8, 256), a4 = align_up(n * 4, 256);
    const size_t o_off2 = a8, o_node = 2 * a8, o_lab = 3 * a8, o_len = 4 * a8, o_split = o_len + a4, o_llen = o_split + a4;
    const size_t meta_bytes = o_llen + a4;
    if ((rc = pinned_reserve(&s.h_meta, &s.h_meta_cap, meta_bytes))) return rc;
    if (s.meta.reserve(meta_bytes)) return RD_ERR_NOMEM;
    char* hm = (char*)s.h_meta;
    int64_t *h1 = (int64_t*)hm, *h2 = (int64_t*)(hm + o_off2), *hn = (int64_t*)(hm + o_node), *hl = (int64_t*)(hm + o_lab);
    int32_t *hlen = (int32_t*)(hm + o_len), *hsp = (int32_t*)(hm + o_split);
    for (size_t k = 0; k < n; k++) {
        const RSeq& q = s.seqs[s.order[k]];
        h1[k] = q.off1;
        h2[k] = q.off2;
        hsp[k] = q.split;
        hlen[k] = q.len;
        hl[k] = q.label_off;
    }
    // the two passes -- and, inside a pass, the runs a wide beam or a very large group is cut into (rd_plan_trie_runs) -- go one after the
    // other on the decode stream and share the trie workspace
    std::vector<TrieRun> runs[2];
    rd_plan_trie_runs(ctx, s.W, 0, s.n64, [&](int k) { return (int64_t)hlen[k]; }, hn, runs[0]);
    rd_plan_trie_runs(ctx, s.W, s.n64, (int)n, [&](int k) { return (int64_t)hlen[k]; }, hn, runs[1]);
    if (s.labels.reserve((size_t)s.labels_total + 16)) return RD_ERR_NOMEM;
    const size_t ho_len = align_up((size_t)s.labels_total + 16, 256);
    s.status_off = ho_len + align_up(n * 4, 256);
    if ((rc = pinned_reserve(&s.h_out, &s.h_out_cap, s.status_off + (size_t)s.n_reads * 4 + 16))) return rc;
    if ((rc = rd_pipe_drain_decode_internal(ctx))) return rc;   // (beam searches of the chunk pipeline use the same trie workspace)
    // the group's beam search runs on the decode partition when its forwards kept one clear and its sequences are few
    // enough to run there at chain pace, else on the whole chip; the two streams share the trie workspace, so a launch on
    // one waits for the other's latest
    const bool on_part = s.part && ((int)n <= part_seq_limit(s.part, s.W) || s.oversub);
    hipStream_t ds = on_part ? p->s_part : p->s_dec;
    if (p->last_dec && p->last_dec != ds) {
        RD_HIP(hipEventRecord(p->ev_switch, p->last_dec));
        RD_HIP(hipStreamWaitEvent(ds, p->ev_switch, 0));
    }
    p->last_dec = ds;
    s.dec_stream = ds;
    for (int l = 0; l < 2 * RD_MAX_LANES; l++)   // every forward / assembly that wrote into this group has finished
        if (s.lane_mask & (1u << l)) RD_HIP(hipStreamWaitEvent(ds, ctx->lanes[l].done, 0));
    const bool kc = s.mode == 1;     // global mode: the search behind these copies runs for tens of milliseconds
    if ((rc = copy_on_stream(s.meta.p, s.h_meta, o_llen, ds, kc))) return rc;
    char* dm = (char*)s.meta.p;
    // global mode: the group's beam search is timed for the group policy (Calib)
    Calib::Dec* cd = nullptr;
    // (an oversubscribed partition -- work queue, every slot taken from the start -- gives the pace at three waves per SIMD as long as its longest
    // chain, not its total work, decides when it ends: steps_total / slots well below the longest chain)
    const bool q_launch = on_part && (int)n > part_seq_limit(s.part, s.W);
    if (s.mode == 1 && s.longest > 0 && (!q_launch || s.steps_total / queue_resident_seqs(s.part, s.W) <= s.longest * 4 / 5)) {
        calib_harvest(p->calib);
        Calib::Dec& d = p->calib.d[p->calib.d_next % Calib::ND];
        if (!d.pending) {
            cd = &d;
            p->calib.d_next++;
            d.longest = s.longest;
            d.key = Calib::key_of(s.W, on_part ? std::min(3, part_waves_per_simd(s.part, s.W, (int64_t)n)) : 0, ctx->decode_math, ctx->precision, s.use_lm);
            RD_HIP(hipEventRecord(d.e0, ds));
        }
    }
    for (int pass = 0; pass < 2; pass++)
    for (const TrieRun& run : runs[pass]) {
        const int k0 = run.k0, k1 = run.k1;
        const bool chunk = s.mode == 0;
        const void* src = pass == 0 ? s.mat.p : s.probs.p;
        const int ptype = pass == 0 ? 1 : (s.f16 ? 2 : 0);
        rc = rd_decode_dev(ctx, src, ptype, (const int64_t*)dm + k0, (const int32_t*)(dm + o_len) + k0, (const int64_t*)(dm + o_node) + k0,
                           (const int64_t*)(dm + o_lab) + k0, k1 - k0, run.nodes, s.W, s.use_lm, s.s_thr, s.r_thr, s.labels.as<uint8_t>(),
                           (int32_t*)(dm + o_llen) + k0, nullptr, ds, chunk ? (const int64_t*)(dm + o_off2) + k0 : nullptr,
                           chunk ? (const int32_t*)(dm + o_split) + k0 : nullptr, on_part ? RD_XCDS * s.part : 0,
                           // an oversubscribed partition (s.oversub): resident workgroups + a work queue, three waves per SIMD
                           on_part && (int)n > part_seq_limit(s.part, s.W) ? RD_XCDS * s.part * 4 * 3 : 0);
        if (rc) return rc;
    }
    if (cd) {
        RD_HIP(hipEventRecord(cd->e1, ds));
        cd->pending = true;
    }
    if ((rc = copy_on_stream(s.h_out, s.labels.p, (size_t)s.labels_total, ds, kc))) return rc;
    if ((rc = copy_on_stream((char*)s.h_out + ho_len, dm + o_llen, n * 4, ds, kc))) return rc;
    if (s.n_reads && s.status.p)
        if ((rc = copy_on_stream((char*)s.h_out + s.status_off, s.status.p, (size_t)s.n_reads * 4, ds, kc))) return rc;
    RD_HIP(hipEventRecord(s.dec_done, ds));
    s.busy = true;
    s.launch_seq = ++p->launches;
    if (s.mode == 1 && s.part && s.oversub && (int)s.seqs.size() > part_seq_limit(s.part, s.W)) p->queue_launches++;
    return RD_OK;
}

int close_group(rd_ctx* ctx, ReadsPipe* p)
{
    int rc = slot_launch_decode(ctx, p, p->slot[p->cur]);
    if (rc) return rc;
    p->cur = (p->cur + 1) % ReadsPipe::NSLOT;
    return RD_OK;
}

// a group that can take `rows` more probability rows with these decode parameters; closes / recycles groups as needed
int open_slot(rd_ctx* ctx, ReadsPipe* p, int mode, int W, int f16, int use_lm, double s_thr, double r_thr, int part, int64_t rows,
              int64_t expect_rows, RSlot** out)
{
    int rc;
    RSlot* s = &p->slot[p->cur];
    // The current slot may still hold the group BEFORE the last one (launched, not yet handed to its callers): that group goes out first.
    // (Round 5: judged by the tests below while still busy -- its rows plus this batch "did not fit" -- it was "closed" again, which only
    // flipped the slots, and the host then waited for the group it had just launched: no forward under that group's beam search.)
    if (s->busy && (rc = slot_collect(p, *s))) return rc;
    const bool same = s->mode == mode && s->W == W && s->f16 == f16 && s->use_lm == use_lm && s->s_thr == s_thr && s->r_thr == r_thr && s->part == part;
    if (!s->seqs.empty() && (!same || s->rows + rows > s->cap_rows)) {
        if (same) s->grow_hint = 2 * (s->rows + rows);   // closed for lack of room: the slot grows when it is empty again
        if ((rc = close_group(ctx, p))) return rc;
        s = &p->slot[p->cur];
    }
    if (s->busy && (rc = slot_collect(p, *s))) return rc;   // the slot's previous group goes to its callers first
    if (s->seqs.empty()) {
        // room for the group this batch will probably become (expect_rows: what covers its longest read's chain), so that a
        // group closes because it is covered, not because its slot happens to be small
        int64_t want = rows > s->grow_hint ? rows : s->grow_hint;
        if (expect_rows > want) want = expect_rows;
        if (want > s->cap_rows) {
            const int64_t cap = want + want / 4;
            // (nothing in flight reads this slot: its group was delivered)
            if (s->probs.reserve((size_t)cap * 20)) return RD_ERR_NOMEM;
            s->cap_rows = cap;
        }
        // the assembled matrix exists for global-mode groups only: sized when the first one arrives, whatever grew the slot before
        if (mode == 1 && s->mat.reserve((size_t)(s->cap_rows + 1) * 40)) return RD_ERR_NOMEM;
    }
    s->mode = mode;
    s->part = part;
    s->W = W;
    s->f16 = f16;
    s->use_lm = use_lm;
    s->s_thr = s_thr;
    s->r_thr = r_thr;
    *out = s;
    return RD_OK;
}

// Plan (segments, tile descriptors) of a batch on a lane, host side.  *miss: the batch's read lengths differ from the lane's
// previous batch, R.plan was rebuilt and its descriptors (*n_desc of them) must be uploaded (lane_plan_upload).
int lane_plan_build(rd_ctx* ctx, RLane& R, const int64_t* read_off, int n_reads, int chunk, int step, int mode, bool* miss, size_t* n_desc)
{
    const int halo = rd_model_halo(ctx);
    std::vector<int64_t> lens(n_reads);
    for (int r = 0; r < n_reads; r++) lens[r] = read_off[r + 1] - read_off[r];
    bool hit = R.valid && R.chunk == chunk && R.step == step && R.mode == mode && R.nblocks == ctx->model.nblocks && R.lens == lens;
    for (int b = 0; hit && b < ctx->model.nblocks; b++) hit = R.dil[b] == ctx->model.dil[b];
    *miss = !hit;
    *n_desc = R.n_desc;
    if (hit) return RD_OK;
    R.valid = false;
    R.plan = ReadsPlan();
    int rc = mode == 0 ? plan_reads_chunk(ctx->model, read_off, n_reads, chunk, step, halo, R.plan)
                       : plan_reads_global(ctx->model, read_off, n_reads, chunk, step, halo, R.plan, &R.streamed);
    if (rc) return rc;
    R.n_desc = plan_pad_tiles(R.plan);
    *n_desc = R.n_desc;
    R.chunk = chunk;
    R.step = step;
    R.mode = mode;
    R.nblocks = ctx->model.nblocks;
    for (int b = 0; b < RD_MAX_BLOCKS; b++) R.dil[b] = b < ctx->model.nblocks ? ctx->model.dil[b] : 0;
    R.lens.swap(lens);
    return RD_OK;
}

// descriptors of a rebuilt plan: into the staging block at hs, one copy to the lane's descriptor buffer on the lane's
// stream (behind the lane's previous forward, which still reads the old ones)
int lane_plan_upload(FwdLane* L, RLane& R, char* hs)
{
    ReadsPlan& P = R.plan;
    size_t off = 0;
    for (int li = 0; li < RD_MAX_LAYERS; li++) {
        R.lists.d[li] = nullptr;
        R.lists.n[li] = 0;
        R.lists.rows[li] = 0;
    }
    for (int li = 0; li < P.n_layers; li++) {
        if (P.per_layer || li == 0) {
            const std::vector<TileDesc>& v = P.tiles[li];
            if (!v.empty()) memcpy(hs + off * sizeof(TileDesc), v.data(), v.size() * sizeof(TileDesc));
            R.lists.d[li] = R.tiles.as<TileDesc>() + off;
            R.lists.n[li] = (int)(v.size() / 4);
            R.lists.rows[li] = P.rows[li];
            off += v.size();
        } else {
            R.lists.d[li] = R.lists.d[0];
            R.lists.n[li] = R.lists.n[0];
            R.lists.rows[li] = R.lists.rows[0];
        }
    }
    if (off) {
        int rc_ = copy_on_stream(R.tiles.p, hs, off * sizeof(TileDesc), L->st);
        if (rc_) return rc_;
    }
    // the host copies of the descriptors are not needed again (the per-sequence vectors of the plan are)
    for (int li = 0; li < RD_MAX_LAYERS; li++) std::vector<TileDesc>().swap(P.tiles[li]);
    R.valid = true;
    return RD_OK;
}

int check_args(rd_ctx* ctx, const void* signal, const int64_t* read_off, int n_reads, int chunk_len, int step, int W)
{
    RD_REQUIRE(ctx && signal && read_off, "null argument");
    RD_REQUIRE(n_reads >= 1 && chunk_len >= 1, "bad shape");
    RD_REQUIRE(step >= 1 && step <= chunk_len, "step %d must be in [1, chunk_len]", step);
    RD_REQUIRE(W >= 1 && W <= rd_decode_max_width(), "beam_width %d out of range", W);
    RD_REQUIRE(read_off[0] == 0, "read_off[0] must be 0");
    for (int r = 0; r < n_reads; r++) {
        RD_REQUIRE(read_off[r + 1] > read_off[r], "read %d is empty (the caller skips empty reads, basecall.py:77-82)", r);
        RD_REQUIRE(rd_decode_len_ok(W, read_off[r + 1] - read_off[r]), "read %d has %lld samples; beam width %d supports at most %lld (1 + W * rows < 2^29)", r,
                   (long long)(read_off[r + 1] - read_off[r]), W, (long long)((((int64_t)1 << 29) - 2) / W));
    }
    if (!ctx->model.loaded) {
        rd_set_error("no weights loaded (rd_load_weights)");
        return RD_ERR_STATE;
    }
    return RD_OK;
}

// One submit: [raw -> normalise] -> plan -> forward -> [assembly] on the next lane, its sequences appended to the open group.
// raw == nullptr: d_signal is the normalised signal resident in HBM.
int submit(rd_ctx* ctx, int mode, const float* d_signal, const int16_t* raw, int clip, const int64_t* read_off, int n_reads, int chunk_len,
           int step, int W, int use_lm, double s_thr, double r_thr, uint8_t* labels_out, const int64_t* label_off, int32_t* label_len,
           int32_t* status)
{
    int rc;
    RD_REQUIRE_WIDTH_LM(ctx, W, use_lm && mode == 1);   // (refused at the submit, before anything of the job is queued -- not when the group is launched)
    RD_HIP(hipSetDevice(ctx->device));
    ReadsPipe* p = nullptr;
    if ((rc = rpipe_get(ctx, &p))) return rc;
    const int n_lanes = ctx->pipe_lanes < 1 ? 1 : ctx->pipe_lanes;
    // global mode, a batch of few reads (long reads, small steps): its forward keeps `part` CUs of every XCD clear and the
    // group's beam search runs there at chain pace; a batch of many reads uses every CU (see chain_rows)
    // (the CU masks are laid out for 8 XCDs x 32 CUs: on any other device there is no partition)
    const int part_cus = mode == 1 && ctx->n_cu == RD_XCDS * 32 ? (ctx->part_mode < 0 ? auto_part_cus(W) : ctx->part_mode) : 0;
    const int part = part_cus && n_reads <= part_seq_limit(part_cus, W) / 2 ? part_cus : 0;
    if (part && part != p->part_cus) {   // (part_cus of the pipe = the size its masked streams exist for)
        // another partition size (first use, or the beam width's class changed): drain, then new masked streams
        if ((rc = rd_rpipe_flush(ctx))) return rc;
        if (p->s_part) {
            rd_masked_stream_release(p->s_part);
            p->s_part = nullptr;
            if (p->last_dec != p->s_dec) p->last_dec = nullptr;
        }
        if ((rc = rd_part_set(ctx, part))) return rc;
        if ((rc = rd_masked_stream_acquire(ctx->device, part, false, &p->s_part))) return rc;
        p->part_cus = part;
    }
    const int lane = p->next_lane % n_lanes;
    const int plane = lane + (part ? RD_MAX_LANES : 0);   // the lane's partitioned twin: same staging, masked stream, own activations
    FwdLane* L = nullptr;
    if ((rc = rd_lane_get(ctx, plane, &L))) return rc;
    RLane& R = p->lane[plane];
    const size_t n_samples = (size_t)read_off[n_reads];
    const int f16 = ctx->logits_f16;

    // ---- host side first: the lane's staging block is free again, the plan is known, every buffer is large enough
    if (R.staged_pending) {
        RD_HIP(hipEventSynchronize(R.staged));
        R.staged_pending = false;
    }
    if (!R.staged) RD_HIP(hipEventCreateWithFlags(&R.staged, hipEventDisableTiming));
    bool miss = false;
    size_t n_desc = 0;
    if ((rc = lane_plan_build(ctx, R, read_off, n_reads, chunk_len, step, mode, &miss, &n_desc))) return rc;
    const ReadsPlan& P = R.plan;
    // staging block: [read offsets | raw samples] | descriptors (plan miss) | AsmRead records (global mode)
    const size_t st_off = align_up((size_t)(n_reads + 1) * 8, 256);
    const size_t st_raw = raw ? align_up(n_samples * 2 + 16, 256) : 0;
    const size_t st_desc = miss ? align_up(n_desc * sizeof(TileDesc), 256) : 0;
    const size_t st_asm = mode == 1 ? align_up((size_t)n_reads * sizeof(AsmRead), 256) : 0;
    const size_t o_raw = st_off, o_desc = raw ? st_off + st_raw : 0, o_asm = o_desc + st_desc;
    if ((rc = pinned_reserve(&R.h_stage, &R.h_stage_cap, o_asm + st_asm + 256))) return rc;
    char* hs = (char*)R.h_stage;
    // device: descriptors | AsmRead records
    const size_t d_asm = align_up(n_desc * sizeof(TileDesc), 256);
    if (d_asm + st_asm + 256 > R.tiles.cap) {
        RD_HIP(hipStreamSynchronize(L->st));   // the lane's previous forward may still read the old descriptors
        if (!miss) {                           // (cannot happen: a hit has the same reads, hence the same sizes)
            rd_set_error("internal: descriptor buffer too small on a plan hit");
            return RD_ERR_STATE;
        }
        if (R.tiles.reserve(d_asm + st_asm + (d_asm + st_asm) / 4 + 256)) {
            R.valid = false;
            return RD_ERR_NOMEM;
        }
    }
    // device: read offsets | status | int16 samples ; normalised signal
    const size_t d_st = st_off, d_raw = st_off + align_up((size_t)n_reads * 4, 256);
    if (raw && (d_raw + st_raw > R.raw.cap || n_samples * 4 + 16 > R.sig.cap)) {
        RD_HIP(hipStreamSynchronize(L->st));   // the lane's previous forward still reads its signal
        if (R.raw.reserve(d_raw + st_raw + (d_raw + st_raw) / 4) || R.sig.reserve(n_samples * 4 + n_samples + 16)) return RD_ERR_NOMEM;
    }

    // ---- the group this batch joins (may close / deliver earlier groups)
    RSlot* s = nullptr;
    int64_t expect_rows = 0;
    int64_t longest_b = 0;
    if (mode == 1) {
        for (int r = 0; r < n_reads; r++) longest_b = std::max<int64_t>(longest_b, read_off[r + 1] - read_off[r]);
        calib_harvest(p->calib);
        expect_rows = std::min<int64_t>(kGroupRowsCap, chain_rows(ctx, p->calib, W, part ? 2 : 0, use_lm) * longest_b + 2 * P.total_rows);
    }
    if (mode == 1 && part) {
        // A partition-lane group that THIS batch would take past what the partition keeps resident (part_seq_limit), before its forward
        // rows cover its longest chain.  Flipping it onto the whole chip (beside the next group's conv waves) is a regime that needs 5-6x
        // the rows (configs[4] leg: 26.3 M samples/s closing here, 20.6 M flipping), so round 4 closed the group at this point, covered or
        // not -- which starves a stream whose batches mix a few long reads with many short ones: the short reads use up the sequence
        // budget after a few batches and the group closes with 40 ms of forward under a 150-ms chain (tools/policy_probe.py, batches
        // alternating between 64 x 4096 and 6 x 40960 samples at W = 25: 6.5 M samples/s against 32 / 22 M for either kind alone).  Now the
        // group may keep growing ON the partition -- its workgroups queue longest-first: the long chains start at once, the short ones
        // fill the slots behind them -- as long as the partition's total work keeps up with the forward:
        //     steps_total x pace(three waves per SIMD) / resident sequences  <=  forward time of the group's rows
        // and closes here, as before, where it does not (many short reads at a wide beam).  Decided when the crossing batch ARRIVES, with
        // that batch counted in: a rule that predicts "another batch like the last one" flips a coin on a stream of alternating batches.
        RSlot& g = p->slot[p->cur];
        const int limit = part_seq_limit(part, W);
        if (!g.busy && !g.seqs.empty() && g.mode == 1 && g.part == part && g.W == W && !g.oversub && (int)g.seqs.size() <= limit &&
            (int)g.seqs.size() + n_reads > limit) {
            const double ns = p->calib.ns_row[ctx->precision] > 0.0 ? p->calib.ns_row[ctx->precision] : kDefaultNsPerRow;
            const double* us3 = p->calib.find(Calib::key_of(W, 3, ctx->decode_math, ctx->precision, use_lm));
            const double pace3 = us3 ? *us3 : (double)chain_rows_default(W, true) * kDefaultNsPerRow * 1e-3;
            const double work_ms = (double)(g.steps_total + (int64_t)n_samples) * pace3 * 1e-3 / (double)queue_resident_seqs(part, W);
            const double fwd_ms = (double)(g.rows + P.total_rows) * ns * 1e-6;
            // (a quarter over is tolerated: a partition 25 % behind the forward costs that much at worst; closing here costs the uncovered chain.
            // Beyond that the group closes here only if its longest chain is covered by now: a partition that cannot keep up is the
            // bottleneck whatever the groups' size, but a group closed with a 150-ms chain over 20 ms of forward stalls the lanes for the
            // difference -- bf16x3 at W = 25, where the partition runs at 0.9-1.5 of the forward's pace depending on what the pace
            // measurements last saw, alternated between 36 and 24 M samples/s on this rule alone: 11 of 37 groups closed here uncovered.)
            const int64_t chain_need = chain_rows(ctx, p->calib, W, 3, use_lm) * std::max<int64_t>(g.longest, longest_b);
            const bool uncovered = g.rows + P.total_rows < chain_need;
            if ((work_ms <= 1.25 * fwd_ms || uncovered) && W <= RD_LANE_MAX_W /* (the work-queue kernel's shapes) */ && !(use_lm && ctx->lm.hashed)) g.oversub = true;
            else {
                p->limit_closes++;
                if ((rc = close_group(ctx, p))) return rc;
            }
        }
    }
    if ((rc = open_slot(ctx, p, mode, W, f16, use_lm, s_thr, r_thr, part, P.total_rows, expect_rows, &s))) return rc;

    // ---- its sequences (and, in global mode, the per-read assembly records), not yet part of the group
    RSub sb;
    sb.seq0 = (int)s->seqs.size();
    sb.read0 = s->n_reads;
    sb.n_reads = n_reads;
    sb.user_labels = labels_out;
    sb.user_lens = label_len;
    sb.user_status = raw ? status : nullptr;
    std::vector<RSeq> seqs;
    int64_t longest = 0, rows64 = s->rows64, labels_total = s->labels_total, max_n = 0;
    int n64 = 0;
    if (mode == 1) {
        AsmRead* ar = (AsmRead*)(hs + o_asm);
        for (int r = 0; r < n_reads; r++) {
            const int nW = P.read_win_off[r + 1] - P.read_win_off[r];
            const int pad = P.valid[r];
            const int64_t N = assembled_rows(nW, chunk_len, pad, step);
            RD_REQUIRE(N == read_off[r + 1] - read_off[r], "internal: assembled length mismatch for read %d", r);
            RSeq q;
            q.split = 0;
            q.len = (int32_t)N;
            q.is64 = assembled_is_f64(nW, chunk_len, pad, step);
            q.label_off = labels_total;
            if (q.is64) {
                AsmRead& a = ar[n64++];
                a.src_row = s->rows + P.read_row[r];
                a.out_row = rows64;
                a.N = (int32_t)N;
                a.nW = nW;
                a.pad = pad;
                a.pad_ = 0;
                q.off1 = q.off2 = rows64;   // float64 rows go behind the group's previous ones
                rows64 += N;
                if (N > max_n) max_n = N;
            } else {
                q.off1 = q.off2 = s->rows + P.read_row[r];   // single coverage: the forward's rows are consecutive time steps
            }
            labels_total += N;
            if (N > longest) longest = N;
            seqs.push_back(q);
            sb.user_label_off.push_back(label_off[r]);
        }
        sb.n_seq = n_reads;
    } else {
        for (int w = 0; w < P.n_windows; w++) {
            RSeq q;
            q.off1 = P.off1[w] + s->rows;
            q.off2 = P.off2[w] + s->rows;
            q.split = P.split[w];
            q.len = P.valid[w];
            q.is64 = 0;
            q.label_off = labels_total;
            labels_total += chunk_len;
            seqs.push_back(q);
            sb.user_label_off.push_back((int64_t)w * chunk_len);
        }
        sb.n_seq = P.n_windows;
    }
    // (what the kernels below index: checked on the host before anything is launched)
    if ((size_t)(s->rows + P.total_rows) * (f16 ? 10 : 20) > s->probs.cap || (n64 && (size_t)rows64 * 40 > s->mat.cap)) {
        rd_set_error("internal: group buffers too small (%lld + %lld rows, %lld matrix rows; probs %zu B, matrix %zu B)", (long long)s->rows,
                     (long long)P.total_rows, (long long)rows64, s->probs.cap, s->mat.cap);
        return RD_ERR_STATE;
    }
    if (raw && (size_t)(s->n_reads + n_reads) * 4 > s->status.cap) {
        // growing moves the statuses already gathered for this group (few bytes; the lanes that wrote them must be done)
        DevBuf nb;
        if (nb.reserve((size_t)(s->n_reads + n_reads) * 8 + 4096)) return RD_ERR_NOMEM;
        if (s->n_reads && s->status.p) {
            for (int l = 0; l < 2 * RD_MAX_LANES; l++)
                if (s->lane_mask & (1u << l)) RD_HIP(hipStreamSynchronize(ctx->lanes[l].st));
            RD_HIP(hipMemcpy(nb.p, s->status.p, (size_t)s->n_reads * 4, hipMemcpyDeviceToDevice));
        }
        s->status.release();
        s->status = nb;
    }

    // ---- the lane's work, in stream order: copies out of the staging block, normalise, forward, assembly
    Calib::Fwd* cf = nullptr;
    if (mode == 1) {   // (the policy exists for global-mode groups; chunk-mode groups close by a batch count)
        calib_harvest(p->calib);
        cf = &p->calib.f[p->calib.f_next % Calib::NF];
        RD_HIP(hipEventRecord(cf->e0, L->st));
    }
    const float* sig = d_signal;
    if (raw) {
        memcpy(hs, read_off, (size_t)(n_reads + 1) * 8);
        memcpy(hs + o_raw, raw, n_samples * 2);
        char* base = (char*)R.raw.p;
        if ((rc = copy_on_stream(base, hs, (size_t)(n_reads + 1) * 8, L->st))) return rc;
        if ((rc = copy_on_stream(base + d_raw, hs + o_raw, n_samples * 2, L->st))) return rc;
        if ((rc = rd_normalise_dev(ctx, (const int16_t*)(base + d_raw), (const int64_t*)base, n_reads, clip, R.sig.as<float>(),
                                   (int32_t*)(base + d_st), L->st)))
            return rc;
        sig = R.sig.as<float>();
    }
    if (miss && (rc = lane_plan_upload(L, R, hs + o_desc))) {
        R.valid = false;
        return rc;
    }
    AsmRead* d_ar = (AsmRead*)((char*)R.tiles.p + d_asm);
    if (n64 && (rc = copy_on_stream(d_ar, hs + o_asm, (size_t)n64 * sizeof(AsmRead), L->st))) return rc;
    RD_HIP(hipEventRecord(R.staged, L->st));   // the staging block is free once these copies are done
    R.staged_pending = true;
    const size_t rb = f16 ? 10 : 20;
    if ((rc = rd_forward_tiles_dev(ctx, sig, R.lists, P.total_rows, (char*)s->probs.p + (size_t)s->rows * rb, plane, f16))) return rc;
    if (n64 && (rc = rd_assemble_batch_dev(L->st, s->probs.p, d_ar, n64, max_n, chunk_len, step, s->mat.as<double>(), R.streamed ? 1 : 0, f16)))
        return rc;
    if (raw)
        if ((rc = copy_on_stream((char*)s->status.p + (size_t)s->n_reads * 4, (char*)R.raw.p + d_st, (size_t)n_reads * 4, L->st))) return rc;
    RD_HIP(hipEventRecord(L->done, L->st));   // the decode stream waits for this before it reads the group
    if (cf) {   // the policy's forward-pace measurement: the end of this submit's lane work
        RD_HIP(hipEventRecord(cf->e1, L->st));
        cf->rows = P.total_rows;
        cf->prec = ctx->precision;
        p->calib.f_next++;
    }

    // ---- the batch is part of the group
    s->seqs.insert(s->seqs.end(), seqs.begin(), seqs.end());
    s->rows64 = rows64;
    s->labels_total = labels_total;
    s->lane_mask |= 1u << plane;
    s->n_reads += n_reads;
    s->rows += P.total_rows;
    if (longest > s->longest) s->longest = longest;
    if (mode == 1) s->steps_total += (int64_t)n_samples;     // (a read's search takes one time step per sample)
    s->subs.push_back(std::move(sb));
    p->next_lane = (lane + 1) % n_lanes;
    p->submitted++;
    bool close;
    if (mode == 1) {
        // global mode: by coverage of the longest read's chain (see chain_rows), not by a batch count
        const int limit = part ? part_seq_limit(part, W) : 0;
        const bool few = part && ((int)s->seqs.size() <= limit || s->oversub);
        const int m = few ? std::min(3, part_waves_per_simd(part, W, (int64_t)s->seqs.size())) : 0;
        int64_t need = chain_rows(ctx, p->calib, W, m, use_lm) * s->longest;
        // An oversubscribed partition (decided when the crossing batch arrived, above) finishes no sooner than its total work allows: every
        // resident slot steps at the saturated pace, so the group's forward rows must also cover  steps_total x pace(3) / slots  -- a rate
        // against a rate, hence WITHOUT the 20 % that chain_rows adds for the one chain that decides a covered group's end (x 5/6): with
        // the margin a partition at 0.9 of the forward's pace could never "cover" itself and the group grew to its buffer's cap -- 27
        // batches, half-second searches, 20 M samples/s on 64 x 4096-sample batches at W = 25.  A partition slower than the forward (the
        // pre-check tolerates 1.25x) covers nothing however long the group: there the group ends with its chain covered and three rounds
        // of resident sequences gathered.
        const int resident = part ? queue_resident_seqs(part, W) : 1;
        const int64_t need_chain = need;
        if (s->oversub) need = std::max(need, chain_rows(ctx, p->calib, W, 3, use_lm) * 5 / 6 * s->steps_total / resident);
        close = s->rows >= need || s->rows >= kGroupRowsCap || (s->oversub && s->rows >= need_chain && (int)s->seqs.size() >= 3 * resident);
        // (the very first group of a context closes with its first batch: nothing is decoding yet, and its chains start one
        // group's forward time earlier -- a quarter of a second on a job of long reads)
        close = close || p->launches == 0;
    } else {
        close = (int)s->subs.size() >= ctx->pipe_group;
    }
    if (close) return close_group(ctx, p);
    return RD_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ read-out of the policy
// What the context has measured for its group policy so far (0: not yet) and the rule it would apply now.
extern "C" int rd_pipe_policy_read(rd_ctx* ctx, int beam_width, int on_partition, int use_lm, double* ns_per_row, double* us_per_step,
                                   int64_t* rows_per_step)
{
    RD_REQUIRE(ctx && ns_per_row && us_per_step && rows_per_step, "rd_pipe_policy_read: null argument");
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "beam_width %d out of range", beam_width);
    RD_REQUIRE(on_partition >= 0 && on_partition <= 3, "rd_pipe_policy_read: on_partition %d (0 = whole chip, 1..3 = waves per SIMD of the decode partition)", on_partition);
    RD_HIP(hipSetDevice(ctx->device));
    ReadsPipe* p = nullptr;
    int rc = rpipe_get(ctx, &p);
    if (rc) return rc;
    calib_harvest(p->calib);
    *ns_per_row = p->calib.ns_row[ctx->precision];
    const double* us = p->calib.find(Calib::key_of(beam_width, on_partition, ctx->decode_math, ctx->precision, use_lm != 0));
    *us_per_step = us ? *us : 0.0;
    *rows_per_step = chain_rows(ctx, p->calib, beam_width, on_partition, use_lm != 0);
    return RD_OK;
}

// ------------------------------------------------------------------------------------------------ internal hooks
bool rd_rpipe_idle(const rd_ctx* ctx)
{
    const ReadsPipe* p = (const ReadsPipe*)ctx->rpipe;
    if (!p) return true;
    for (const RSlot& s : p->slot)
        if (!s.seqs.empty() || s.busy) return false;
    return true;
}

int rd_rpipe_drain_decode(rd_ctx* ctx)
{
    ReadsPipe* p = (ReadsPipe*)ctx->rpipe;
    bool any_busy = false;
    if (p)
        for (const RSlot& s : p->slot) any_busy |= s.busy;
    if (any_busy) {
        if (p->s_dec) RD_HIP(hipStreamSynchronize(p->s_dec));
        if (p->s_part) RD_HIP(hipStreamSynchronize(p->s_part));
    }
    return RD_OK;
}

int rd_rpipe_flush(rd_ctx* ctx)
{
    ReadsPipe* p = (ReadsPipe*)ctx->rpipe;
    if (!p) return RD_OK;
    int rc;
    if ((rc = slot_launch_decode(ctx, p, p->slot[p->cur]))) return rc;
    for (;;) {   // every launched group, in launch order
        RSlot* first = nullptr;
        for (RSlot& s : p->slot)
            if (s.busy && (!first || s.launch_seq < first->launch_seq)) first = &s;
        if (!first) return RD_OK;
        if ((rc = slot_collect(p, *first))) return rc;
    }
}

void rd_rpipe_destroy(rd_ctx* ctx)
{
    ReadsPipe* p = (ReadsPipe*)ctx->rpipe;
    if (!p) return;
    (void)rd_sync_lanes(ctx);
    if (p->s_dec) (void)hipStreamSynchronize(p->s_dec);
    rd_masked_stream_release(p->s_part);   // (CU-masked streams are pooled, never destroyed: forward.hip)
    if (p->ev_switch) (void)hipEventDestroy(p->ev_switch);
    for (auto& f : p->calib.f) {
        if (f.e0) (void)hipEventDestroy(f.e0);
        if (f.e1) (void)hipEventDestroy(f.e1);
    }
    for (auto& d : p->calib.d) {
        if (d.e0) (void)hipEventDestroy(d.e0);
        if (d.e1) (void)hipEventDestroy(d.e1);
    }
    for (int i = 0; i < ReadsPipe::NSLOT; i++) {
        RSlot& s = p->slot[i];
        s.probs.release();
        s.mat.release();
        s.meta.release();
        s.labels.release();
        s.status.release();
        if (s.h_meta) (void)hipHostFree(s.h_meta);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.dec_done) (void)hipEventDestroy(s.dec_done);
    }
    for (int i = 0; i < 2 * RD_MAX_LANES; i++) {
        RLane& R = p->lane[i];
        R.tiles.release();
        R.raw.release();
        R.sig.release();
        if (R.h_stage) (void)hipHostFree(R.h_stage);
        if (R.staged) (void)hipEventDestroy(R.staged);
    }
    if (p->s_dec) (void)hipStreamDestroy(p->s_dec);
    delete p;
    ctx->rpipe = nullptr;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int rd_pipe_submit_reads_global(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len, int step,
                                           int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                                           const int64_t* label_off, int32_t* label_len)
{
    int rc = check_args(ctx, d_signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_off && label_len, "rd_pipe_submit_reads_global: null output");
    return submit(ctx, 1, d_signal, nullptr, 0, read_off, n_reads, chunk_len, step, beam_width, use_lm ? 1 : 0, s_thr, r_thr, labels_out, label_off,
                  label_len, nullptr);
}

extern "C" int rd_pipe_submit_raw_global(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip, int chunk_len,
                                         int step, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                                         const int64_t* label_off, int32_t* label_len, int32_t* status)
{
    int rc = check_args(ctx, raw, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_off && label_len && status, "rd_pipe_submit_raw_global: null output");
    return submit(ctx, 1, nullptr, raw, outlier_clip, read_off, n_reads, chunk_len, step, beam_width, use_lm ? 1 : 0, s_thr, r_thr, labels_out,
                  label_off, label_len, status);
}

extern "C" int rd_pipe_submit_raw_chunk(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip, int chunk_len,
                                        int step, int beam_width, uint8_t* labels_out, int32_t* label_len, int32_t* status)
{
    int rc = check_args(ctx, raw, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_len && status, "rd_pipe_submit_raw_chunk: null output");
    return submit(ctx, 0, nullptr, raw, outlier_clip, read_off, n_reads, chunk_len, step, beam_width, 0, 0.0, 0.0, labels_out, nullptr, label_len,
                  status);
}

extern "C" int rd_pipe_stats(rd_ctx* ctx, int64_t* out, int n)
{
    RD_REQUIRE(ctx && out && n >= 1, "rd_pipe_stats: null argument");
    const ReadsPipe* p = (const ReadsPipe*)ctx->rpipe;
    const int64_t v[5] = {p ? p->submitted : 0, p ? p->delivered : 0, p ? p->launches : 0, p ? p->queue_launches : 0, p ? p->limit_closes : 0};
    for (int i = 0; i < n && i < 5; i++) out[i] = v[i];
    return RD_OK;
}

extern "C" int rd_pipe_submitted(rd_ctx* ctx, int64_t* submitted)
{
    RD_REQUIRE(ctx && submitted, "rd_pipe_submitted: null argument");
    const ReadsPipe* p = (const ReadsPipe*)ctx->rpipe;
    *submitted = p ? p->submitted : 0;
    return RD_OK;
}

extern "C" int rd_pipe_progress(rd_ctx* ctx, int64_t wait_for, int64_t* delivered)
{
    RD_REQUIRE(ctx, "rd_pipe_progress: null context");
    ReadsPipe* p = (ReadsPipe*)ctx->rpipe;
    if (!p) {
        if (delivered) *delivered = 0;
        RD_REQUIRE(wait_for <= 0, "rd_pipe_progress: waiting for %lld submits, none made", (long long)wait_for);
        return RD_OK;
    }
    RD_REQUIRE(wait_for <= p->submitted, "rd_pipe_progress: waiting for %lld submits, %lld made", (long long)wait_for, (long long)p->submitted);
    RD_HIP(hipSetDevice(ctx->device));
    int rc;
    for (;;) {
        // groups finish in launch order
        RSlot* first = nullptr;
        for (int i = 0; i < ReadsPipe::NSLOT; i++)
            if (p->slot[i].busy && (!first || p->slot[i].launch_seq < first->launch_seq)) first = &p->slot[i];
        if (first) {
            const bool need = p->delivered < wait_for;
            if (need || hipEventQuery(first->dec_done) == hipSuccess) {
                if ((rc = slot_collect(p, *first))) return rc;
                continue;
            }
            (void)hipGetLastError();   // hipErrorNotReady is not an error
            break;
        }
        if (p->delivered < wait_for) {   // what is awaited sits in the open group: close it
            if ((rc = close_group(ctx, p))) return rc;
            continue;
        }
        break;
    }
    if (delivered) *delivered = p->delivered;
    return RD_OK;
}
