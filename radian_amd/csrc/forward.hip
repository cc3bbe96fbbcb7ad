// forward.hip -- Sig2Seq signal model forward: TCN (6 dilated causal residual blocks, 256 filters, k=3)
// -> Dense(128) -> ReLU -> Dense(5) -> softmax.
//
// Replaces `sig_model.predict(batch)` (radian/basecall.py:88-93) for the graph built by
// radian/model.py:52-89 with radian/models/sig2seq.yaml:34-49; the residual block is keras-tcn 3.5's:
//   x1 = relu(conv_d(x)); x1 = relu(conv_d(x1)); out = relu(match(x) + x1)
// with match = 1x1 conv in block 0 (C_in 1 -> 256) and identity elsewhere; causal padding means
//   out[t] = b + sum_j W[j] . x[t - (K-1-j) d]   with x[<0] = 0 inside each window.
//
// Layout in HBM: activations [row][256] fp32 (1 KiB per time step, channel-contiguous) in ONE row space shared by all
// segments of a batch (windows, whole reads, window heads -- see TileDesc); three tensors: block input, block output
// (ping-pong) and the mid activation.
// Weights are repacked once at load into the exact LDS image of each K-chunk:
//   conv  [chunk = (ci/16)*3 + tap][co 0..255][ci%16]     (16 KiB per chunk, 48 chunks per conv)
//   dense [chunk = ci/16]          [h  0..127][ci%16]
//
// The dilated conv is an implicit GEMM  M = rows (time), N = 256 (co), K = 768 (tap, ci)  on the exact-fp32 matrix
// instruction v_mfma_f32_32x32x2_f32 (gfx950 has no TF32; fp32 MFMA = 157 TFLOP/s peak).  One 256-thread workgroup
// (4 waves, 2x2) owns a 128(t) x 256(co) output tile: all output channels of a time tile, so each activation row is
// read once per tap.  Per K-chunk (one tap, 16 input channels) the A tile (128 shifted rows x 16) and the B tile
// (256 co x 16) go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into 64-B rows
// with XOR-swizzled 16-B slots (conflict-free ds_read_b128).  Three 24-KiB stages, one barrier per chunk, the DMA of
// chunk c+2 issued piecewise between the MFMAs of chunk c and counted by hand (uncounted inline-asm loads + an explicit
// s_waitcnt vmcnt(N): with the builtin hipcc drains vmcnt(0) before the next ds_read).  Two workgroups per CU.
// Each wave holds a 64 x 128 accumulator (8 MFMA tiles = 128 VGPRs).  One ds_read_b128 feeds four MFMAs: lane (r, h)
// reads k = 8g+4h .. 8g+4h+3 of its row, and MFMA number kr of the group consumes element kr of both operands, i.e.
// k-pair (8g+kr, 8g+4+kr).  Bias, ReLU, the residual add (identity or block-0 1x1 match) and the second ReLU are fused
// into the epilogue.  The head reuses the same core with N = 128, then reduces 128 -> 5 and applies softmax from LDS.
#include "common.h"

#include <algorithm>
#include <vector>
#include <mutex>

#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;       // time steps per workgroup tile
constexpr int BK = 16;        // K-chunk depth (64-B LDS rows)

enum { EPI_RELU = 0, EPI_RES_IDENT = 1, EPI_RES_MATCH = 2, EPI_HEAD = 3 };

// The zero rows of a forward's three activation tensors (the source of the causal left padding, one row behind the batch's last).
// The first kernel of a forward clears them with its workgroup 0 -- nothing reads them before the second kernel, nothing ever
// writes them, and their place moves with the batch's row count.  (They were three hipMemsetAsync launches in front of every
// forward: on a lane of the two-lane pipeline every launch boundary costs 0.1-0.35 ms before the next kernel's first
// workgroup runs -- profiles/r03h_conv_slots.txt -- so three empty launches were ~0.6 ms of lane time per step.)
struct ZeroRows {
    uint4* row[3];
    int n16;   // 16-byte pieces per row
};
__device__ __forceinline__ void clear_zero_rows(const ZeroRows& z)
{
    if (blockIdx.x != 0) return;
    for (int r = 0; r < 3; r++)
        for (int i = threadIdx.x; i < z.n16; i += blockDim.x) z.row[r][i] = make_uint4(0u, 0u, 0u, 0u);
}

struct ConvArgs {
    const float* in;      // [nW][T][256]
    float* out;           // [nW][T][256]     (EPI_HEAD: unused)
    const float* wpk;     // packed weights
    const float* bias;    // [BN]
    const float* resid;   // EPI_RES_IDENT: [nW][T][256] (may alias out)
    const float* x;       // EPI_RES_MATCH: raw windows [nW][T]
    const float* wmatch;  // [256]
    const float* bmatch;  // [256]
    const float* w2;      // EPI_HEAD: [128][5]
    const float* b2;      // EPI_HEAD: [5]
    float* probs;         // EPI_HEAD: [nW][T][5]
    int probs_f16;        // EPI_HEAD: 1 = probs is _Float16 [rows][5] (f16 logits mode)
    int zero_row;         // a row of `in` that holds zeros and is never written: source of the causal left padding
    float* sink;          // 1024 floats nobody reads: target of the stores past a segment's end
    const TileDesc* tiles; // one per workgroup
    int dil;
    // FIN variant (block 0's second conv with the first conv folded in): the first conv's parameters and the zero rows to clear
    const float* w_in;    // [3][256]
    const float* b_in;    // [256]
    ZeroRows zr;
};

// block 0, first conv, one output: relu(b + w0 x[t - 2d] + w1 x[t - d] + w2 x[t]) as ONE fma chain in this order -- written out so that
// tcn_in_kernel and the conv kernel that computes these values on the fly (FIN) produce the same bits
__device__ __forceinline__ float conv_in_value(float b, float w0, float w1, float w2, float x0, float x1, float x2)
{
    const float v = __builtin_fmaf(x2, w2, __builtin_fmaf(x1, w1, __builtin_fmaf(x0, w0, b)));
    return v > 0.f ? v : 0.f;
}

// LDS image of a K-chunk tile: [row][16 floats] (64-B rows, no padding -- LDS-DMA writes 1 KiB per wave-instruction
// contiguously), with the four 16-B slots of each row XOR-swizzled by (row >> 2) & 3: the 16 lanes of a ds_read_b128
// group hold four runs of 4 consecutive rows (row & 3 = 0..3 picks the 64-B quarter of the 256-B bank row) whose
// row >> 2 differ mod 4, so the group hits 16 different 16-B bank slots.  The swizzle is applied on the global SOURCE
// address of the DMA and again on the read.  (The split-f16 kernel uses the same image with [16 hi | 16 lo] halves.)
//
// The LDS-DMA instruction is issued by inline asm, outside hipcc's s_waitcnt bookkeeping: with the builtin
// (__builtin_amdgcn_global_load_lds) hipcc drains vmcnt(0) before the first ds_read that follows (it cannot tell the
// LDS stages apart), which would serialise the prefetch of the next chunks with the MFMAs of this one.  The kernels
// count these loads by hand (wait_dma_and_barrier<N>).  M0 carries the wave-uniform LDS byte address and is
// compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16_uncounted(const float* src, float* lds_dst)
{
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}
// Same, for a source that is one scalar base for the wave + 16 B per lane (the packed weights): scalar-base addressing,
// and piece r of the wave's share through the instruction offset, which the hardware adds to the global AND the LDS
// address (the wave's pieces are contiguous in both).
template <int OFFSET>
__device__ __forceinline__ void glds16_uncounted_saddr(unsigned lane_off, const void* sbase, float* lds_dst)
{
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(sbase), "s"(dst), "n"(OFFSET)
                 : "memory");
}
// All but the newest LEAVE LDS-DMA instructions of this wave have landed (loads return in order); then the workgroup
// barrier: everyone's pieces of that chunk are in LDS and every wave is past its reads of the stage overwritten next.
template <int LEAVE, bool LGKM = false>
__device__ __forceinline__ void wait_dma_and_barrier()
{
    // LGKM: this wave's LDS stores (the FIN variant writes A tiles with ds_write) have completed too
    if constexpr (LGKM) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LEAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LEAVE) : "memory");
}

// ReLU of an accumulator.  Written as `v > 0 ? v : 0` hipcc emits v_max v, v, v (quieting a possible signalling NaN) in front
// of the v_max with 0: 128 extra vector instructions per wave in the epilogue.
__device__ __forceinline__ float relu_raw(float v)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

// 16-B global access at a wave-uniform base (kept in scalar registers: the asm is opaque to hipcc) + a per-lane BYTE offset:
// `global_* v, voffset, s[base:base+1]`.  (Left alone, hipcc folds base + lane offset into one 64-bit vector address and then
// spends two vector adds per further row of the tile; `zext(off) << 2` would not match the scalar-base form either.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __attribute__((address_space(1))) char* sgpr_base(const void* p)
{
    // (readfirstlane: free on a value that already lives in scalar registers, and keeps the statement legal when register
    // pressure made hipcc park the wave-uniform descriptor fields in vector registers)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)p >> 32));
    uint64_t u = (uint64_t)hi << 32 | lo;
    asm("" : "+s"(u));
    return (__attribute__((address_space(1))) char*)u;
}
__device__ __forceinline__ float4 gload16(const void* sbase, unsigned byte_off)
{
    asm("" : "+v"(byte_off));   // (keeps the 32-bit offset's zero-extension in the block of the access, where the selector can see it)
    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(sgpr_base(sbase) + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore16(void* sbase, unsigned byte_off, float4 v)
{
    asm("" : "+v"(byte_off));
    *(__attribute__((address_space(1))) f32x4*)(sgpr_base(sbase) + byte_off) = f32x4{v.x, v.y, v.z, v.w};
}

// Dense(5) + softmax of a head tile (model.py:73-75), shared by the three matrix-product modes.  hs = the tile's Dense(128) + ReLU rows
// in LDS, [ROWS][RD_H + 1]; w2s = Dense(5)'s [128][5] kernel + [5] bias in LDS; part = [ROWS][5] floats of LDS.  TWO threads per row
// (2 x ROWS threads): threads < ROWS sum hidden units 0..63 on top of the bias, the others units 64..127 from zero and pass their sums
// through `part`; the first half's thread adds them and writes the softmax row (float32 or, in f16-logits mode, float16).
template <int ROWS, typename Args>
__device__ __forceinline__ void head_dense5_softmax(const float* hs, const float* w2s, float* part, const TileDesc* __restrict__ tds,
                                                    const Args& a, int tid)
{
    constexpr int LDH = RD_H + 1;
    const int hrow = tid < ROWS ? tid : tid - ROWS;
    const int j0 = tid < ROWS ? 0 : RD_H / 2;
    float lg[5];
#pragma unroll
    for (int o = 0; o < 5; o++) lg[o] = tid < ROWS ? w2s[RD_H * 5 + o] : 0.f;
    for (int j = j0; j < j0 + RD_H / 2; j++) {
        const float h = hs[hrow * LDH + j];
#pragma unroll
        for (int o = 0; o < 5; o++) lg[o] += h * w2s[j * 5 + o];
    }
    if (tid >= ROWS) {
#pragma unroll
        for (int o = 0; o < 5; o++) part[hrow * 5 + o] = lg[o];
    }
    __syncthreads();
    if (tid < ROWS) {
        const TileDesc sd = tds[tid >> 5];
        const int t = sd.t0 + (tid & 31);
        const int64_t seg_row = sd.seg_row;
        if (t < sd.seg_len) {
#pragma unroll
            for (int o = 0; o < 5; o++) lg[o] += part[tid * 5 + o];
            float mx = lg[0];
#pragma unroll
            for (int o = 1; o < 5; o++) mx = lg[o] > mx ? lg[o] : mx;
            float e[5], sum = 0.f;
#pragma unroll
            for (int o = 0; o < 5; o++) {
                e[o] = expf(lg[o] - mx);
                sum += e[o];
            }
            if (a.probs_f16) {
                _Float16* pr = (_Float16*)a.probs + ((size_t)seg_row + t) * 5;
#pragma unroll
                for (int o = 0; o < 5; o++) pr[o] = (_Float16)(e[o] / sum);   // round to nearest even
            } else {
                float* pr = a.probs + ((size_t)seg_row + t) * 5;
#pragma unroll
                for (int o = 0; o < 5; o++) pr[o] = e[o] / sum;
            }
        }
    }
}

// WM = waves along M (64 time steps each): 2 -> the 256-thread, 128-row tile, two workgroups per CU (the product's shape);
// 4 -> a 512-thread, 256-row tile, one workgroup per CU: the B (weight) tile is shared by twice the rows, so a CU moves
// (256 + 256) x 64 B = 32 KiB per chunk by LDS-DMA instead of 2 x (128 + 256) x 64 B = 48 KiB for the same FLOPs
// (measured, DESIGN.md 4.1: rd_set_conv_shape / tools/conv_shape.py).
//
// FIN (round 4): block 0's second conv with the block's FIRST conv folded in.  That conv has one input channel -- three fmas and a
// ReLU per output -- so instead of a kernel that writes its 1 KiB per time step to HBM (tcn_in_kernel) and a DMA that reads it back
// three times, the A tiles are computed in LDS from the raw samples: per 16-channel slice one region of 4 x (32 + 2 dil) rows (each
// 32-row sub-tile with the 2 dil rows in front of it that the three taps reach back to), written by ds_write_b128 in the chunk
// layout (same 16-B slot swizzle) while the previous slice is being multiplied; the three taps of a slice read it at row offsets
// 0, dil, 2 dil.  Only the weight tiles travel by LDS-DMA.  Values are tcn_in_kernel's bit for bit (conv_in_value), so the layer's
// output is too (tests/test_gpu_forward.py::test_first_conv_fused_is_bit_identical).
constexpr int FIN_DMAX = 2;               // dilations of block 0 the FIN variant is built for (sig2seq.yaml: 1)
template <int NT, int TAPS, int EPI, int WM = 2, bool FIN = false>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void tcn_gemm_kernel(ConvArgs a)
{
    static_assert(!FIN || (EPI == EPI_RES_MATCH && WM == 2 && TAPS == 3 && NT == 4), "FIN: block 0's second conv, product shape");
    constexpr int BM = 64 * WM;           // time steps per workgroup tile (shadows the file-scope 128)
    constexpr int NWAVE = 2 * WM;
    constexpr int BN = 2 * NT * 32;       // output channels per workgroup (2 waves along N)
    constexpr int NCHUNK = TAPS * (RD_C / BK);
    constexpr int A_FLOATS = FIN ? 0 : BM * BK;    // FIN: no A part in the DMA stages
    constexpr int STAGE_FLOATS = A_FLOATS + BN * BK;   // one K-chunk of A and B
    constexpr int HEAD_FLOATS = BM * (RD_H + 1) + RD_H * 5 + 8 + BM * 5;
    constexpr int NSTAGE = 3;
    constexpr int MIDROWS = 4 * (32 + 2 * FIN_DMAX);                      // rows of a slice's A region
    constexpr int FIN_FLOATS = FIN ? 2 * MIDROWS * BK + 4 * RD_C : 0;     // two regions + the first conv's [w0 | w1 | w2 | b] x 256
    constexpr int SMEM_FLOATS = (EPI == EPI_HEAD && HEAD_FLOATS > NSTAGE * STAGE_FLOATS) ? HEAD_FLOATS : NSTAGE * STAGE_FLOATS + FIN_FLOATS;

    __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];  // 3 x 24 KiB (conv, WM = 2) / 3 x 32 KiB (WM = 4) / 67 KiB (head) / 70 KiB (FIN)

    if constexpr (FIN) clear_zero_rows(a.zr);   // (this kernel is then the forward's first)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;   // wave row (64 time steps)
    const int wn = wave & 1;    // wave col (NT*32 channels)
    // tile descriptor (wave-uniform scalar loads): the segment's first global row, this tile's first local time
    // step, the segment length.  A segment is a window, a whole read, or the first rows of a window.
    // A workgroup tile is four independent 32-row sub-tiles (rows 32s .. 32s+31), each with its own descriptor: stream
    // tiles use four consecutive sub-tiles of one segment, short head segments are packed four to a tile.
    const TileDesc* __restrict__ tds = a.tiles + (size_t)blockIdx.x * NWAVE;   // one 32-row sub-tile descriptor per wave
    const TileDesc sdm[2] = {tds[wm * 2], tds[wm * 2 + 1]};           // this wave's two sub-tiles
    const bool mval[2] = {sdm[0].seg_len > sdm[0].t0, sdm[1].seg_len > sdm[1].t0};

    // DMA roles, fixed for the whole tile: a wave-instruction moves 16 rows x 64 B; lane -> (row-in-piece, physical
    // 16-B slot).  Wave w stages sub-tile w (two pieces), so the descriptor fields it needs are wave-uniform scalars.
    const int dma_r = lane >> 2;
    const int dma_ps = lane & 3;
    const TileDesc sst = tds[wave];
    const int64_t d_seg = sst.seg_row, d_alt = sst.alt_row;
    const int d_t0 = sst.t0, d_ain = sst.alt_in;
    const int d_len = sst.seg_len > sst.t0 ? sst.in_len : 0;            // empty sub-tile: zero page only
    const int lane_slot = (dma_ps ^ ((dma_r >> 2) & 3)) * 4;           // (tile row >> 2) & 3 = (dma_r >> 2) & 3
    // This lane's A source row for each tap and piece, fixed for the tile: a row of the input tensor, or the tensor's
    // zero row where the time step lies before the segment (causal padding) or in an empty sub-tile.  Per chunk the
    // address is then one multiply-add on a per-chunk base.
    int arow[TAPS][2];
#pragma unroll
    for (int tap = 0; tap < TAPS; tap++)
#pragma unroll
        for (int pc = 0; pc < 2; pc++) {
            const int t = d_t0 + pc * 16 + dma_r - (TAPS - 1 - tap) * a.dil;
            arow[tap][pc] = (t >= 0 && t < d_len) ? (int)((t < d_ain ? d_seg : d_alt) + t) : a.zero_row;
        }
    const char* lanebase = (const char*)(a.in + lane_slot);
    const unsigned lane_off = lane * 16;

    // one piece (wave-instruction) of a chunk's staging: pieces 0,1 = this wave's two 16-row pieces of A, 2.. = its
    // BN/64 consecutive 1-KiB pieces of B.  `tap` is the chunk's tap (a literal at every call site).
    constexpr int PB = BN / (16 * NWAVE);   // 1-KiB pieces (16 rows) of B per wave
    constexpr int NA = FIN ? 0 : 2;         // A pieces per wave (FIN: the A tiles are computed, not loaded)
    constexpr int NPIECE = NA + PB;
    auto stage_piece = [&](int chunk, int tap, float* st, int pc) {
        // chunk order: input-channel slice outer, tap inner -> the three shifted reads of the same rows are adjacent in time
        if (pc < NA) {
            const int cc = chunk / TAPS;
            const char* src = lanebase + (size_t)cc * (BK * 4) + (uint64_t)(unsigned)arow[tap][pc] * (RD_C * 4);
            glds16_uncounted((const float*)src, st + (wave * 2 + pc) * 256);   // 1 KiB piece = tile rows 16*piece .. 16*piece+15
        } else {
            const float* wb = a.wpk + (size_t)chunk * BN * BK + wave * PB * 256;   // pre-swizzled on the host: linear copy
            float* dst = st + A_FLOATS + wave * PB * 256;
            const int pb = pc - NA;
            if (pb == 0) glds16_uncounted_saddr<0>(lane_off, wb, dst);
            else if (pb == 1) glds16_uncounted_saddr<1024>(lane_off, wb, dst);
            else if (pb == 2) glds16_uncounted_saddr<2048>(lane_off, wb, dst);
            else glds16_uncounted_saddr<3072>(lane_off, wb, dst);
        }
    };
    auto stage = [&](int chunk, int tap, float* st) {
#pragma unroll
        for (int pc = 0; pc < NPIECE; pc++) stage_piece(chunk, tap, st, pc);
    };

    // ---- FIN: the A tiles computed from the raw samples.  Wave w computes sub-tile w's region rows i = 0 .. 31 + 2 dil (time step
    //      d_t0 - 2 dil + i) in three passes of 16 rows; a lane owns row dma_r of a pass and the four channels dma_ps * 4 .. + 3 of the slice.
    float* const midb = smem + NSTAGE * STAGE_FLOATS;          // [2][MIDROWS][16]
    float* const wins = midb + 2 * MIDROWS * BK;               // [4][256]: w0 | w1 | w2 | bias of the first conv
    const int RS = 32 + 2 * a.dil;                             // region rows per sub-tile
    float xs[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // per pass: x[t - 2 dil], x[t - dil], x[t] (0 before the segment)
    bool rowv[3] = {false, false, false};                      // the row exists as conv input (else it is a zero row: causal padding / past the end)
    if constexpr (FIN) {
        *(float4*)(wins + tid * 4) = tid < 192 ? *(const float4*)(a.w_in + tid * 4) : *(const float4*)(a.b_in + (tid - 192) * 4);
        const float* __restrict__ xw = a.x + sst.src_row;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int i = dma_r + 16 * p;
            const int t = d_t0 - 2 * a.dil + i;
            rowv[p] = i < RS && t >= 0 && t < d_len;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int u = t - (2 - j) * a.dil;
                xs[p][j] = (rowv[p] && u >= 0) ? xw[u] : 0.f;
            }
        }
        __syncthreads();
    }
    auto fin_pass = [&](int cc, int p) {        // pass p of slice cc into region cc & 1
        const int i = dma_r + 16 * p;
        if (i < RS) {
            const int c0 = cc * BK + dma_ps * 4;
            const float4 w0 = *(const float4*)(wins + c0), w1 = *(const float4*)(wins + RD_C + c0), w2 = *(const float4*)(wins + 2 * RD_C + c0),
                         bb = *(const float4*)(wins + 3 * RD_C + c0);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowv[p]) {
                v.x = conv_in_value(bb.x, w0.x, w1.x, w2.x, xs[p][0], xs[p][1], xs[p][2]);
                v.y = conv_in_value(bb.y, w0.y, w1.y, w2.y, xs[p][0], xs[p][1], xs[p][2]);
                v.z = conv_in_value(bb.z, w0.z, w1.z, w2.z, xs[p][0], xs[p][1], xs[p][2]);
                v.w = conv_in_value(bb.w, w0.w, w1.w, w2.w, xs[p][0], xs[p][1], xs[p][2]);
            }
            const int rho = wave * RS + i;
            *(float4*)(midb + (cc & 1) * (MIDROWS * BK) + rho * BK + ((dma_ps ^ ((rho >> 2) & 3)) * 4)) = v;
        }
    };
    if constexpr (FIN) {
        // slice 0's region now, BEFORE the first weight DMA goes out: the sample loads above are the compiler's own loads, and its wait
        // for them would also wait for every hand-counted DMA issued in between (visible behind the first chunk's barrier)
#pragma unroll
        for (int p = 0; p < 3; p++) fin_pass(0, p);
    }

    const int fr = lane & 31;
    const int fh = lane >> 5;
    // the accumulators start at the bias of their output channel (a lane holds ONE channel of each N tile): the epilogue
    // then has no add to do before the ReLU
    f32x16 acc[2][NT];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const float bias = a.bias[wn * NT * 32 + n * 32 + fr];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = bias;
    }

    const int swz = (fr >> 2) & 3;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; g++) koff[g] = ((2 * g + fh) ^ swz) * 4;
    const int a_off = (wm * 64 + fr) * BK;
    const int b_off = A_FLOATS + (wn * NT * 32 + fr) * BK;

    // One chunk: multiply the landed stage while the chunk after next is issued into the stage that was read last.  The
    // DMA pieces go out one by one between groups of 2 x NT MFMAs, so their address arithmetic issues in the shadow of
    // the matrix pipe instead of in front of it; the second k-group's fragments are requested after the first group of
    // MFMAs (their LDS latency sits under the other 3 groups).  sched_barrier pins that order.
    const bool work = mval[0] || mval[1];   // (tiles are packed, so a wave with work almost always has both sub-tiles)
    // FIN: this lane's fragment rows in a slice region, per tap and sub-tile: region row = (sub-tile) * RS + fr + tap * dil
    int frow[3][2] = {{0, 0}, {0, 0}, {0, 0}}, fswz[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    if constexpr (FIN) {
#pragma unroll
        for (int tap = 0; tap < 3; tap++)
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int rho = (wm * 2 + m) * RS + fr + tap * a.dil;
                frow[tap][m] = rho * BK;
                fswz[tap][m] = (rho >> 2) & 3;
            }
    }
    // (cur = this chunk's index, cur_tap = its tap: literals at every call site)
    auto chunk_step = [&](const float* st, int cur, int cur_tap, int next, int next_tap, float* nst) {
        // MFMA order: k-group g, then N tile n, then (kr, m): 8 MFMAs per (g, n) step on two accumulators.  A step needs
        // the A fragments of its k-group (8 registers) and ONE B fragment (4): the next step's B fragment and, during the
        // last step of a group, the next group's A fragments are requested before the step's MFMAs, so 24 fragment
        // registers are live instead of 48 -- that keeps the kernel at <= 208 VGPRs, which leaves room for a beam-search
        // wave (96) beside two of these on a SIMD.
        const float* Ab = st + a_off;
        const float* Bb = st + b_off;
        const float* Mb = midb + ((cur / TAPS) & 1) * (MIDROWS * BK);     // FIN: the region of this chunk's slice
        constexpr int NSTEP = (BK / 8) * NT;
        float4 af[2][2], bq[2];
        auto rdA = [&](int g) {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                if constexpr (FIN) af[g & 1][m] = *(const float4*)(Mb + frow[cur_tap][m] + (((2 * g + fh) ^ fswz[cur_tap][m]) * 4));
                else af[g & 1][m] = *(const float4*)(Ab + m * 32 * BK + koff[g]);
            }
        };
        auto rdB = [&](int sidx) { bq[sidx & 1] = *(const float4*)(Bb + (sidx % NT) * 32 * BK + koff[sidx / NT]); };
        const bool st_ok = next < NCHUNK;
        if (work) {
            rdA(0);
            rdB(0);
        }
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; sidx++) {
            const int g = sidx / NT, n = sidx % NT;
            if (work) {
                if (sidx + 1 < NSTEP) rdB(sidx + 1);
                if (n == NT - 1 && g + 1 < BK / 8) rdA(g + 1);
#pragma unroll
                for (int kr = 0; kr < 4; kr++) {
                    const float bv = kr == 0 ? bq[sidx & 1].x : kr == 1 ? bq[sidx & 1].y : kr == 2 ? bq[sidx & 1].z : bq[sidx & 1].w;
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const float4 aq = af[g & 1][m];
                        const float av = kr == 0 ? aq.x : kr == 1 ? aq.y : kr == 2 ? aq.z : aq.w;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
                    }
                }
                // the reads first, then the 8 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st_ok && sidx < NPIECE) stage_piece(next, next_tap, nst, sidx);
            if constexpr (FIN) {
                // pass cur_tap of the NEXT slice's region, behind the weight pieces: its ~20 vector instructions issue in the shadow
                // of the matrix pipe like the DMA's address arithmetic (the region it writes was last read two slices ago)
                if (sidx == NPIECE && cur / TAPS + 1 < RD_C / BK) fin_pass(cur / TAPS + 1, cur_tap);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Three stages of one K-chunk each, one barrier per chunk: while chunk c is multiplied, chunk c+1 is landing and chunk
    // c+2 is issued, so a row that comes from HBM has two chunk times; the wait at the top of a chunk leaves the newest
    // chunk's NPIECE instructions in flight.  (Two workgroups per CU: 2 x 72 KiB of LDS, 2 waves per SIMD.)
    static_assert(NPIECE <= (BK / 8) * NT, "one DMA piece per MFMA step");
    float* st0 = smem;
    float* st1 = smem + STAGE_FLOATS;
    float* st2 = smem + 2 * STAGE_FLOATS;
    constexpr bool T3 = TAPS == 3;   // chunk % 3 is the tap; the loop below advances by the 3 stages
    static_assert(TAPS == 3 || TAPS == 1, "tap of a chunk is a literal in the unrolled loop");
    stage(0, 0, st0);
    stage(1, T3 ? 1 : 0, st1);
    auto step = [&](int c, int tap, const float* st, int tap2, float* nst) {
        if (c + 1 < NCHUNK) wait_dma_and_barrier<NPIECE, FIN>(); else wait_dma_and_barrier<0, FIN>();
        chunk_step(st, c, tap, c + 2, tap2, nst);
    };
    for (int chunk = 0; chunk < NCHUNK; chunk += 3) {
        step(chunk, 0, st0, T3 ? 2 : 0, st2);
        if (chunk + 1 < NCHUNK) step(chunk + 1, T3 ? 1 : 0, st1, 0, st0);
        if (chunk + 2 < NCHUNK) step(chunk + 2, T3 ? 2 : 0, st2, T3 ? 1 : 0, st1);
    }
    __syncthreads();   // every wave is done with the staging LDS: the epilogues reuse it

    // ---------------- epilogue ----------------
    // C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    if constexpr (EPI != EPI_HEAD) {
        // The accumulator layout gives a lane ONE channel of 16 rows: stored directly that is 128 dword stores per lane
        // (store-issue bound, ~10 % of the kernel).  Instead each wave transposes its tile through a private LDS patch,
        // 32 rows x 64 channels at a time, and leaves with 16-B-per-lane accesses: 4 x 256-B row segments per
        // instruction, 32 store (and 32 residual load) instructions per lane instead of 128.
        // patch row stride in floats.  64 = one 256-B bank row per patch row: a ds_read_b128 lane group (16 lanes: two quarter-rows of
        // one patch row + one half-row of the next) then covers 64 distinct banks.  (Round 1's 68 put the second row's quarter on
        // the first row's banks: 1.2-1.5 M conflict cycles per launch.)
        constexpr int TSTR = 64;
        float* ts = smem + wave * (32 * TSTR);
        float4* sink4 = (float4*)a.sink + (threadIdx.x & 255);   // (1024 floats; two lanes of a 512-thread workgroup may share a slot: nobody reads it)
        const int rrow = lane >> 4;                    // 0..3: row inside a 4-row store group
        const int c4 = (lane & 15) * 4;                // channel offset inside the 64-channel patch
#pragma unroll
        for (int m = 0; m < 2; m++) {
            if (!mval[m]) continue;
            const TileDesc sd = sdm[m];
            const int T = sd.seg_len;
            float* __restrict__ outw = a.out + (size_t)sd.seg_row * RD_C;
            const float* __restrict__ resw = a.resid + (size_t)sd.seg_row * RD_C;      // block input (separate tensor)
            const float* __restrict__ resalt = a.resid + (size_t)sd.alt_row * RD_C;
            const bool interior = sd.t0 + 32 <= T;
            // this lane's accumulators of N tiles 2 np, 2 np + 1 (ReLU) -> the wave's LDS patch
            auto to_patch = [&](int np) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int n = 2 * np + j;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
                        ts[rl * TSTR + j * 32 + fr] = relu_raw(acc[m][n][e]);   // bias included since the start
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            // Fast path (every stream tile but a segment's last): the 32 rows are all inside the segment and on one side of the
            // residual switch, so every address is a wave-uniform base + one per-lane offset -- no per-row selects or 64-bit
            // vector adds.  That matters more than it looks: beside the CU partner's MFMA stream a vector instruction of this
            // wave issues about once per MFMA (tools/probe/mfma_valu_probe.hip), so the epilogue's length IS its vector
            // instruction count.  The block input added here is a previous block's output, i.e. already ReLU'd (>= +0): the sum
            // with ReLU(acc) cannot be negative and the second ReLU of keras-tcn's block is the identity on it -- dropped.
            const bool res_one = EPI != EPI_RES_IDENT || sd.t0 + 32 <= sd.alt_res || sd.t0 >= sd.alt_res;
            if (EPI != EPI_RES_MATCH && interior && res_one) {
                const unsigned lbyte = (rrow * RD_C + c4) * 4;
                float* __restrict__ ob = outw + (size_t)sd.t0 * RD_C + wn * (NT * 32);
                const float* __restrict__ rb = (sd.t0 < sd.alt_res ? resw : resalt) + (size_t)sd.t0 * RD_C + wn * (NT * 32);
#pragma unroll
                for (int np = 0; np < NT / 2; np++) {
                    to_patch(np);
#pragma unroll
                    for (int ih = 0; ih < 8; ih += 4) {   // (four rows at a time: 16 residual registers live, not 32)
                        float4 rv[4];
                        if constexpr (EPI == EPI_RES_IDENT) {
#pragma unroll
                            for (int i = 0; i < 4; i++) rv[i] = gload16(rb + np * 64 + (ih + i) * 4 * RD_C, lbyte);
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            float4 v = *(const float4*)(ts + ((ih + i) * 4 + rrow) * TSTR + c4);
                            if constexpr (EPI == EPI_RES_IDENT) {
                                v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
                            }
                            gstore16(ob + np * 64 + (ih + i) * 4 * RD_C, lbyte, v);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                continue;
            }
#pragma unroll
            for (int np = 0; np < NT / 2; np++) {
                to_patch(np);
                // ---- LDS patch -> (residual) -> global, 16 B per lane
                const int ch = wn * NT * 32 + np * 64 + c4;
                float4 wm4 = make_float4(0.f, 0.f, 0.f, 0.f), bm4 = wm4;
                if constexpr (EPI == EPI_RES_MATCH) {
                    wm4 = *(const float4*)(a.wmatch + ch);
                    bm4 = *(const float4*)(a.bmatch + ch);
                }
#pragma unroll
                for (int ih = 0; ih < 8; ih += 4) {   // four rows at a time (registers: see the fast path)
                    float4 rv[4];
                    int tt[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int t = sd.t0 + (ih + i) * 4 + rrow;
                        tt[i] = t;
                        if constexpr (EPI == EPI_RES_IDENT) {
                            const int tc = (interior || t < T) ? t : T - 1;
                            rv[i] = *(const float4*)((tc < sd.alt_res ? resw : resalt) + (size_t)tc * RD_C + ch);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float4 v = *(const float4*)(ts + ((ih + i) * 4 + rrow) * TSTR + c4);
                        const int t = tt[i];
                        const bool inb = interior || t < T;
                        if constexpr (EPI == EPI_RES_IDENT) {
                            v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;   // (>= +0: see the fast path)
                        } else if constexpr (EPI == EPI_RES_MATCH) {
                            const float xv = a.x[(size_t)sd.src_row + (inb ? t : T - 1)];
                            v.x = (bm4.x + xv * wm4.x) + v.x; v.y = (bm4.y + xv * wm4.y) + v.y;
                            v.z = (bm4.z + xv * wm4.z) + v.z; v.w = (bm4.w + xv * wm4.w) + v.w;
                            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                            v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                        }
                        float4* dst = inb ? (float4*)(outw + (size_t)t * RD_C + ch) : sink4;
                        *dst = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        // Dense(128) bias + ReLU into LDS, then Dense(5) + softmax   (model.py:72-75)
        constexpr int LDH = RD_H + 1;
        float* hs = smem;  // [BM][LDH]  (staging buffers are dead after the last barrier)
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int hcol = wn * NT * 32 + n * 32 + fr;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const float v = acc[m][n][e];
                    hs[row * LDH + hcol] = v > 0.f ? v : 0.f;
                }
        }
        float* w2s = smem + BM * LDH;  // [128*5 + 5]
        for (int i = tid; i < RD_H * 5; i += 64 * NWAVE) w2s[i] = a.w2[i];
        if (tid < 5) w2s[RD_H * 5 + tid] = a.b2[tid];
        __syncthreads();
        static_assert(64 * NWAVE == 2 * BM, "head epilogue: two threads per row");
        head_dense5_softmax<BM>(hs, w2s, w2s + RD_H * 5 + 8, tds, a, tid);
    }
}

// Block 0, first conv: C_in = 1 (VALU; memory-bound 1 KiB write per time step), fused bias + ReLU.
// One workgroup per tile descriptor: 4 rows per pass (64 lanes x float4 = one 1 KiB row per wave).
__global__ __launch_bounds__(256) void tcn_in_kernel(const float* __restrict__ x, const float* __restrict__ w /*[3][256]*/,
                                                      const float* __restrict__ b, float* __restrict__ out,
                                                      const TileDesc* __restrict__ tiles, int dil, ZeroRows zr)
{
    clear_zero_rows(zr);
    const TileDesc td = tiles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)];   // wave w owns sub-tile w: one row per pass
    const int c4 = (threadIdx.x & 63) * 4;
    const float4 w0 = *(const float4*)(w + c4), w1 = *(const float4*)(w + 256 + c4), w2 = *(const float4*)(w + 512 + c4);
    const float4 bb = *(const float4*)(b + c4);
    const float* xw = x + td.src_row;
    float* ow = out + (size_t)td.seg_row * RD_C;
    const int tend = td.t0 + 32 < td.seg_len ? td.t0 + 32 : td.seg_len;
    for (int t = td.t0; t < tend; t++) {
        const float x2 = xw[t];
        const float x1 = t - dil >= 0 ? xw[t - dil] : 0.f;
        const float x0 = t - 2 * dil >= 0 ? xw[t - 2 * dil] : 0.f;
        float4 v;
        v.x = conv_in_value(bb.x, w0.x, w1.x, w2.x, x0, x1, x2);
        v.y = conv_in_value(bb.y, w0.y, w1.y, w2.y, x0, x1, x2);
        v.z = conv_in_value(bb.z, w0.z, w1.z, w2.z, x0, x1, x2);
        v.w = conv_in_value(bb.w, w0.w, w1.w, w2.w, x0, x1, x2);
        *(float4*)(ow + (size_t)t * RD_C + c4) = v;
    }
}

// =====================================================================================================================
// Split-f16 ("f16x3") variant of the same kernels: fp32-equivalent arithmetic on the 16x faster f16 matrix pipe.
//
// Every fp32 operand is carried as hi + lo with hi = f16(v), lo = f16(v - hi) (22 significant bits), and a product is
// evaluated as hi*hi + hi*lo + lo*hi in fp32 accumulators on v_mfma_f32_32x32x16_f16 (the dropped lo*lo term is
// < 2^-22 relative).  Weights are scaled by a power of two per tensor before splitting so that their lo parts stay
// normal f16 numbers; the epilogue multiplies the accumulator by the exact inverse.  Measured against a float64
// reference the softmax error is the same ~1e-6..1e-5 as the exact-fp32 path's; bf16x3 is not
// offered because its 16-bit significand pairs exceed the 1e-4 bound on peaky outputs.
//
// Activation rows keep their 1 KiB: channel group g (32 channels) occupies 128 B = [32 hi halves | 32 lo halves], so a
// K-chunk tile row is again 128 B and the LDS-DMA, the XOR swizzle and the tile geometry are those of the fp32 kernel;
// 16-B slot s of a row holds hi k = 8s..8s+7 for s < 4 and lo k = 8(s-4).. for s >= 4.
// =====================================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int ROWH = 2 * RD_C;   // halves per activation row (512)

struct SplitArgs {
    const _Float16* in;
    _Float16* out;
    const _Float16* wpk;   // [chunk][BN][hi 32 | lo 32], slots pre-swizzled
    const float* bias;
    float inv_scale;       // 1 / (power-of-two scale applied to the weights before splitting)
    const _Float16* resid;
    const float* x;
    const float* wmatch;
    const float* bmatch;
    const float* w2;
    const float* b2;
    float* probs;
    int probs_f16;
    int zero_row;
    float* sink;
    const TileDesc* tiles;
    int dil;
};

template <int NT, int TAPS, int EPI>
__global__ __launch_bounds__(256, 2) void tcn_gemm_split_kernel(SplitArgs a)
{
    // Same pipeline as tcn_gemm_kernel: 16-channel chunks (one MFMA k-step), 64-B LDS rows [16 hi | 16 lo], three stages,
    // hand-counted LDS-DMA issued piecewise between the MFMA steps.
    constexpr int BN = 2 * NT * 32;
    constexpr int BKC = 16;                       // input channels per chunk
    constexpr int NCHUNK = TAPS * (RD_C / BKC);
    constexpr int STAGE_FLOATS = (BM + BN) * 16;  // 64 B per row
    constexpr int NSTAGE = 3;
    constexpr int HEAD_FLOATS = BM * (RD_H + 1) + RD_H * 5 + 8 + BM * 5;
    constexpr int SMEM_FLOATS = (EPI == EPI_HEAD && HEAD_FLOATS > NSTAGE * STAGE_FLOATS) ? HEAD_FLOATS : NSTAGE * STAGE_FLOATS;

    __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const TileDesc* __restrict__ tds = a.tiles + (size_t)blockIdx.x * 4;   // four 32-row sub-tiles (see the fp32 kernel)
    const TileDesc sdm[2] = {tds[wm * 2], tds[wm * 2 + 1]};
    const bool mval[2] = {sdm[0].seg_len > sdm[0].t0, sdm[1].seg_len > sdm[1].t0};

    // DMA roles: a wave-instruction moves 16 rows x 64 B; wave w stages sub-tile w (two pieces) and its share of B.
    // LDS slot s of a row: s = 0,1 hi halves of channels 0-7 / 8-15 of the chunk, s = 2,3 their lo halves; in HBM a
    // 32-channel group is 128 B = [32 hi | 32 lo], so the slot's source is group base + (c16 & 1) * 32 B + (s & 1) * 16 B
    // + (s >> 1) * 64 B.
    const int dma_r = lane >> 2;
    const int dma_ps = lane & 3;
    const TileDesc sst = tds[wave];
    const int64_t d_seg = sst.seg_row, d_alt = sst.alt_row;
    const int d_t0 = sst.t0, d_ain = sst.alt_in;
    const int d_len = sst.seg_len > sst.t0 ? sst.in_len : 0;
    const int lslot = dma_ps ^ ((dma_r >> 2) & 3);
    const int lane_half = (lslot & 1) * 8 + (lslot >> 1) * 32;   // offset in halves inside the 128-B group, before (c16 & 1) * 16
    int arow[TAPS][2];                                           // see the fp32 kernel
#pragma unroll
    for (int tap = 0; tap < TAPS; tap++)
#pragma unroll
        for (int pc = 0; pc < 2; pc++) {
            const int t = d_t0 + pc * 16 + dma_r - (TAPS - 1 - tap) * a.dil;
            arow[tap][pc] = (t >= 0 && t < d_len) ? (int)((t < d_ain ? d_seg : d_alt) + t) : a.zero_row;
        }
    const char* lanebase = (const char*)(a.in + lane_half);
    const unsigned lane_off = lane * 16;

    constexpr int PB = BN / 64;
    constexpr int NPIECE = 2 + PB;
    auto stage_piece = [&](int chunk, int tap, float* st, int pc) {
        if (pc < 2) {
            const int cc = chunk / TAPS;
            const char* src = lanebase + (size_t)((cc >> 1) * 128 + (cc & 1) * 32) + (uint64_t)(unsigned)arow[tap][pc] * (ROWH * 2);
            glds16_uncounted((const float*)src, st + (wave * 2 + pc) * 256);
        } else {
            const float* wb = (const float*)(a.wpk + (size_t)chunk * BN * 32) + wave * PB * 256;
            float* dst = st + BM * 16 + wave * PB * 256;
            if (pc == 2) glds16_uncounted_saddr<0>(lane_off, wb, dst);
            else if (pc == 3) glds16_uncounted_saddr<1024>(lane_off, wb, dst);
            else if (pc == 4) glds16_uncounted_saddr<2048>(lane_off, wb, dst);
            else glds16_uncounted_saddr<3072>(lane_off, wb, dst);
        }
    };
    auto stage = [&](int chunk, int tap, float* st) {
#pragma unroll
        for (int pc = 0; pc < NPIECE; pc++) stage_piece(chunk, tap, st, pc);
    };

    const int fr = lane & 31;
    const int fh = lane >> 5;
    // accumulators start at bias / inv_scale (inv_scale is a power of two: exact), see the fp32 kernel
    f32x16 acc[2][NT];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const float bias = a.bias[wn * NT * 32 + n * 32 + fr] / a.inv_scale;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = bias;
    }

    const int swz = (fr >> 2) & 3;
    const int oh = (fh ^ swz) * 8;         // halves: hi slot of this lane's k = 8 fh .. 8 fh + 7
    const int ol = ((2 + fh) ^ swz) * 8;   // lo slot
    const int a_off = (wm * 64 + fr) * 32;                       // halves
    const int b_off = BM * 32 + (wn * NT * 32 + fr) * 32;

    const bool work = mval[0] || mval[1];
    auto chunk_step = [&](const float* stf, int next, int next_tap, float* nst) {
        const _Float16* st = (const _Float16*)stf;
        const _Float16* Ab = st + a_off;
        const _Float16* Bb = st + b_off;
        f16x8 ah[2], al[2], bh[2], bl[2];
        const bool st_ok = next < NCHUNK;
        if (work) {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                ah[m] = *(const f16x8*)(Ab + m * 32 * 32 + oh);
                al[m] = *(const f16x8*)(Ab + m * 32 * 32 + ol);
            }
            bh[0] = *(const f16x8*)(Bb + oh);
            bl[0] = *(const f16x8*)(Bb + ol);
        }
        constexpr int PPS = (NPIECE + NT - 1) / NT;   // DMA pieces issued after each N step
#pragma unroll
        for (int n = 0; n < NT; n++) {
            if (work) {
                if (n + 1 < NT) {
                    bh[(n + 1) & 1] = *(const f16x8*)(Bb + (n + 1) * 32 * 32 + oh);
                    bl[(n + 1) & 1] = *(const f16x8*)(Bb + (n + 1) * 32 * 32 + ol);
                }
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n & 1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n & 1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n & 1], acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PPS; q++)
                if (st_ok && n * PPS + q < NPIECE) stage_piece(next, next_tap, nst, n * PPS + q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float* st0 = smem;
    float* st1 = smem + STAGE_FLOATS;
    float* st2 = smem + 2 * STAGE_FLOATS;
    constexpr bool T3 = TAPS == 3;
    static_assert(TAPS == 3 || TAPS == 1, "tap of a chunk is a literal in the unrolled loop");
    stage(0, 0, st0);
    stage(1, T3 ? 1 : 0, st1);
    auto step = [&](int c, const float* st, int tap2, float* nst) {
        if (c + 1 < NCHUNK) wait_dma_and_barrier<NPIECE>(); else wait_dma_and_barrier<0>();
        chunk_step(st, c + 2, tap2, nst);
    };
    for (int chunk = 0; chunk < NCHUNK; chunk += 3) {
        step(chunk, st0, T3 ? 2 : 0, st2);
        if (chunk + 1 < NCHUNK) step(chunk + 1, st1, 0, st0);
        if (chunk + 2 < NCHUNK) step(chunk + 2, st2, T3 ? 1 : 0, st1);
    }
    __syncthreads();

    if constexpr (EPI != EPI_HEAD) {
        constexpr int TSTR = 64;   // see the fp32 kernel
        float* ts = smem + wave * (32 * TSTR);
        f16x4* sinkh = (f16x4*)a.sink + 2 * threadIdx.x;
        const int rrow = lane >> 4;
        const int c4 = (lane & 15) * 4;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            if (!mval[m]) continue;
            const TileDesc sd = sdm[m];
            const int T = sd.seg_len;
            _Float16* __restrict__ outw = a.out + (size_t)sd.seg_row * ROWH;
            const _Float16* __restrict__ resw = a.resid + (size_t)sd.seg_row * ROWH;
            const _Float16* __restrict__ resalt = a.resid + (size_t)sd.alt_row * ROWH;
            const bool interior = sd.t0 + 32 <= T;
#pragma unroll
            for (int np = 0; np < NT / 2; np++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int n = 2 * np + j;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
                        const float v = acc[m][n][e] * a.inv_scale;
                        ts[rl * TSTR + j * 32 + fr] = v > 0.f ? v : 0.f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ch = wn * NT * 32 + np * 64 + c4;                // first of this lane's 4 channels
                const int hoff = (ch >> 5) * 64 + (ch & 31);               // halves offset of their hi parts in a row
                float4 wm4 = make_float4(0.f, 0.f, 0.f, 0.f), bm4 = wm4;
                if constexpr (EPI == EPI_RES_MATCH) {
                    wm4 = *(const float4*)(a.wmatch + ch);
                    bm4 = *(const float4*)(a.bmatch + ch);
                }
                f16x4 rh[8], rl4[8];
                int tt[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int t = sd.t0 + i * 4 + rrow;
                    tt[i] = t;
                    if constexpr (EPI == EPI_RES_IDENT) {
                        const int tc = (interior || t < T) ? t : T - 1;
                        const _Float16* rb = (tc < sd.alt_res ? resw : resalt) + (size_t)tc * ROWH + hoff;
                        rh[i] = *(const f16x4*)rb;
                        rl4[i] = *(const f16x4*)(rb + 32);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 p = *(const float4*)(ts + (i * 4 + rrow) * TSTR + c4);
                    float v[4] = {p.x, p.y, p.z, p.w};
                    const int t = tt[i];
                    const bool inb = interior || t < T;
                    if constexpr (EPI == EPI_RES_IDENT) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v[q] += (float)rh[i][q] + (float)rl4[i][q];
                            v[q] = v[q] > 0.f ? v[q] : 0.f;
                        }
                    } else if constexpr (EPI == EPI_RES_MATCH) {
                        const float xv = a.x[(size_t)sd.src_row + (inb ? t : T - 1)];
                        const float wq[4] = {wm4.x, wm4.y, wm4.z, wm4.w}, bq[4] = {bm4.x, bm4.y, bm4.z, bm4.w};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v[q] = (bq[q] + xv * wq[q]) + v[q];
                            v[q] = v[q] > 0.f ? v[q] : 0.f;
                        }
                    }
                    f16x4 hi, lo;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        hi[q] = (_Float16)v[q];
                        lo[q] = (_Float16)(v[q] - (float)hi[q]);
                    }
                    f16x4* dh = inb ? (f16x4*)(outw + (size_t)t * ROWH + hoff) : sinkh;
                    f16x4* dl = inb ? (f16x4*)(outw + (size_t)t * ROWH + hoff + 32) : sinkh + 1;
                    *dh = hi;
                    *dl = lo;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        constexpr int LDH = RD_H + 1;
        float* hs = smem;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int hcol = wn * NT * 32 + n * 32 + fr;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const float v = acc[m][n][e] * a.inv_scale;
                    hs[row * LDH + hcol] = v > 0.f ? v : 0.f;
                }
        }
        float* w2s = smem + BM * LDH;
        for (int i = tid; i < RD_H * 5; i += 256) w2s[i] = a.w2[i];
        if (tid < 5) w2s[RD_H * 5 + tid] = a.b2[tid];
        __syncthreads();
        head_dense5_softmax<BM>(hs, w2s, w2s + RD_H * 5 + 8, tds, a, tid);
    }
}

// Block 0, first conv (C_in = 1) writing split-f16 rows.
__global__ __launch_bounds__(256) void tcn_in_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            _Float16* __restrict__ out, const TileDesc* __restrict__ tiles, int dil, ZeroRows zr)
{
    clear_zero_rows(zr);
    const TileDesc td = tiles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)];   // wave w owns sub-tile w
    const int c4 = (threadIdx.x & 63) * 4;
    const float4 w0 = *(const float4*)(w + c4), w1 = *(const float4*)(w + 256 + c4), w2 = *(const float4*)(w + 512 + c4);
    const float4 bb = *(const float4*)(b + c4);
    const float* xw = x + td.src_row;
    _Float16* ow = out + (size_t)td.seg_row * ROWH;
    const int hoff = (c4 >> 5) * 64 + (c4 & 31);
    const int tend = td.t0 + 32 < td.seg_len ? td.t0 + 32 : td.seg_len;
    for (int t = td.t0; t < tend; t++) {
        const float x2 = xw[t];
        const float x1 = t - dil >= 0 ? xw[t - dil] : 0.f;
        const float x0 = t - 2 * dil >= 0 ? xw[t - 2 * dil] : 0.f;
        float v[4];
        v[0] = bb.x + x0 * w0.x + x1 * w1.x + x2 * w2.x;
        v[1] = bb.y + x0 * w0.y + x1 * w1.y + x2 * w2.y;
        v[2] = bb.z + x0 * w0.z + x1 * w1.z + x2 * w2.z;
        v[3] = bb.w + x0 * w0.w + x1 * w1.w + x2 * w2.w;
        f16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float r = v[q] > 0.f ? v[q] : 0.f;
            hi[q] = (_Float16)r;
            lo[q] = (_Float16)(r - (float)hi[q]);
        }
        *(f16x4*)(ow + (size_t)t * ROWH + hoff) = hi;
        *(f16x4*)(ow + (size_t)t * ROWH + hoff + 32) = lo;
    }
}

// =====================================================================================================================
// Three-term bf16 split ("bf16x3"): the fp32 contraction on the bf16 matrix pipe with EVERY operand bit kept.
//
// v = hi + mid + lo with hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), round-to-nearest: three 8-bit
// significands with fp32's exponent range, so the three terms reconstruct every finite fp32 value exactly (as long as lo
// is not below bf16's smallest subnormal: |v| >= 2^-110; below that the defect is < 2^-133 absolute).  A product a*b is
// evaluated as the six cross terms of order 2^0, 2^-8 and 2^-16
//      ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped terms (am*bl + al*bm + al*bl) are < 2^-23 |a*b|, the
// size of ONE fp32 rounding of the product.  Six MFMAs at 16x the fp32-MFMA rate: peak 2516 / 6 = 419 TFLOP/s-equivalent.
//
// Activation rows are 16 groups of 16 channels, each group 96 B = [16 hi | 16 mid | 16 lo]: a K-chunk (16 channels, one
// MFMA k-step) of a row is one contiguous 96-B run in HBM and six 16-B slots (term * 2 + k-half) of a 96-B LDS row.  The
// slots of LDS row r sit at physical slot s ^ ((r >> 3) & 1): rows r and r + 8, whose 96-B stride puts the same slot on
// the same banks, swap neighbours, which makes every ds_read_b128 group (16 lanes, 16 distinct r mod 16) conflict-free.
// Two 36-KiB stages (A 128 x 96 B + B 256 x 96 B), two workgroups per CU; chunk c+1 is copied by LDS-DMA (pieces issued
// between the MFMA steps) while chunk c is multiplied.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ROW3 = 3 * RD_C;   // bf16 elements per activation row (768 = 1536 B)

// fp32 -> (hi, mid, lo); exact: hi + mid + lo == v
__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo)
{
    hi = (__bf16)v;
    if (__builtin_isinf((float)hi)) {   // |v| within half a bf16 ulp of 2^128 rounds to infinity: truncate instead (v itself infinite: stays)
        const unsigned u = __float_as_uint(v) & 0xffff0000u;
        const float tv = __uint_as_float(u);
        hi = __builtin_isinf(v) ? hi : (__bf16)tv;
    }
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

struct Bf3Args {
    const __bf16* in;
    __bf16* out;
    const __bf16* wpk;     // [chunk][BN][6 slots x 8], slots pre-swizzled: the exact LDS image
    const float* bias;
    const __bf16* resid;
    const float* x;
    const float* wmatch;
    const float* bmatch;
    const float* w2;
    const float* b2;
    float* probs;
    int probs_f16;
    int zero_row;
    float* sink;
    const TileDesc* tiles;
    int dil;
};

// One 512-thread workgroup per CU: 8 waves as 4 (time) x 2 (channels), each 64 x 128 as in the other kernels, so a workgroup
// tile is 256 time steps x all 256 output channels = EIGHT 32-row sub-tile descriptors.  Against two 256-thread workgroups
// per CU this halves the weight traffic into the CU (the B tile is shared by twice the rows) -- the split kernels are
// bound by the L2 -> LDS copy volume, not by the matrix pipe -- and frees LDS for a third stage:
// 3 x (A 256 x 96 B + B 256 x 96 B) = 144 KiB.
constexpr int BM3 = 256;

template <int NT, int TAPS, int EPI>
__global__ __launch_bounds__(512, 2) void tcn_gemm_bf3_kernel(Bf3Args a)
{
    constexpr int BN = 2 * NT * 32;
    constexpr int NCHUNK = TAPS * (RD_C / 16);
    constexpr int RB = 96;                                   // bytes per LDS row (16 channels x 3 terms x 2 B)
    constexpr int STAGE_BYTES = (BM3 + BN) * RB;             // 48 KiB (conv) / 36 KiB (head)
    constexpr int NSTAGE = 3;
    constexpr int HEAD_FLOATS = BM3 * (RD_H + 1) + RD_H * 5 + 8 + BM3 * 5;
    constexpr int EPI_FLOATS = 8 * 32 * 64;
    constexpr int STG_FLOATS = NSTAGE * STAGE_BYTES / 4;
    constexpr int SMEM_FLOATS = (EPI == EPI_HEAD && HEAD_FLOATS > STG_FLOATS) ? HEAD_FLOATS : (STG_FLOATS > EPI_FLOATS ? STG_FLOATS : EPI_FLOATS);

    __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;   // 0..3: 64 time steps each
    const int wn = wave & 1;
    const TileDesc* __restrict__ tds = a.tiles + (size_t)blockIdx.x * 8;   // eight 32-row sub-tiles (see the fp32 kernel)
    const TileDesc sdm[2] = {tds[wm * 2], tds[wm * 2 + 1]};
    const bool mval[2] = {sdm[0].seg_len > sdm[0].t0, sdm[1].seg_len > sdm[1].t0};

    // ---- DMA roles.  Wave w stages sub-tile w: 32 rows x 6 slots = 192 slots = 3 wave-instructions; slot g = 64 p + lane
    // of the sub-tile is (row g / 6, physical slot g % 6).  B: the wave's BN / 8 rows are a linear copy of the packed weights.
    const TileDesc sst = tds[wave];
    const int64_t d_seg = sst.seg_row, d_alt = sst.alt_row;
    const int d_t0 = sst.t0, d_ain = sst.alt_in;
    const int d_len = sst.seg_len > sst.t0 ? sst.in_len : 0;
    int arow[TAPS][3];        // source row of this lane's slot, per tap and piece
    int aoff[3];              // byte offset of the slot inside the row's 96-B chunk
#pragma unroll
    for (int pc = 0; pc < 3; pc++) {
        const int g = pc * 64 + lane;
        const int r = g / 6, ps = g - 6 * r;
        aoff[pc] = (ps ^ ((r >> 3) & 1)) * 16;
#pragma unroll
        for (int tap = 0; tap < TAPS; tap++) {
            const int t = d_t0 + r - (TAPS - 1 - tap) * a.dil;
            arow[tap][pc] = (t >= 0 && t < d_len) ? (int)((t < d_ain ? d_seg : d_alt) + t) : a.zero_row;
        }
    }
    const char* inb = (const char*)a.in;
    const unsigned lane_off = lane * 16;

    constexpr int PBX2 = BN * 6 / 64 / 4;   // B pieces per PAIR of waves per chunk: 6 (conv) / 3 (head)
    static_assert(BN * RB % (8 * 512) == 0, "B rows split evenly over 8 waves in 512-B halves");
    // B share of a wave: BN * 96 / 8 bytes = 3072 (conv: 3 pieces) / 1536 (head: 1.5 pieces -> waves pair up: even waves 2 pieces, odd 1)
    constexpr int BSH = BN * RB / 8;        // bytes per wave
    constexpr int PBW = (BSH + 1023) / 1024;                   // pieces a wave may issue: 3 (conv) / 2 (head)
    constexpr int NPIECE = 3 + PBW;
    auto stage_piece = [&](int chunk, int tap, char* st, int pc) {
        if (pc < 3) {
            const int cc = chunk / TAPS;
            const char* src = inb + (uint64_t)(unsigned)arow[tap][pc] * (ROW3 * 2) + cc * RB + aoff[pc];
            glds16_uncounted((const float*)src, (float*)(st + (wave * 3 + pc) * 1024));
        } else {
            const int q = pc - 3;
            if constexpr (BSH % 1024 == 0) {
                const char* wb = (const char*)a.wpk + (size_t)chunk * BN * RB + (size_t)wave * BSH;
                float* dst = (float*)(st + BM3 * RB + wave * BSH);
                if (q == 0) glds16_uncounted_saddr<0>(lane_off, wb, dst);
                else if (q == 1) glds16_uncounted_saddr<1024>(lane_off, wb, dst);
                else glds16_uncounted_saddr<2048>(lane_off, wb, dst);
            } else {
                // head: 12 pieces over 8 waves: wave w takes piece w, waves 0..3 also piece 8 + w (always issued by every
                // wave so that the hand-counted vmcnt is uniform: waves 4..7 re-copy piece w: same bytes, same place)
                const int piece = q == 0 ? wave : 8 + (wave & 3);
                const char* wb = (const char*)a.wpk + (size_t)chunk * BN * RB + (size_t)piece * 1024;
                float* dst = (float*)(st + BM3 * RB + piece * 1024);
                glds16_uncounted_saddr<0>(lane_off, wb, dst);
            }
        }
    };
    (void)PBX2;

    const int fr = lane & 31;
    const int fh = lane >> 5;
    f32x16 acc[2][NT];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const float bias = a.bias[wn * NT * 32 + n * 32 + fr];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = bias;
    }

    // fragment addresses: row (.. + fr), term t -> logical slot 2 t + fh -> physical slot ^ ((row >> 3) & 1)
    const int sw = (fr >> 3) & 1;
    int toff[3];
#pragma unroll
    for (int t = 0; t < 3; t++) toff[t] = ((2 * t + fh) ^ sw) * 16;
    const int a_off = (wm * 64 + fr) * RB;
    const int b_off = BM3 * RB + (wn * NT * 32 + fr) * RB;

    const bool work = mval[0] || mval[1];
    auto chunk_step = [&](const char* st, int next, int next_tap, char* nst) {
        const char* Ab = st + a_off;
        const char* Bb = st + b_off;
        bf16x8 af[2][3], bf[2][3];
        const bool st_ok = next < NCHUNK;
        if (work) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int t = 0; t < 3; t++) af[m][t] = *(const bf16x8*)(Ab + m * 32 * RB + toff[t]);
#pragma unroll
            for (int t = 0; t < 3; t++) bf[0][t] = *(const bf16x8*)(Bb + toff[t]);
        }
        constexpr int PPS = (NPIECE + NT - 1) / NT;   // DMA pieces issued after each N step
#pragma unroll
        for (int n = 0; n < NT; n++) {
            if (work) {
                if (n + 1 < NT) {
#pragma unroll
                    for (int t = 0; t < 3; t++) bf[(n + 1) & 1][t] = *(const bf16x8*)(Bb + (n + 1) * 32 * RB + toff[t]);
                }
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    // smallest terms first: 2^-16 order, 2^-8 order, then hi*hi
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][2], bf[n & 1][0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[n & 1][2], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bf[n & 1][1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bf[n & 1][0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[n & 1][1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[n & 1][0], acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PPS; q++)
                if (st_ok && n * PPS + q < NPIECE) {
                    stage_piece(next, next_tap, nst, n * PPS + q);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage = [&](int chunk, int tap, char* st) {
#pragma unroll
        for (int pc = 0; pc < NPIECE; pc++) stage_piece(chunk, tap, st, pc);
    };

    // Three stages, one barrier per chunk (see the fp32 kernel): while chunk c is multiplied, c+1 is landing and c+2 is issued;
    // the wait at the top of a chunk leaves the newest chunk's NPIECE instructions of this wave in flight.
    char* st0 = (char*)smem;
    char* st1 = (char*)smem + STAGE_BYTES;
    char* st2 = (char*)smem + 2 * STAGE_BYTES;
    static_assert(TAPS == 3 || TAPS == 1, "tap of a chunk is a literal in the unrolled loop");
    constexpr bool T3 = TAPS == 3;
    stage(0, 0, st0);
    stage(1, T3 ? 1 : 0, st1);
    constexpr int NWAIT = NPIECE;
    auto step = [&](int c, const char* st, int tap2, char* nst) {
        if (c + 1 < NCHUNK && c >= 2) wait_dma_and_barrier<NWAIT>(); else if (c + 1 < NCHUNK) wait_dma_and_barrier<(NWAIT < NPIECE ? 0 : NPIECE)>(); else wait_dma_and_barrier<0>();
        chunk_step(st, c + 2, tap2, nst);
    };
    for (int chunk = 0; chunk < NCHUNK; chunk += 3) {
        step(chunk, st0, T3 ? 2 : 0, st2);
        if (chunk + 1 < NCHUNK) step(chunk + 1, st1, 0, st0);
        if (chunk + 2 < NCHUNK) step(chunk + 2, st2, T3 ? 1 : 0, st1);
    }
    __syncthreads();

    if constexpr (EPI != EPI_HEAD) {
        constexpr int TSTR = 64;   // see the fp32 kernel
        float* ts = smem + wave * (32 * TSTR);
        bf16x4* sinkh = (bf16x4*)a.sink + (threadIdx.x & 255);   // past-the-end rows store here (nobody reads it)
        const int rrow = lane >> 4;
        const int c4 = (lane & 15) * 4;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            if (!mval[m]) continue;
            const TileDesc sd = sdm[m];
            const int T = sd.seg_len;
            __bf16* __restrict__ outw = a.out + (size_t)sd.seg_row * ROW3;
            const __bf16* __restrict__ resw = a.resid + (size_t)sd.seg_row * ROW3;
            const __bf16* __restrict__ resalt = a.resid + (size_t)sd.alt_row * ROW3;
            const bool interior = sd.t0 + 32 <= T;
#pragma unroll
            for (int np = 0; np < NT / 2; np++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int n = 2 * np + j;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
                        const float v = acc[m][n][e];
                        ts[rl * TSTR + j * 32 + fr] = v > 0.f ? v : 0.f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ch = wn * NT * 32 + np * 64 + c4;                // first of this lane's 4 channels
                const int hoff = (ch >> 4) * 48 + (ch & 15);               // element offset of their hi terms in a row
                float4 wm4 = make_float4(0.f, 0.f, 0.f, 0.f), bm4 = wm4;
                if constexpr (EPI == EPI_RES_MATCH) {
                    wm4 = *(const float4*)(a.wmatch + ch);
                    bm4 = *(const float4*)(a.bmatch + ch);
                }
                bf16x4 r3[8][3];
                int tt[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int t = sd.t0 + i * 4 + rrow;
                    tt[i] = t;
                    if constexpr (EPI == EPI_RES_IDENT) {
                        const int tc = (interior || t < T) ? t : T - 1;
                        const __bf16* rb = (tc < sd.alt_res ? resw : resalt) + (size_t)tc * ROW3 + hoff;
#pragma unroll
                        for (int q = 0; q < 3; q++) r3[i][q] = *(const bf16x4*)(rb + 16 * q);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 p = *(const float4*)(ts + (i * 4 + rrow) * TSTR + c4);
                    float v[4] = {p.x, p.y, p.z, p.w};
                    const int t = tt[i];
                    const bool inb = interior || t < T;
                    if constexpr (EPI == EPI_RES_IDENT) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            // hi + (mid + lo) is exact in fp32: mid + lo is the 16-bit remainder v - hi, and the sum is v
                            v[q] += (float)r3[i][0][q] + ((float)r3[i][1][q] + (float)r3[i][2][q]);
                            v[q] = v[q] > 0.f ? v[q] : 0.f;
                        }
                    } else if constexpr (EPI == EPI_RES_MATCH) {
                        const float xv = a.x[(size_t)sd.src_row + (inb ? t : T - 1)];
                        const float wq[4] = {wm4.x, wm4.y, wm4.z, wm4.w}, bq[4] = {bm4.x, bm4.y, bm4.z, bm4.w};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v[q] = (bq[q] + xv * wq[q]) + v[q];
                            v[q] = v[q] > 0.f ? v[q] : 0.f;
                        }
                    }
                    bf16x4 o3[3];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        __bf16 h, md, l;
                        split3(v[q], h, md, l);
                        o3[0][q] = h;
                        o3[1][q] = md;
                        o3[2][q] = l;
                    }
                    __bf16* ob = outw + (size_t)t * ROW3 + hoff;
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        bf16x4* d = inb ? (bf16x4*)(ob + 16 * q) : sinkh;
                        *d = o3[q];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        constexpr int LDH = RD_H + 1;
        float* hs = smem;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int hcol = wn * NT * 32 + n * 32 + fr;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const float v = acc[m][n][e];
                    hs[row * LDH + hcol] = v > 0.f ? v : 0.f;
                }
        }
        float* w2s = smem + BM3 * LDH;
        for (int i = tid; i < RD_H * 5; i += 512) w2s[i] = a.w2[i];
        if (tid < 5) w2s[RD_H * 5 + tid] = a.b2[tid];
        __syncthreads();
        head_dense5_softmax<BM3>(hs, w2s, w2s + RD_H * 5 + 8, tds, a, tid);
    }
}

// Block 0, first conv (C_in = 1) writing three-term bf16 rows.
__global__ __launch_bounds__(256) void tcn_in_bf3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          __bf16* __restrict__ out, const TileDesc* __restrict__ tiles, int dil, ZeroRows zr)
{
    clear_zero_rows(zr);
    const TileDesc td = tiles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)];   // wave w owns sub-tile w
    const int c4 = (threadIdx.x & 63) * 4;
    const float4 w0 = *(const float4*)(w + c4), w1 = *(const float4*)(w + 256 + c4), w2 = *(const float4*)(w + 512 + c4);
    const float4 bb = *(const float4*)(b + c4);
    const float* xw = x + td.src_row;
    __bf16* ow = out + (size_t)td.seg_row * ROW3;
    const int hoff = (c4 >> 4) * 48 + (c4 & 15);
    const int tend = td.t0 + 32 < td.seg_len ? td.t0 + 32 : td.seg_len;
    for (int t = td.t0; t < tend; t++) {
        const float x2 = xw[t];
        const float x1 = t - dil >= 0 ? xw[t - dil] : 0.f;
        const float x0 = t - 2 * dil >= 0 ? xw[t - 2 * dil] : 0.f;
        float v[4];
        v[0] = bb.x + x0 * w0.x + x1 * w1.x + x2 * w2.x;
        v[1] = bb.y + x0 * w0.y + x1 * w1.y + x2 * w2.y;
        v[2] = bb.z + x0 * w0.z + x1 * w1.z + x2 * w2.z;
        v[3] = bb.w + x0 * w0.w + x1 * w1.w + x2 * w2.w;
        bf16x4 o3[3];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float r = v[q] > 0.f ? v[q] : 0.f;
            __bf16 h, md, l;
            split3(r, h, md, l);
            o3[0][q] = h;
            o3[1][q] = md;
            o3[2][q] = l;
        }
#pragma unroll
        for (int q = 0; q < 3; q++) *(bf16x4*)(ow + (size_t)t * ROW3 + hoff + 16 * q) = o3[q];
    }
}

// test hook: split an fp32 array on the device, [n] -> [3][n] bf16 bit patterns
__global__ void split3_kernel(const float* __restrict__ v, uint16_t* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __bf16 h, m, l;
    split3(v[i], h, m, l);
    out[i] = *(const uint16_t*)&h;
    out[n + i] = *(const uint16_t*)&m;
    out[2 * n + i] = *(const uint16_t*)&l;
}

int timer_begin(hipStream_t st, KernelTimer& tm)
{
    if (tm.enabled && tm.used < tm.starts.size()) RD_HIP(hipEventRecord(tm.starts[tm.used], st));
    return RD_OK;
}
int timer_end(hipStream_t st, KernelTimer& tm, double flops, double bytes, int tag = 0)
{
    if (tm.enabled && tm.used < tm.starts.size()) {
        RD_HIP(hipEventRecord(tm.stops[tm.used], st));
        tm.used++;
        tm.flops += flops;
        tm.bytes += bytes;
        tm.each_flops.push_back(flops);
        tm.each_tag.push_back(tag);
    }
    return RD_OK;
}

}  // namespace

// Forward over a set of independent SEGMENTS packed in one row space: d_signal [*] fp32 (already MAD-normalised),
// per-layer tile lists -> d_probs [total_rows][5].
// A segment is causally zero-padded at its own start; nothing leaks between segments (heads read their later rows
// from their read's stream, see TileDesc).  Three activation tensors: X_a / X_b (block input / output, ping-pong) and
// MID.
namespace {

// block 0's first conv folded into its second (tcn_gemm_kernel<..., FIN>): exact-fp32 mode, product workgroup shape, dilation within what
// the variant's LDS regions hold; rd_set_conv_fuse(ctx, 0) turns it off (A/B runs, the bit-identity test)
bool fuse_first_conv(const rd_ctx* ctx)
{
    return ctx->conv_fuse && ctx->precision == 0 && ctx->conv_shape == 0 && ctx->model.nblocks >= 1 && ctx->model.dil[0] >= 1 &&
           ctx->model.dil[0] <= FIN_DMAX;
}

int launch_layer(rd_ctx* ctx, hipStream_t st, int b, int kind /*0 in, 1 conv0, 2 conv1, 3 head*/, const TileDesc* tiles, int n,
                double rows, int zero_row, const float* d_signal, float* Xin, float* Xout, float* MID, float* d_probs, int probs_f16)
{
    if (n <= 0) return RD_OK;
    Model& m = ctx->model;
    const bool split = ctx->precision == 1;
    const bool bf3 = ctx->precision == 2;
    // the 512-thread kernels (bf16x3; fp32 under rd_set_conv_shape 1) own TWO 128-row tiles per workgroup and are launched over n / 2:
    // every tile list is padded to eight sub-tiles (plan_pad_tiles, rd_uniform_tiles) -- checked here, not assumed
    if (kind != 0 && (bf3 || (!split && ctx->conv_shape == 1)))
        RD_REQUIRE(n % 2 == 0, "internal: %d tiles in a layer of 256-row workgroups (lists are padded to pairs)", n);
    int rc;
    ZeroRows zr = {};
    const bool fin = kind == 2 && b == 0 && fuse_first_conv(ctx);
    if (kind == 0 || fin) {
        const size_t row_bytes = (size_t)RD_C * (bf3 ? 6 : 4);
        zr.n16 = (int)(row_bytes / 16);
        zr.row[0] = (uint4*)((char*)Xin + (size_t)zero_row * row_bytes);
        zr.row[1] = (uint4*)((char*)Xout + (size_t)zero_row * row_bytes);
        zr.row[2] = (uint4*)((char*)MID + (size_t)zero_row * row_bytes);
    }
    if (kind == 0 && fuse_first_conv(ctx)) return RD_OK;   // (computed on the fly by block 0's second conv)
    if (bf3) {
        Bf3Args h = {};
        h.zero_row = zero_row;
        h.sink = m.sink;
        h.tiles = tiles;
        if (kind == 0) {
            if ((rc = timer_begin(st, ctx->timer_in))) return rc;
            hipLaunchKernelGGL(tcn_in_bf3_kernel, dim3(n), dim3(256), 0, st, d_signal, m.w_in, m.b_in, (__bf16*)MID, tiles, m.dil[0], zr);
            RD_HIP(hipGetLastError());
            return timer_end(st, ctx->timer_in, 2.0 * rows * RD_C * RD_K, rows * (RD_C * 6.0 + 4.0));
        }
        if (kind == 3) {
            if ((rc = timer_begin(st, ctx->timer_head))) return rc;
            h.in = (const __bf16*)Xin;
            h.wpk = (const __bf16*)m.w3_d1;
            h.bias = m.b_d1;
            h.w2 = m.w_d2;
            h.b2 = m.b_d2;
            h.probs = d_probs;
            h.probs_f16 = probs_f16;
            hipLaunchKernelGGL((tcn_gemm_bf3_kernel<2, 1, EPI_HEAD>), dim3(n / 2), dim3(512), 0, st, h);
            RD_HIP(hipGetLastError());
            return timer_end(st, ctx->timer_head, 2.0 * rows * (RD_C * RD_H + RD_H * 5), rows * (RD_C * 6.0 + 20.0));
        }
        const int wi = 2 * b + (kind == 2 ? 1 : 0);
        h.dil = m.dil[b];
        h.wpk = (const __bf16*)m.w3_conv[wi];
        h.bias = m.b_conv[wi];
        if (kind == 1) {
            h.in = (const __bf16*)Xin;
            h.out = (__bf16*)MID;
        } else {
            h.in = (const __bf16*)MID;
            h.out = (__bf16*)Xout;
            h.resid = (const __bf16*)Xin;
            h.x = d_signal;
            h.wmatch = m.w_match;
            h.bmatch = m.b_match;
        }
        if ((rc = timer_begin(st, ctx->timer_conv))) return rc;
        if (kind == 1) hipLaunchKernelGGL((tcn_gemm_bf3_kernel<4, 3, EPI_RELU>), dim3(n / 2), dim3(512), 0, st, h);
        else if (b == 0) hipLaunchKernelGGL((tcn_gemm_bf3_kernel<4, 3, EPI_RES_MATCH>), dim3(n / 2), dim3(512), 0, st, h);
        else hipLaunchKernelGGL((tcn_gemm_bf3_kernel<4, 3, EPI_RES_IDENT>), dim3(n / 2), dim3(512), 0, st, h);
        RD_HIP(hipGetLastError());
        return timer_end(st, ctx->timer_conv, 2.0 * rows * RD_C * RD_C * RD_K, (kind == 2 && b > 0 ? 3.0 : 2.0) * rows * RD_C * 6.0, kind == 1 ? 0 : b == 0 ? 2 : 1);
    }
    if (kind == 0) {
        if ((rc = timer_begin(st, ctx->timer_in))) return rc;
        if (split)
            hipLaunchKernelGGL(tcn_in_split_kernel, dim3(n), dim3(256), 0, st, d_signal, m.w_in, m.b_in, (_Float16*)MID, tiles, m.dil[0], zr);
        else
            hipLaunchKernelGGL(tcn_in_kernel, dim3(n), dim3(256), 0, st, d_signal, m.w_in, m.b_in, MID, tiles, m.dil[0], zr);
        RD_HIP(hipGetLastError());
        return timer_end(st, ctx->timer_in, 2.0 * rows * RD_C * RD_K, rows * (RD_C * 4.0 + 4.0));
    }
    if (kind == 3) {
        if ((rc = timer_begin(st, ctx->timer_head))) return rc;
        if (split) {
            SplitArgs h = {};
            h.zero_row = zero_row;
            h.sink = m.sink;
            h.tiles = tiles;
            h.in = (const _Float16*)Xin;
            h.wpk = (const _Float16*)m.ws_d1;
            h.inv_scale = m.inv_scale_d1;
            h.bias = m.b_d1;
            h.w2 = m.w_d2;
            h.b2 = m.b_d2;
            h.probs = d_probs;
            h.probs_f16 = probs_f16;
            hipLaunchKernelGGL((tcn_gemm_split_kernel<2, 1, EPI_HEAD>), dim3(n), dim3(256), 0, st, h);
        } else {
            ConvArgs h = {};
            h.zero_row = zero_row;
            h.sink = m.sink;
            h.tiles = tiles;
            h.in = Xin;
            h.wpk = m.w_d1;
            h.bias = m.b_d1;
            h.w2 = m.w_d2;
            h.b2 = m.b_d2;
            h.probs = d_probs;
            h.probs_f16 = probs_f16;
            if (ctx->conv_shape == 1) hipLaunchKernelGGL((tcn_gemm_kernel<2, 1, EPI_HEAD, 4>), dim3(n / 2), dim3(512), 0, st, h);
            else hipLaunchKernelGGL((tcn_gemm_kernel<2, 1, EPI_HEAD>), dim3(n), dim3(256), 0, st, h);
        }
        RD_HIP(hipGetLastError());
        return timer_end(st, ctx->timer_head, 2.0 * rows * (RD_C * RD_H + RD_H * 5), rows * (RD_C * 4.0 + 20.0));
    }
    const int wi = 2 * b + (kind == 2 ? 1 : 0);
    ConvArgs a = {};
    a.zero_row = zero_row;
    a.sink = m.sink;
    a.dil = m.dil[b];
    a.tiles = tiles;
    a.wpk = m.w_conv[wi];
    a.bias = m.b_conv[wi];
    SplitArgs sa = {};
    sa.zero_row = zero_row;
    sa.sink = m.sink;
    sa.dil = m.dil[b];
    sa.tiles = tiles;
    sa.wpk = (const _Float16*)m.ws_conv[wi];
    sa.inv_scale = m.inv_scale[wi];
    sa.bias = m.b_conv[wi];
    if (kind == 1) {
        a.in = Xin;
        a.out = MID;
        sa.in = (const _Float16*)Xin;
        sa.out = (_Float16*)MID;
    } else {
        a.in = MID;
        a.out = Xout;
        a.resid = Xin;
        a.x = d_signal;
        a.wmatch = m.w_match;
        a.bmatch = m.b_match;
        sa.in = (const _Float16*)MID;
        sa.out = (_Float16*)Xout;
        sa.resid = (const _Float16*)Xin;
        sa.x = d_signal;
        sa.wmatch = m.w_match;
        sa.bmatch = m.b_match;
    }
    if ((rc = timer_begin(st, ctx->timer_conv))) return rc;
    if (kind == 1) {
        if (split) hipLaunchKernelGGL((tcn_gemm_split_kernel<4, 3, EPI_RELU>), dim3(n), dim3(256), 0, st, sa);
        else if (ctx->conv_shape == 1) hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RELU, 4>), dim3(n / 2), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RELU>), dim3(n), dim3(256), 0, st, a);
    } else if (b == 0) {
        if (fin) {
            a.w_in = m.w_in;
            a.b_in = m.b_in;
            a.zr = zr;
            a.in = nullptr;      // (the A tiles come from a.x)
        }
        if (split) hipLaunchKernelGGL((tcn_gemm_split_kernel<4, 3, EPI_RES_MATCH>), dim3(n), dim3(256), 0, st, sa);
        else if (ctx->conv_shape == 1) hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RES_MATCH, 4>), dim3(n / 2), dim3(512), 0, st, a);
        else if (fin) hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RES_MATCH, 2, true>), dim3(n), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RES_MATCH>), dim3(n), dim3(256), 0, st, a);
    } else {
        if (split) hipLaunchKernelGGL((tcn_gemm_split_kernel<4, 3, EPI_RES_IDENT>), dim3(n), dim3(256), 0, st, sa);
        else if (ctx->conv_shape == 1) hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RES_IDENT, 4>), dim3(n / 2), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((tcn_gemm_kernel<4, 3, EPI_RES_IDENT>), dim3(n), dim3(256), 0, st, a);
    }
    RD_HIP(hipGetLastError());
    return timer_end(st, ctx->timer_conv, 2.0 * rows * RD_C * RD_C * RD_K, (kind == 2 && b > 0 ? 3.0 : 2.0) * rows * RD_C * 4.0, kind == 1 ? 0 : b == 0 ? 2 : 1);
}

}  // namespace

int rd_split3_dev(rd_ctx* ctx, const float* d_in, size_t n, uint16_t* d_out)
{
    if (n == 0) return RD_OK;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, d_out, n);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// CU-mask bit i of a gfx950 device = CU i / 8 of XCD i % 8 (tools/probe/cumask_probe.hip): the first 8 k bits are k CUs of
// every XCD.  (A mask that leaves an XCD without any CU enables that whole XCD: k stays within [1, 31].)
// CU-masked streams are POOLED per process and never destroyed: on ROCm 7.2 hipStreamDestroy of a CU-masked stream can leave
// the runtime hanging in the next call that waits for the device (hipFree / hipStreamDestroy of another stream; seen after
// a run of a few hundred launches on the masked streams, tools/repro_refdefaults.py).  A context takes streams out of the
// pool and hands them back, idle, when it is destroyed or resizes its partition.
namespace {
struct MaskedStream {
    int device, cus;
    bool complement, in_use;
    hipStream_t st;
};
std::vector<MaskedStream> g_masked;
std::mutex g_masked_mu;
}  // namespace

int rd_masked_stream_acquire(int device, int cus_per_xcd, bool complement, hipStream_t* out)
{
    std::lock_guard<std::mutex> lk(g_masked_mu);
    for (MaskedStream& m : g_masked)
        if (!m.in_use && m.device == device && m.cus == cus_per_xcd && m.complement == complement) {
            m.in_use = true;
            *out = m.st;
            return RD_OK;
        }
    uint32_t mask[RD_XCDS] = {0};
    const int nbits = RD_XCDS * 32, k = RD_XCDS * cus_per_xcd;
    for (int i = 0; i < nbits; i++)
        if ((i < k) != complement) mask[i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    RD_HIP(hipExtStreamCreateWithCUMask(&st, RD_XCDS, mask));
    g_masked.push_back({device, cus_per_xcd, complement, true, st});
    *out = st;
    return RD_OK;
}

void rd_masked_stream_release(hipStream_t st)
{
    if (!st) return;
    (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_masked_mu);
    for (MaskedStream& m : g_masked)
        if (m.st == st) m.in_use = false;
}

int rd_lane_get(rd_ctx* ctx, int lane, FwdLane** out)
{
    if (lane < 0 || lane >= 2 * RD_MAX_LANES) {
        rd_set_error("forward lane %d out of range", lane);
        return RD_ERR_ARG;
    }
    FwdLane& L = ctx->lanes[lane];
    if (!L.st) {
        if (lane == 0) L.st = ctx->stream;
        else if (lane < RD_MAX_LANES) {
            RD_HIP(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
        }
        else {
            if (ctx->part_cus < 1 || ctx->part_cus > 31) {
                rd_set_error("internal: partitioned lane %d without a decode partition", lane);
                return RD_ERR_STATE;
            }
            int rc = rd_masked_stream_acquire(ctx->device, ctx->part_cus, true, &L.st);
            if (rc) return rc;
        }
    }
    if (!L.done) RD_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    *out = &L;
    return RD_OK;
}

int rd_part_set(rd_ctx* ctx, int cus_per_xcd)
{
    if (ctx->part_cus == cus_per_xcd) return RD_OK;
    for (int i = RD_MAX_LANES; i < 2 * RD_MAX_LANES; i++) {   // streams masked for another partition size (the caller has flushed)
        FwdLane& L = ctx->lanes[i];
        if (L.st) {
            rd_masked_stream_release(L.st);   // (idle first; back to the process's pool)
            L.st = nullptr;
        }
    }
    ctx->part_cus = cus_per_xcd;
    return RD_OK;
}

int rd_sync_lanes(rd_ctx* ctx)
{
    RD_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 1; i < 2 * RD_MAX_LANES; i++)
        if (ctx->lanes[i].st) RD_HIP(hipStreamSynchronize(ctx->lanes[i].st));
    return RD_OK;
}

int rd_forward_tiles_dev(rd_ctx* ctx, const float* d_signal, const TileLists& tl, int64_t total_rows, void* d_probs, int lane, int probs_f16)
{
    Model& m = ctx->model;
    if (!m.loaded) {
        rd_set_error("rd_forward: no weights loaded (rd_load_weights)");
        return RD_ERR_STATE;
    }
    if (total_rows == 0) return RD_OK;
    int rc = RD_OK;
    FwdLane* L = nullptr;
    if ((rc = rd_lane_get(ctx, lane, &L))) return rc;
    if (total_rows >= INT32_MAX) {
        rd_set_error("rd_forward: %lld rows in one batch (the kernels index rows with 32 bits)", (long long)total_rows);
        return RD_ERR_ARG;
    }
    // each activation tensor carries one extra row of zeros behind its last row: the source of the causal left padding
    const size_t row_bytes = (size_t)RD_C * (ctx->precision == 2 ? 6 : 4);   // bf16x3 rows carry three bf16 per channel
    const size_t act_bytes = (size_t)(total_rows + 1) * row_bytes;
    if (L->act[0].reserve(act_bytes) || L->act[1].reserve(act_bytes) || L->act[2].reserve(act_bytes)) return RD_ERR_NOMEM;
    float* Xin = L->act[0].as<float>();
    float* Xout = L->act[1].as<float>();
    float* MID = L->act[2].as<float>();
    const int zero_row = (int)total_rows;
    // (the three tensors' zero rows are cleared by the forward's first kernel: clear_zero_rows)
    const int nl = 2 * m.nblocks + 1;
    for (int li = 0; li < nl; li++) {
        const int b = li == nl - 1 ? m.nblocks : li / 2;
        const int kind = li == nl - 1 ? 3 : (li == 0 ? 0 : (li & 1 ? 2 : 1));
        if ((rc = launch_layer(ctx, L->st, b, kind, tl.d[li], tl.n[li], (double)tl.rows[li], zero_row, d_signal, Xin, Xout, MID, (float*)d_probs, probs_f16))) return rc;
        if (kind == 2) {   // block finished: its output becomes the next block's input
            float* t = Xin;
            Xin = Xout;
            Xout = t;
        }
    }
    RD_HIP(hipEventRecord(L->done, L->st));
    return RD_OK;
}

// Tile descriptors of nW uniform windows of T rows; cached on the device per (nW, T).  One list serves every layer.
int rd_uniform_tiles(rd_ctx* ctx, int nW, int T, TileLists* out)
{
    const int subs = (T + 31) / 32;                                  // 32-row sub-tiles per window
    const size_t nsub = (size_t)nW * subs;
    const size_t n = (nsub + 7) / 8 * 2;                              // workgroup tiles (four sub-tiles each), an even number: the bf16x3 kernel's tiles are eight
    if (ctx->tiles_nW != nW || ctx->tiles_T != T) {
        std::vector<TileDesc> h(n * 4);
        for (size_t i = 0; i < n * 4; i++) {
            TileDesc& td = h[i];
            const size_t w = i / subs;
            const int k = (int)(i % subs);
            const bool real = i < nsub;
            td.seg_row = real ? (int64_t)w * T : 0;
            td.src_row = td.seg_row;
            td.alt_row = td.seg_row;
            td.t0 = real ? k * 32 : 0;
            td.seg_len = real ? T : 0;                                // padding sub-tiles are empty
            td.in_len = td.seg_len;
            td.alt_in = INT32_MAX;
            td.alt_res = INT32_MAX;
            td.pad_ = 0;
        }
        if (ctx->ws_tiles.reserve(h.size() * sizeof(TileDesc))) return RD_ERR_NOMEM;
        if (int rcs = rd_sync_lanes(ctx)) return rcs;   // no forward may still be reading the previous descriptors
        RD_HIP(hipMemcpy(ctx->ws_tiles.p, h.data(), h.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        ctx->tiles_nW = nW;
        ctx->tiles_T = T;
    }
    const int nl = 2 * ctx->model.nblocks + 1;
    for (int li = 0; li < RD_MAX_LAYERS; li++) {
        out->d[li] = li < nl ? ctx->ws_tiles.as<TileDesc>() : nullptr;
        out->n[li] = li < nl ? (int)n : 0;
        out->rows[li] = li < nl ? (int64_t)nW * T : 0;
    }
    return RD_OK;
}

// d_windows [nW][T] fp32 (already MAD-normalised) -> d_probs [nW][T][5] fp32; all on ctx->stream.
int rd_forward_dev(rd_ctx* ctx, const float* d_windows, int nW, int T, float* d_probs, int lane)
{
    RD_REQUIRE(nW >= 0 && T >= 1, "rd_forward: bad shape nW=%d T=%d", nW, T);
    if (nW == 0) return RD_OK;
    if (!ctx->model.loaded) {
        rd_set_error("rd_forward: no weights loaded (rd_load_weights)");
        return RD_ERR_STATE;
    }
    TileLists tl;
    int rc = rd_uniform_tiles(ctx, nW, T, &tl);
    if (rc) return rc;
    return rd_forward_tiles_dev(ctx, d_windows, tl, (int64_t)nW * T, d_probs, lane);
}

