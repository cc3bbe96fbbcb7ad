// api.hip -- the C ABI of libradian_hip.so (declared in include/radian_hip.h).
// Context, artefact loading/repacking, host-pointer wrappers around the device paths, fused paths,
// kernel timers and the RCCL start-up broadcast.
#include "common.h"
#include "plan.h"
#include "../../include/radian_hip.h"
#include "../../include/radian_hip_diag.h"   // measurement / diagnostic entry points defined in this file


#include <cmath>
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

using namespace rdi;

// --------------------------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";

void rd_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rd_last_error(void) { return g_err; }
extern "C" int rd_version(void) { return 1; }

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return 0;
    // allocate, then swap: a failed growth leaves the old (smaller, still valid) buffer in place.  Only if the new block
    // does not fit BESIDE the old one is the old one given up first (workspaces carry no state between calls).
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess && p) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();   // (see below)
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
        e = hipMalloc(&np, want);
        if (e != hipSuccess) e = hipMalloc(&np, want = align_up(bytes, 1 << 20));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rd_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return -1;
    }
    if (p) {
        // A workspace can be regrown while kernels launched earlier on another stream still use the old block (the pipeline's shared
        // trie workspace while the previous group's search runs): wait for the device explicitly rather than lean on hipFree's own
        // synchronisation.  Growth is geometric (+ 1/8), so this happens a handful of times in a context's life.
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
    }
    p = np;
    cap = want;
    return 0;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

extern "C" int rd_device_count(int* n)
{
    RD_REQUIRE(n != nullptr, "rd_device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        rd_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return RD_ERR_HIP;
    }
    *n = c;
    return RD_OK;
}


// --------------------------------------------------------------------------------------------- context
extern "C" int rd_create(int device_id, rd_ctx** out)
{
    RD_REQUIRE(out != nullptr, "rd_create: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        rd_set_error("rd_create: no HIP device available (%s); this backend has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RD_ERR_HIP;
    }
    RD_REQUIRE(device_id >= 0 && device_id < n, "rd_create: device_id %d out of range [0,%d)", device_id, n);
    RD_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    RD_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rd_set_error("rd_create: device %d is %s; libradian_hip is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return RD_ERR_HIP;
    }
    rd_ctx* ctx = new rd_ctx();
    ctx->device = device_id;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream_hi, hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        rd_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->stream_hi) (void)hipStreamDestroy(ctx->stream_hi);
        delete ctx;
        return RD_ERR_HIP;
    }
    *out = ctx;
    return RD_OK;
}

static void timer_free(KernelTimer& t)
{
    for (auto ev : t.starts) (void)hipEventDestroy(ev);
    for (auto ev : t.stops) (void)hipEventDestroy(ev);
    t.starts.clear();
    t.stops.clear();
    t.each_flops.clear();
    t.each_tag.clear();
    t.used = 0;
    t.enabled = false;
}

extern "C" int rd_rccl_finalize(rd_ctx* ctx);
void rd_pipe_destroy_internal(rd_ctx* ctx);
int rd_pipe_drain_decode_internal(rd_ctx* ctx);
void rd_plan_cache_destroy_internal(rd_ctx* ctx);

extern "C" int rd_destroy(rd_ctx* ctx)
{
    if (!ctx) return RD_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    rd_rccl_finalize(ctx);
    rd_pipe_destroy_internal(ctx);
    rd_rpipe_destroy(ctx);
    rd_plan_cache_destroy_internal(ctx);
    timer_free(ctx->timer_conv);
    timer_free(ctx->timer_decode);
    timer_free(ctx->timer_head);
    timer_free(ctx->timer_in);
    for (int i = 0; i < 2 * RD_MAX_LANES; i++) {
        FwdLane& L = ctx->lanes[i];
        if (i > 0 && L.st) {
            (void)hipStreamSynchronize(L.st);
            if (i >= RD_MAX_LANES) rd_masked_stream_release(L.st);   // CU-masked streams are pooled, never destroyed (forward.hip)
            else (void)hipStreamDestroy(L.st);
        }
        if (L.done) (void)hipEventDestroy(L.done);
        for (DevBuf& b : L.act) b.release();
    }
    DevBuf* bufs[] = {&ctx->ws_tiles, &ctx->ws_raw, &ctx->ws_in, &ctx->ws_probs, &ctx->ws_mat, &ctx->ws_seq,
                      &ctx->ws_nodes_child, &ctx->ws_nodes_back, &ctx->ws_wide, &ctx->ws_wide_slot, &ctx->ws_queue, &ctx->ws_labels, &ctx->ws_misc, &ctx->model.storage,
                      &ctx->lm.storage, &ctx->lm.gate_storage};
    for (DevBuf* b : bufs) b->release();
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);   // (ctx->stream was synchronised at the top)
    if (ctx->stream_hi) {
        (void)hipStreamSynchronize(ctx->stream_hi);
        (void)hipStreamDestroy(ctx->stream_hi);
    }
    if (ctx->ev_chain) (void)hipEventDestroy(ctx->ev_chain);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RD_OK;
}

extern "C" int rd_set_precision(rd_ctx* ctx, int mode)
{
    RD_REQUIRE(ctx, "rd_set_precision: null context");
    RD_REQUIRE(mode >= 0 && mode <= 2, "rd_set_precision: mode %d (0 = fp32 MFMA, 1 = split-f16 f16x3, 2 = three-term bf16x3)", mode);
    ctx->precision = mode;
    return RD_OK;
}

extern "C" int rd_sync(rd_ctx* ctx)
{
    RD_REQUIRE(ctx, "rd_sync: null context");
    return rd_sync_lanes(ctx);
}

// --------------------------------------------------------------------------------------------- weights
namespace {

constexpr size_t CONV_PK = (size_t)RD_K * RD_C * RD_C;  // 196608 floats
constexpr size_t D1_PK = (size_t)RD_C * RD_H;

struct ModelLayout {
    size_t sink, w_in, b_in, w_match, b_match, w_conv[2 * RD_MAX_BLOCKS], b_conv[2 * RD_MAX_BLOCKS], w_d1, b_d1, w_d2, b_d2;
    size_t ws_conv[2 * RD_MAX_BLOCKS], ws_d1, w3_conv[2 * RD_MAX_BLOCKS], w3_d1, total;
};

ModelLayout model_layout(int nblocks)
{
    ModelLayout L = {};
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += align_up(n, 64);
        return o;
    };
    L.sink = take(1024);
    L.w_in = take(RD_K * RD_C);
    L.b_in = take(RD_C);
    L.w_match = take(RD_C);
    L.b_match = take(RD_C);
    for (int b = 0; b < nblocks; b++)
        for (int w = 0; w < 2; w++) {
            if (b == 0 && w == 0) continue;
            L.w_conv[2 * b + w] = take(CONV_PK);
            L.b_conv[2 * b + w] = take(RD_C);
        }
    for (int b = 0; b < nblocks; b++)
        for (int w = 0; w < 2; w++) {
            if (b == 0 && w == 0) continue;
            L.ws_conv[2 * b + w] = take(CONV_PK);   // split-f16 image: same byte size as the fp32 one
        }
    L.ws_d1 = take(D1_PK);
    for (int b = 0; b < nblocks; b++)
        for (int w = 0; w < 2; w++) {
            if (b == 0 && w == 0) continue;
            L.w3_conv[2 * b + w] = take(CONV_PK * 3 / 2);   // three bf16 per weight = 6 B
        }
    L.w3_d1 = take(D1_PK * 3 / 2);
    L.w_d1 = take(D1_PK);
    L.b_d1 = take(RD_H);
    L.w_d2 = take(RD_H * RD_NCLS);
    L.b_d2 = take(RD_NCLS);
    L.total = off;
    return L;
}

void model_bind(Model& m, const ModelLayout& L)
{
    float* base = m.storage.as<float>();
    m.sink = base + L.sink;
    m.w_in = base + L.w_in;
    m.b_in = base + L.b_in;
    m.w_match = base + L.w_match;
    m.b_match = base + L.b_match;
    for (int b = 0; b < m.nblocks; b++)
        for (int w = 0; w < 2; w++) {
            if (b == 0 && w == 0) continue;
            m.w_conv[2 * b + w] = base + L.w_conv[2 * b + w];
            m.b_conv[2 * b + w] = base + L.b_conv[2 * b + w];
            m.ws_conv[2 * b + w] = base + L.ws_conv[2 * b + w];
            m.w3_conv[2 * b + w] = base + L.w3_conv[2 * b + w];
        }
    m.ws_d1 = base + L.ws_d1;
    m.w3_d1 = base + L.w3_d1;
    m.w_d1 = base + L.w_d1;
    m.b_d1 = base + L.b_d1;
    m.w_d2 = base + L.w_d2;
    m.b_d2 = base + L.b_d2;
}

// LDS image order (forward.hip): rows of 32 floats whose 16-B slots are XOR-swizzled by (row >> 1) & 7.
// fp32 image: 64-B rows (16 floats), 16-B slot XOR (row >> 2) & 3 -- the LDS image of forward.hip's half-stages
static inline int swz_k(int row, int k) { return ((((k >> 2) ^ ((row >> 2) & 3)) << 2) | (k & 3)); }

// Keras conv kernel [j][ci][co] -> [chunk = (ci/16)*3 + j][co][swizzled ci%16]
void pack_conv(const float* k, float* dst)
{
    for (int j = 0; j < RD_K; j++)
        for (int ci = 0; ci < RD_C; ci++) {
            const float* src = k + ((size_t)j * RD_C + ci) * RD_C;
            const int chunk = (ci / 16) * RD_K + j;
            float* d = dst + (size_t)chunk * RD_C * 16;
            for (int co = 0; co < RD_C; co++) d[(size_t)co * 16 + swz_k(co, ci % 16)] = src[co];
        }
}

// Keras dense kernel [ci][h] -> [chunk = ci/32][h][swizzled ci%32]
void pack_dense(const float* k, float* dst)
{
    for (int ci = 0; ci < RD_C; ci++) {
        const float* src = k + (size_t)ci * RD_H;
        float* d = dst + (size_t)(ci / 16) * RD_H * 16;
        for (int h = 0; h < RD_H; h++) d[(size_t)h * 16 + swz_k(h, ci % 16)] = src[h];
    }
}

// power-of-two scale that brings max|w| into [512, 1024): the lo halves of the split stay normal f16 numbers
float split_scale(const float* w, size_t n)
{
    float mx = 0.f;
    for (size_t i = 0; i < n; i++) mx = fabsf(w[i]) > mx ? fabsf(w[i]) : mx;
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int e = 0;
    frexpf(mx, &e);              // mx = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, 10 - e);  // mx * scale in [512, 1024)
}

// one 64-B row of a 16-channel chunk: [16 hi | 16 lo] halves, 16-B slots (hi 0-7, hi 8-15, lo 0-7, lo 8-15) XOR (row >> 2) & 3
inline void split_store(_Float16* row, int k, int rowidx, float v)
{
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int sw = (rowidx >> 2) & 3;
    row[((k >> 3) ^ sw) * 8 + (k & 7)] = hi;
    row[((2 + (k >> 3)) ^ sw) * 8 + (k & 7)] = lo;
}

// Keras conv kernel [j][ci][co] -> split image [chunk = (ci/16)*3 + j][co][hi 16 | lo 16] (16-B slots swizzled)
float pack_conv_split(const float* k, _Float16* dst)
{
    const float sc = split_scale(k, CONV_PK);
    for (int j = 0; j < RD_K; j++)
        for (int ci = 0; ci < RD_C; ci++) {
            const float* src = k + ((size_t)j * RD_C + ci) * RD_C;
            const int chunk = (ci / 16) * RD_K + j;
            _Float16* d = dst + (size_t)chunk * RD_C * 32;
            for (int co = 0; co < RD_C; co++) split_store(d + (size_t)co * 32, ci % 16, co, src[co] * sc);
        }
    return 1.f / sc;
}

float pack_dense_split(const float* k, _Float16* dst)
{
    const float sc = split_scale(k, D1_PK);
    for (int ci = 0; ci < RD_C; ci++) {
        const float* src = k + (size_t)ci * RD_H;
        _Float16* d = dst + (size_t)(ci / 16) * RD_H * 32;
        for (int h = 0; h < RD_H; h++) split_store(d + (size_t)h * 32, ci % 16, h, src[h] * sc);
    }
    return 1.f / sc;
}

// fp32 -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does); finite inputs
inline uint16_t f2bf(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}
// one 96-B row of a 16-channel chunk: six 16-B slots (term * 2 + k / 8), physical slot = slot ^ ((row >> 3) & 1)
inline void bf3_store(uint16_t* row, int k, int rowidx, float v)
{
    uint16_t hi = f2bf(v);
    if ((hi & 0x7fffu) == 0x7f80u && std::isfinite(v)) {   // rounded up to infinity: truncate (as split3 on the device)
        uint32_t u;
        memcpy(&u, &v, 4);
        hi = (uint16_t)(u >> 16);
    }
    const float r1 = v - bf2f(hi);
    const uint16_t mid = f2bf(r1);
    const uint16_t lo = f2bf(r1 - bf2f(mid));
    const int sw = (rowidx >> 3) & 1, kh = k >> 3, kl = k & 7;
    row[((0 + kh) ^ sw) * 8 + kl] = hi;
    row[((2 + kh) ^ sw) * 8 + kl] = mid;
    row[((4 + kh) ^ sw) * 8 + kl] = lo;
}
// Keras conv kernel [j][ci][co] -> bf16x3 image [chunk = (ci/16)*3 + j][co][6 slots x 8]
void pack_conv_bf3(const float* k, uint16_t* dst)
{
    for (int j = 0; j < RD_K; j++)
        for (int ci = 0; ci < RD_C; ci++) {
            const float* src = k + ((size_t)j * RD_C + ci) * RD_C;
            const int chunk = (ci / 16) * RD_K + j;
            uint16_t* d = dst + (size_t)chunk * RD_C * 48;
            for (int co = 0; co < RD_C; co++) bf3_store(d + (size_t)co * 48, ci % 16, co, src[co]);
        }
}
void pack_dense_bf3(const float* k, uint16_t* dst)
{
    for (int ci = 0; ci < RD_C; ci++) {
        const float* src = k + (size_t)ci * RD_H;
        uint16_t* d = dst + (size_t)(ci / 16) * RD_H * 48;
        for (int h = 0; h < RD_H; h++) bf3_store(d + (size_t)h * 48, ci % 16, h, src[h]);
    }
}

}  // namespace

extern "C" int rd_load_weights(rd_ctx* ctx, const void* blob, size_t nbytes)
{
    RD_REQUIRE(ctx && blob, "rd_load_weights: null argument");
    RD_REQUIRE(nbytes >= sizeof(rd_weights_header), "rd_load_weights: blob too small (%zu bytes)", nbytes);
    rd_weights_header h;
    memcpy(&h, blob, sizeof(h));
    RD_REQUIRE(h.magic == 0x574e4452u, "rd_load_weights: bad magic 0x%08x", h.magic);
    RD_REQUIRE(h.version == 1, "rd_load_weights: unsupported version %u", h.version);
    RD_REQUIRE(h.nb_filters == RD_C && h.kernel_size == RD_K && h.relu_units == RD_H && h.n_classes == RD_NCLS,
               "rd_load_weights: geometry (%u filters, k=%u, %u relu units, %u classes) is not sig2seq.yaml's (256,3,128,5)",
               h.nb_filters, h.kernel_size, h.relu_units, h.n_classes);
    RD_REQUIRE(h.n_blocks >= 1 && h.n_blocks <= RD_MAX_BLOCKS, "rd_load_weights: n_blocks %u out of range", h.n_blocks);
    const int nb = (int)h.n_blocks;
    size_t expect = (size_t)RD_K * RD_C + RD_C + CONV_PK + RD_C + RD_C + RD_C;
    expect += (size_t)(nb - 1) * 2 * (CONV_PK + RD_C);
    expect += D1_PK + RD_H + (size_t)RD_H * RD_NCLS + RD_NCLS;
    RD_REQUIRE(h.n_floats == expect, "rd_load_weights: header says %u floats, geometry needs %zu", h.n_floats, expect);
    RD_REQUIRE(nbytes == sizeof(h) + expect * sizeof(float), "rd_load_weights: blob is %zu bytes, expected %zu", nbytes,
               sizeof(h) + expect * sizeof(float));
    for (int b = 0; b < nb; b++) RD_REQUIRE(h.dilations[b] >= 1 && h.dilations[b] <= 4096, "rd_load_weights: bad dilation");
    const float* w = (const float*)((const char*)blob + sizeof(h));

    RD_HIP(hipSetDevice(ctx->device));
    Model& m = ctx->model;
    m.loaded = false;
    m.nblocks = nb;
    for (int b = 0; b < nb; b++) m.dil[b] = (int)h.dilations[b];
    ModelLayout L = model_layout(nb);
    std::vector<float> host(L.total, 0.f);
    size_t off = 0;
    for (int b = 0; b < nb; b++) {
        if (b == 0) {
            memcpy(&host[L.w_in], w + off, sizeof(float) * RD_K * RD_C);
            off += RD_K * RD_C;
            memcpy(&host[L.b_in], w + off, sizeof(float) * RD_C);
            off += RD_C;
        } else {
            pack_conv(w + off, &host[L.w_conv[2 * b]]);
            m.inv_scale[2 * b] = pack_conv_split(w + off, (_Float16*)&host[L.ws_conv[2 * b]]);
            pack_conv_bf3(w + off, (uint16_t*)&host[L.w3_conv[2 * b]]);
            off += CONV_PK;
            memcpy(&host[L.b_conv[2 * b]], w + off, sizeof(float) * RD_C);
            off += RD_C;
        }
        pack_conv(w + off, &host[L.w_conv[2 * b + 1]]);
        m.inv_scale[2 * b + 1] = pack_conv_split(w + off, (_Float16*)&host[L.ws_conv[2 * b + 1]]);
        pack_conv_bf3(w + off, (uint16_t*)&host[L.w3_conv[2 * b + 1]]);
        off += CONV_PK;
        memcpy(&host[L.b_conv[2 * b + 1]], w + off, sizeof(float) * RD_C);
        off += RD_C;
        if (b == 0) {
            memcpy(&host[L.w_match], w + off, sizeof(float) * RD_C);
            off += RD_C;
            memcpy(&host[L.b_match], w + off, sizeof(float) * RD_C);
            off += RD_C;
        }
    }
    pack_dense(w + off, &host[L.w_d1]);
    m.inv_scale_d1 = pack_dense_split(w + off, (_Float16*)&host[L.ws_d1]);
    pack_dense_bf3(w + off, (uint16_t*)&host[L.w3_d1]);
    off += D1_PK;
    memcpy(&host[L.b_d1], w + off, sizeof(float) * RD_H);
    off += RD_H;
    memcpy(&host[L.w_d2], w + off, sizeof(float) * RD_H * RD_NCLS);
    off += RD_H * RD_NCLS;
    memcpy(&host[L.b_d2], w + off, sizeof(float) * RD_NCLS);
    off += RD_NCLS;
    if (off != expect) {
        rd_set_error("rd_load_weights: internal size mismatch");
        return RD_ERR_ARG;
    }
    if (m.storage.reserve(L.total * sizeof(float))) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpy(m.storage.p, host.data(), L.total * sizeof(float), hipMemcpyHostToDevice));
    model_bind(m, L);
    m.loaded = true;
    return RD_OK;
}

int rd_model_halo(const rd_ctx* ctx)
{
    int s = 0;
    for (int b = 0; b < ctx->model.nblocks; b++) s += ctx->model.dil[b];
    return (RD_K - 1) * 2 * s;
}

// --------------------------------------------------------------------------------------------- LM
// doubles of the LM image: table [n][4], entropies [n], then one bit per context ("absent from a sparse model"), padded to doubles
static size_t lm_image_doubles(int table_order)
{
    const size_t n = (size_t)1 << (2 * table_order);
    return n * 5 + (n + 63) / 64;
}

static void lm_bind(LM& lm)
{
    const size_t n = (size_t)1 << (2 * lm.table_order);
    lm.table = lm.storage.as<double>();
    lm.d_entropy = lm.table + n * 4;
    lm.d_missing = (uint32_t*)(lm.table + n * 5);
}

static int load_lm_table(rd_ctx* ctx, const double* table, int table_order, int context_len, int hashed)
{
    LM& lm = ctx->lm;
    lm.loaded = false;
    lm.gate_valid = false;
    if (!table) return RD_OK;
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << (2 * table_order);
    // per-context entropy, decode.py:73-76,85-90 (math.log == glibc log; python sum is left-assoc)
    // A row of NaNs marks a context that the (sparse) model does not hold: the reference's dict lookup raises KeyError when the
    // search reaches it (decode.py:83).  Such rows are zeroed on the device, their gate stays closed (entropy +inf) and their bit
    // is set in the "absent" mask, which the beam search checks for every labeling that enters the beam.
    std::vector<double> ent(n);
    std::vector<uint32_t> missing(((n + 63) / 64) * 2, 0u);
    std::vector<double> patched;
    size_t n_missing = 0;
    for (size_t c = 0; c < n && !hashed; c++)
        if (table[c * 4] != table[c * 4]) {
            if (patched.empty()) patched.assign(table, table + n * 4);
            for (int i = 0; i < 4; i++) patched[c * 4 + i] = 0.0;
            missing[c >> 5] |= 1u << (c & 31);
            n_missing++;
        }
    if (n_missing) table = patched.data();
    for (size_t c = 0; c < n; c++) {
        const double* d = table + c * 4;
        double s = 0.0;
        bool any = false;
        for (int i = 0; i < 4; i++)
            if (d[i] > 0) {
                double v = d[i] * log(d[i]);
                s = any ? s + v : v;
                any = true;
            }
        ent[c] = any ? -s : 0.0;
        if (n_missing && ((missing[c >> 5] >> (c & 31)) & 1u)) ent[c] = INFINITY;
    }
    lm.k = context_len;
    lm.table_order = table_order;
    lm.hashed = hashed;
    lm.sparse = n_missing ? 1 : 0;
    if (lm.storage.reserve(lm_image_doubles(table_order) * sizeof(double))) return RD_ERR_NOMEM;
    lm_bind(lm);
    RD_HIP(hipMemcpy(lm.table, table, n * 4 * sizeof(double), hipMemcpyHostToDevice));
    RD_HIP(hipMemcpy(lm.d_entropy, ent.data(), n * sizeof(double), hipMemcpyHostToDevice));
    RD_HIP(hipMemcpy(lm.d_missing, missing.data(), missing.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    lm.loaded = true;
    return RD_OK;
}

extern "C" int rd_load_lm(rd_ctx* ctx, const double* table, int k)
{
    RD_REQUIRE(ctx, "rd_load_lm: null context");
    if (!table) return load_lm_table(ctx, nullptr, 0, 0, 0);
    RD_REQUIRE(k >= 1 && k <= 13, "rd_load_lm: context length %d out of range [1,13] (longer contexts: rd_load_lm_hashed)", k);
    return load_lm_table(ctx, table, k, k, 0);
}

// An RNA model none of whose keys has k characters: `model[context]` (decode.py:83) raises KeyError for every context of k labels.  The
// image is an ordinary sparse one -- every row absent, every gate bit closed (entropy NaN compares false) -- filled on the device.
extern "C" int rd_load_lm_absent(rd_ctx* ctx, int k)
{
    RD_REQUIRE(ctx, "rd_load_lm_absent: null context");
    RD_REQUIRE(k >= 1 && k <= 13, "rd_load_lm_absent: context length %d out of range [1,13]", k);
    LM& lm = ctx->lm;
    lm.loaded = false;
    lm.gate_valid = false;
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << (2 * k);
    lm.k = k;
    lm.table_order = k;
    lm.hashed = 0;
    lm.sparse = 1;
    if (lm.storage.reserve(lm_image_doubles(k) * sizeof(double))) return RD_ERR_NOMEM;
    lm_bind(lm);
    RD_HIP(hipMemset(lm.table, 0, n * 4 * sizeof(double)));
    RD_HIP(hipMemset(lm.d_entropy, 0xff, n * sizeof(double)));                  // NaN: `entropy < r_threshold` is false
    RD_HIP(hipMemset(lm.d_missing, 0xff, ((n + 63) / 64) * sizeof(double)));
    RD_HIP(hipDeviceSynchronize());
    lm.loaded = true;
    return RD_OK;
}

extern "C" int rd_load_lm_hashed(rd_ctx* ctx, const double* table, int table_order, int context_len)
{
    RD_REQUIRE(ctx && table, "rd_load_lm_hashed: null argument");
    RD_REQUIRE(table_order >= 1 && table_order <= 13, "rd_load_lm_hashed: table order %d out of range [1,13]", table_order);
    RD_REQUIRE(context_len >= 1 && context_len <= 256, "rd_load_lm_hashed: context length %d out of range [1,256]", context_len);
    return load_lm_table(ctx, table, table_order, context_len, 1);
}

extern "C" int rd_set_logits(rd_ctx* ctx, int mode)
{
    RD_REQUIRE(ctx, "rd_set_logits: null context");
    RD_REQUIRE(mode == 0 || mode == 1, "rd_set_logits: mode %d (0 = float32 rows, 1 = float16 rows)", mode);
    ctx->logits_f16 = mode;
    return RD_OK;
}

extern "C" int rd_set_decode_form(rd_ctx* ctx, int form)
{
    RD_REQUIRE(ctx, "rd_set_decode_form: null context");
    RD_REQUIRE(form >= 0 && form <= 5, "rd_set_decode_form: form %d (0 = per launch, 1 = waves per sequence, 2 = candidates per lane, 3 = W <= 12: two sequences per "
               "wave, 4 = one, 5 = work queue)", form);
    ctx->decode_form = form;
    return RD_OK;
}

extern "C" int rd_set_trie_budget(rd_ctx* ctx, int64_t bytes)
{
    RD_REQUIRE(ctx, "rd_set_trie_budget: null context");
    RD_REQUIRE(bytes >= 0, "rd_set_trie_budget: negative budget");
    ctx->trie_budget = bytes ? bytes : (int64_t)24 << 30;
    return RD_OK;
}

extern "C" int rd_set_conv_shape(rd_ctx* ctx, int shape)
{
    RD_REQUIRE(ctx, "rd_set_conv_shape: null context");
    RD_REQUIRE(shape == 0 || shape == 1, "rd_set_conv_shape: shape %d (0 = 128-row tiles, two workgroups per CU; 1 = 256-row tiles, one workgroup per CU)", shape);
    ctx->conv_shape = shape;
    return RD_OK;
}

extern "C" int rd_set_conv_fuse(rd_ctx* ctx, int on)
{
    RD_REQUIRE(ctx, "rd_set_conv_fuse: null context");
    RD_REQUIRE(on == 0 || on == 1, "rd_set_conv_fuse: %d (1 = block 0's first conv inside its second, 0 = its own kernel)", on);
    ctx->conv_fuse = on;
    return RD_OK;
}

extern "C" int rd_set_decode_partition(rd_ctx* ctx, int cus_per_xcd)
{
    RD_REQUIRE(ctx, "rd_set_decode_partition: null context");
    RD_REQUIRE(cus_per_xcd >= -1 && cus_per_xcd <= 16, "rd_set_decode_partition: %d CUs per XCD (-1 = by beam width, 0 = off, 1..16)", cus_per_xcd);
    RD_REQUIRE(rd_rpipe_idle(ctx), "rd_set_decode_partition: pipeline not empty (call rd_pipe_flush first)");
    ctx->part_mode = cus_per_xcd;
    return RD_OK;
}

extern "C" int rd_set_decode_math(rd_ctx* ctx, int mode)
{
    RD_REQUIRE(ctx, "rd_set_decode_math: null context");
    RD_REQUIRE(mode == 0 || mode == 1, "rd_set_decode_math: mode %d (0 = library routines, 1 = glibc's operation sequence)", mode);
    ctx->decode_math = mode;
    return RD_OK;
}

// --------------------------------------------------------------------------------------------- helpers
namespace {

struct SeqMeta {
    // device pointers inside ctx->ws_seq
    int64_t *d_seq_off, *d_seq_off2, *d_node_off, *d_label_off;
    int32_t *d_seq_len, *d_split, *d_label_len;
    double* d_score;
    int64_t total_labels;
    std::vector<TrieRun> runs;   // launches that share the trie workspace, one after the other (rd_plan_trie_runs)
};

// uploads per-sequence metadata; seq_off2/split may be null (single source region per sequence)
int prepare_seq_meta(rd_ctx* ctx, const int64_t* seq_off, const int64_t* seq_off2, const int32_t* split, const int32_t* seq_len,
                     int n_seq, int W, SeqMeta& sm, std::vector<int64_t>& label_off_used)
{
    std::vector<int64_t> node_off(n_seq), lab_off(n_seq);
    int64_t labs = 0;
    for (int i = 0; i < n_seq; i++) {
        RD_REQUIRE(seq_len[i] >= 0, "decode: negative sequence length at %d", i);
        RD_REQUIRE(rd_decode_len_ok(W, seq_len[i]), "decode: sequence %d has %d rows; beam width %d supports at most %lld (1 + W * rows < 2^29)", i,
                   seq_len[i], W, (long long)((((int64_t)1 << 29) - 2) / W));
        lab_off[i] = labs;
        labs += seq_len[i];
    }
    sm.runs.clear();
    rd_plan_trie_runs(ctx, W, 0, n_seq, [&](int k) { return (int64_t)seq_len[k]; }, node_off.data(), sm.runs);
    label_off_used = lab_off;
    const size_t n = (size_t)n_seq;
    const size_t a8 = align_up(n * 8, 256), a4 = align_up(n * 4, 256);
    if (ctx->ws_seq.reserve(5 * a8 + 3 * a4)) return RD_ERR_NOMEM;
    char* p = (char*)ctx->ws_seq.p;
    sm.d_seq_off = (int64_t*)p; p += a8;
    sm.d_seq_off2 = (int64_t*)p; p += a8;
    sm.d_node_off = (int64_t*)p; p += a8;
    sm.d_label_off = (int64_t*)p; p += a8;
    sm.d_score = (double*)p; p += a8;
    sm.d_seq_len = (int32_t*)p; p += a4;
    sm.d_split = (int32_t*)p; p += a4;
    sm.d_label_len = (int32_t*)p;
    sm.total_labels = labs;
    // The metadata goes through the context's pinned staging block in ONE copy that needs no host-side wait: the block
    // is only rewritten by the next call on this context, and a caller that runs to completion ends with a stream
    // synchronisation (decode_and_fetch) before that can happen.  A caller that left early on an error did not: then the
    // copy may still be reading the block -- wait for it here before the block is rewritten (or freed).
    if (ctx->h_stage_busy) {
        RD_HIP(hipStreamSynchronize(ctx->stream));
        ctx->h_stage_busy = false;
    }
    const size_t stage_bytes = 4 * a8 + 2 * a4;   // seq_off | seq_off2 | node_off | label_off | seq_len | split
    if (ctx->h_stage_cap < stage_bytes) {
        RD_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr;
        ctx->h_stage_cap = 0;
        const size_t want = align_up(stage_bytes + stage_bytes / 4, 1 << 16);
        RD_HIP(hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault));
        ctx->h_stage_cap = want;
    }
    char* hs = (char*)ctx->h_stage;
    memcpy(hs, seq_off, n * 8);
    if (seq_off2) memcpy(hs + a8, seq_off2, n * 8);
    memcpy(hs + 2 * a8, node_off.data(), n * 8);
    memcpy(hs + 3 * a8, lab_off.data(), n * 8);
    memcpy(hs + 4 * a8, seq_len, n * 4);
    if (split) memcpy(hs + 4 * a8 + a4, split, n * 4);
    // device layout: 5 x a8 (seq_off, seq_off2, node_off, label_off, score) then 3 x a4 (seq_len, split, label_len)
    ctx->h_stage_busy = true;
    RD_HIP(hipMemcpyAsync(sm.d_seq_off, hs, 4 * a8, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipMemcpyAsync(sm.d_seq_len, hs + 4 * a8, 2 * a4, hipMemcpyHostToDevice, ctx->stream));
    if (!seq_off2) {
        sm.d_seq_off2 = nullptr;
        sm.d_split = nullptr;
    }
    return RD_OK;
}

// decode sequences over device rows and deliver labels to host buffers
int decode_and_fetch(rd_ctx* ctx, const void* d_probs, int is_f64, const int64_t* seq_off, const int32_t* seq_len, int n_seq,
                     int W, int use_lm, double s_thr, double r_thr, uint8_t* labels_out, const int64_t* label_off,
                     int32_t* label_len, double* best_score, const int64_t* seq_off2 = nullptr, const int32_t* split = nullptr)
{
    if (n_seq == 0) return RD_OK;
    SeqMeta sm;
    std::vector<int64_t> lab_off;
    // beam searches still in flight on a pipeline's decode stream use the context's trie workspace: let them finish
    int rc = rd_pipe_drain_decode_internal(ctx);
    if (rc) return rc;
    if ((rc = rd_rpipe_drain_decode(ctx))) return rc;
    rc = prepare_seq_meta(ctx, seq_off, seq_off2, split, seq_len, n_seq, W, sm, lab_off);
    if (rc) return rc;
    if (ctx->ws_labels.reserve((size_t)sm.total_labels + 16)) return RD_ERR_NOMEM;
    uint8_t* d_labels = ctx->ws_labels.as<uint8_t>();
    // everything queued on the context's stream so far (forward, assembly, metadata) precedes the beam search, which runs on
    // the high-priority stream: with another context's forward filling the chip, its few waves are dispatched first
    hipStream_t ds = ctx->stream_hi;
    RD_HIP(hipEventRecord(ctx->ev_chain, ctx->stream));
    RD_HIP(hipStreamWaitEvent(ds, ctx->ev_chain, 0));
    for (const TrieRun& r : sm.runs) {
        rc = rd_decode_dev(ctx, d_probs, is_f64, sm.d_seq_off + r.k0, sm.d_seq_len + r.k0, sm.d_node_off + r.k0, sm.d_label_off + r.k0, r.k1 - r.k0,
                           r.nodes, W, use_lm, s_thr, r_thr, d_labels, sm.d_label_len + r.k0, best_score ? sm.d_score + r.k0 : nullptr, ds,
                           sm.d_seq_off2 ? sm.d_seq_off2 + r.k0 : nullptr, sm.d_split ? sm.d_split + r.k0 : nullptr);
        if (rc) return rc;
    }
    std::vector<uint8_t> hl((size_t)sm.total_labels + 16);
    RD_HIP(hipMemcpyAsync(hl.data(), d_labels, (size_t)sm.total_labels, hipMemcpyDeviceToHost, ds));
    RD_HIP(hipMemcpyAsync(label_len, sm.d_label_len, (size_t)n_seq * 4, hipMemcpyDeviceToHost, ds));
    if (best_score) RD_HIP(hipMemcpyAsync(best_score, sm.d_score, (size_t)n_seq * 8, hipMemcpyDeviceToHost, ds));
    RD_HIP(hipStreamSynchronize(ds));
    ctx->h_stage_busy = false;   // (ds waited for ctx->stream's event: the metadata copy is done)
    for (int i = 0; i < n_seq; i++) {
        if (label_len[i] == RD_LEN_MISSING_CONTEXT && use_lm) continue;   // sparse LM: the search reached an absent context (the caller raises KeyError)
        if (label_len[i] < 0 || label_len[i] > seq_len[i]) {
            rd_set_error("decode: sequence %d produced an impossible label length %d (rows %d)", i, label_len[i], seq_len[i]);
            return RD_ERR_STATE;
        }
        if (label_len[i]) memcpy(labels_out + label_off[i], hl.data() + lab_off[i], (size_t)label_len[i]);
    }
    return RD_OK;
}

}  // namespace

// --------------------------------------------------------------------------------------------- seams
extern "C" int rd_forward(rd_ctx* ctx, const float* windows, int n_windows, int chunk_len, float* probs)
{
    RD_REQUIRE(ctx && (n_windows == 0 || (windows && probs)), "rd_forward: null argument");
    RD_REQUIRE(n_windows >= 0 && chunk_len >= 1, "rd_forward: bad shape n_windows=%d chunk_len=%d", n_windows, chunk_len);
    if (n_windows == 0) return RD_OK;
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_windows * chunk_len;
    if (ctx->ws_in.reserve(n * 4) || ctx->ws_probs.reserve(n * 20)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, windows, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rd_forward_dev(ctx, ctx->ws_in.as<float>(), n_windows, chunk_len, ctx->ws_probs.as<float>());
    if (rc) return rc;
    RD_HIP(hipMemcpyAsync(probs, ctx->ws_probs.p, n * 20, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

extern "C" int rd_assemble(rd_ctx* ctx, const float* probs, int n_windows, int chunk_len, int pad, int step, double* out,
                           int64_t out_cap, int64_t* n_rows, int* is_f64)
{
    RD_REQUIRE(ctx && probs && out && n_rows, "rd_assemble: null argument");
    RD_REQUIRE(n_windows >= 1 && chunk_len >= 1, "rd_assemble: bad shape");
    RD_REQUIRE(step >= 1 && step <= chunk_len, "rd_assemble: step %d must be in [1, chunk_len]", step);
    RD_REQUIRE(pad >= 0 && pad <= chunk_len, "rd_assemble: pad %d out of range", pad);
    RD_HIP(hipSetDevice(ctx->device));
    const int64_t N = assembled_rows(n_windows, chunk_len, pad, step);
    RD_REQUIRE(N <= out_cap, "rd_assemble: output needs %lld rows, capacity %lld", (long long)N, (long long)out_cap);
    const size_t nin = (size_t)n_windows * chunk_len * 5;
    if (ctx->ws_probs.reserve(nin * 4) || ctx->ws_mat.reserve((size_t)(N + 1) * 40)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_probs.p, probs, nin * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rd_assemble_dev(ctx, ctx->ws_probs.as<float>(), n_windows, chunk_len, pad, step, ctx->ws_mat.as<double>(), N);
    if (rc) return rc;
    if (N) RD_HIP(hipMemcpyAsync(out, ctx->ws_mat.p, (size_t)N * 40, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    *n_rows = N;
    if (is_f64) *is_f64 = assembled_is_f64(n_windows, chunk_len, pad, step);
    return RD_OK;
}

extern "C" int rd_decode_batch(rd_ctx* ctx, const void* probs, int prob_is_f64, const int64_t* seq_off, const int32_t* seq_len,
                               int n_seq, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                               const int64_t* label_off, int32_t* label_len, double* best_score)
{
    RD_REQUIRE(ctx, "rd_decode_batch: null context");
    RD_REQUIRE(n_seq >= 0, "rd_decode_batch: negative n_seq");
    if (n_seq == 0) return RD_OK;
    RD_REQUIRE(seq_off && seq_len && labels_out && label_off && label_len, "rd_decode_batch: null argument");
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "rd_decode_batch: beam_width %d out of range [1,%d]",
               beam_width, rd_decode_max_width());
    RD_REQUIRE_WIDTH_LM(ctx, beam_width, use_lm);
    RD_HIP(hipSetDevice(ctx->device));
    int64_t rows = 0;
    for (int i = 0; i < n_seq; i++) {
        RD_REQUIRE(seq_len[i] >= 0 && seq_off[i] >= 0, "rd_decode_batch: bad sequence %d", i);
        RD_REQUIRE(rd_decode_len_ok(beam_width, seq_len[i]), "rd_decode_batch: sequence %d has %d rows; beam width %d supports at most %lld (1 + W * rows < 2^29)",
                   i, seq_len[i], beam_width, (long long)((((int64_t)1 << 29) - 2) / beam_width));
        if (seq_off[i] + seq_len[i] > rows) rows = seq_off[i] + seq_len[i];
    }
    RD_REQUIRE(rows == 0 || probs, "rd_decode_batch: null probs");
    const size_t rb = prob_is_f64 ? 40 : 20;
    if (ctx->ws_mat.reserve((size_t)(rows + 1) * rb)) return RD_ERR_NOMEM;
    if (rows) RD_HIP(hipMemcpyAsync(ctx->ws_mat.p, probs, (size_t)rows * rb, hipMemcpyHostToDevice, ctx->stream));
    return decode_and_fetch(ctx, ctx->ws_mat.p, prob_is_f64, seq_off, seq_len, n_seq, beam_width, use_lm, s_thr, r_thr, labels_out,
                            label_off, label_len, best_score);
}

// --------------------------------------------------------------------------------------------- fused
static int chunk_decode_from_probs(rd_ctx* ctx, const float* d_probs, int n_windows, int chunk_len, const int32_t* valid_len,
                                   int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    std::vector<int64_t> seq_off(n_windows), lab_off(n_windows);
    for (int i = 0; i < n_windows; i++) {
        RD_REQUIRE(valid_len[i] >= 0 && valid_len[i] <= chunk_len, "valid_len[%d]=%d out of range [0,%d]", i, valid_len[i], chunk_len);
        seq_off[i] = (int64_t)i * chunk_len;
        lab_off[i] = (int64_t)i * chunk_len;
    }
    return decode_and_fetch(ctx, d_probs, 0, seq_off.data(), valid_len, n_windows, beam_width, 0, 0.0, 0.0, labels_out, lab_off.data(),
                            label_len, nullptr);
}

extern "C" int rd_basecall_chunk_resident(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len,
                                          const int32_t* valid_len, int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    RD_REQUIRE(ctx && d_windows && valid_len && labels_out && label_len, "rd_basecall_chunk_resident: null argument");
    RD_REQUIRE(n_windows >= 1 && chunk_len >= 1, "rd_basecall_chunk_resident: bad shape");
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "beam_width %d out of range", beam_width);
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_windows * chunk_len;
    if (ctx->ws_probs.reserve(n * 20)) return RD_ERR_NOMEM;
    int rc = rd_forward_dev(ctx, d_windows, n_windows, chunk_len, ctx->ws_probs.as<float>());
    if (rc) return rc;
    return chunk_decode_from_probs(ctx, ctx->ws_probs.as<float>(), n_windows, chunk_len, valid_len, beam_width, labels_out, label_len);
}

extern "C" int rd_basecall_chunk(rd_ctx* ctx, const float* windows, int n_windows, int chunk_len, const int32_t* valid_len,
                                 int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    RD_REQUIRE(ctx && windows, "rd_basecall_chunk: null argument");
    RD_REQUIRE(n_windows >= 1 && chunk_len >= 1, "rd_basecall_chunk: bad shape");
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_windows * chunk_len;
    if (ctx->ws_in.reserve(n * 4)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, windows, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return rd_basecall_chunk_resident(ctx, ctx->ws_in.as<float>(), n_windows, chunk_len, valid_len, beam_width, labels_out, label_len);
}

extern "C" int rd_decode_resident(rd_ctx* ctx, const float* d_probs, int n_windows, int chunk_len, const int32_t* valid_len,
                                  int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    RD_REQUIRE(ctx && d_probs && valid_len && labels_out && label_len, "rd_decode_resident: null argument");
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "beam_width %d out of range", beam_width);
    RD_HIP(hipSetDevice(ctx->device));
    return chunk_decode_from_probs(ctx, d_probs, n_windows, chunk_len, valid_len, beam_width, labels_out, label_len);
}

extern "C" int rd_forward_resident(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len, float* d_probs)
{
    RD_REQUIRE(ctx && d_windows, "rd_forward_resident: null argument");
    RD_HIP(hipSetDevice(ctx->device));
    if (!d_probs) {
        if (ctx->ws_probs.reserve((size_t)n_windows * chunk_len * 20)) return RD_ERR_NOMEM;
        d_probs = ctx->ws_probs.as<float>();
    }
    return rd_forward_dev(ctx, d_windows, n_windows, chunk_len, d_probs);
}

extern "C" int rd_basecall_global(rd_ctx* ctx, const float* windows, int chunk_len, int step, const int32_t* read_win_off,
                                  const int32_t* pad, int n_reads, int beam_width, int use_lm, double s_thr, double r_thr,
                                  uint8_t* labels_out, const int64_t* label_off, int32_t* label_len)
{
    RD_REQUIRE(ctx && windows && read_win_off && pad && labels_out && label_off && label_len, "rd_basecall_global: null argument");
    RD_REQUIRE(n_reads >= 1 && chunk_len >= 1, "rd_basecall_global: bad shape");
    RD_REQUIRE(step >= 1 && step <= chunk_len, "rd_basecall_global: step %d must be in [1, chunk_len]", step);
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "beam_width %d out of range", beam_width);
    RD_REQUIRE_WIDTH_LM(ctx, beam_width, use_lm);
    RD_HIP(hipSetDevice(ctx->device));
    const int nW = read_win_off[n_reads];
    RD_REQUIRE(nW >= n_reads, "rd_basecall_global: every read needs at least one window");
    const size_t n = (size_t)nW * chunk_len;
    if (ctx->ws_in.reserve(n * 4) || ctx->ws_probs.reserve(n * 20)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, windows, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rd_forward_dev(ctx, ctx->ws_in.as<float>(), nW, chunk_len, ctx->ws_probs.as<float>());
    if (rc) return rc;
    // per-read assembly into one concatenated float64 matrix; reads whose reference dtype is float32
    // (no time step covered twice) are decoded straight from the float32 window rows.
    std::vector<int64_t> seq_off64(n_reads), seq_off32(n_reads);
    std::vector<int32_t> seq_len(n_reads);
    std::vector<int> is64(n_reads);
    int64_t rows64 = 0;
    for (int r = 0; r < n_reads; r++) {
        const int w0 = read_win_off[r], w1 = read_win_off[r + 1];
        RD_REQUIRE(w1 > w0, "rd_basecall_global: read %d has no windows", r);
        RD_REQUIRE(pad[r] >= 0 && pad[r] <= chunk_len, "rd_basecall_global: pad[%d] out of range", r);
        const int64_t N = assembled_rows(w1 - w0, chunk_len, pad[r], step);
        seq_len[r] = (int32_t)N;
        is64[r] = assembled_is_f64(w1 - w0, chunk_len, pad[r], step);
        seq_off64[r] = rows64;
        seq_off32[r] = (int64_t)w0 * chunk_len;
        if (is64[r]) rows64 += N;
    }
    if (ctx->ws_mat.reserve((size_t)(rows64 + 1) * 40)) return RD_ERR_NOMEM;
    for (int r = 0; r < n_reads; r++) {
        if (!is64[r]) continue;
        const int w0 = read_win_off[r], w1 = read_win_off[r + 1];
        rc = rd_assemble_dev(ctx, ctx->ws_probs.as<float>() + (size_t)w0 * chunk_len * 5, w1 - w0, chunk_len, pad[r], step,
                             ctx->ws_mat.as<double>() + seq_off64[r] * 5, seq_len[r]);
        if (rc) return rc;
    }
    // two decode launches: float64 (assembled) reads and float32 (single-coverage) reads
    for (int pass = 0; pass < 2; pass++) {
        std::vector<int64_t> so, lo;
        std::vector<int32_t> sl;
        std::vector<int> idx;
        for (int r = 0; r < n_reads; r++)
            if (is64[r] == (pass == 0)) {
                so.push_back(pass == 0 ? seq_off64[r] : seq_off32[r]);
                sl.push_back(seq_len[r]);
                lo.push_back(label_off[r]);
                idx.push_back(r);
            }
        if (idx.empty()) continue;
        std::vector<int32_t> ll(idx.size());
        rc = decode_and_fetch(ctx, pass == 0 ? (const void*)ctx->ws_mat.p : (const void*)ctx->ws_probs.p, pass == 0 ? 1 : 0, so.data(),
                              sl.data(), (int)idx.size(), beam_width, use_lm, s_thr, r_thr, labels_out, lo.data(), ll.data(), nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < idx.size(); i++) label_len[idx[i]] = ll[i];
    }
    return RD_OK;
}


// --------------------------------------------------------------------------------------------- reads-level paths
// The reference windows every read (chunk_len rows every step samples, preprocess.py:4-22) and runs the model on every
// window, although the TCN is causal with a finite receptive field RF = 1 + (K-1)*2*sum(dilations) (253 samples):
// row r >= RF-1 of a window does not depend on where the window starts.  So the forward is run ONCE over the whole read
// (the "stream"); a window's rows >= halo = RF-1 are the stream's rows, and only its first `halo` rows -- the ones that
// see the window's own zero left-padding -- are computed separately ("heads").  Results are bit-identical to the
// windowed computation (each output row is the same fp32 fma chain over the same values; tests/test_gpu_reads.py), at
// N + (nW-1)*halo rows per read instead of nW*chunk_len (chunk mode), or N rows (global mode, where only the earliest
// covering window's row of each time step is ever used, matrix_assembly.py:46-53; valid when step <= chunk_len - halo).
namespace {
struct PlanCache {
    int chunk = -1, step = -1, halo = -1, mode = -1, nblocks = -1;
    int dil[RD_MAX_BLOCKS] = {0};   // per-layer head lengths depend on every block's dilation, not only on their sum
    std::vector<int64_t> lens;
    ReadsPlan plan;
    bool streamed = false;
    DevBuf d_tiles;
    TileLists lists;
};

// plan + device tile descriptors for a batch of reads, cached while consecutive batches have the same read lengths
}  // namespace

extern "C" int rd_count_windows(int64_t n_samples, int chunk_len, int step)
{
    if (n_samples < 0 || chunk_len < 1 || step < 1 || step > chunk_len) return -1;
    return count_windows(n_samples, chunk_len, step);
}

// --------------------------------------------------------------------------------------------- software pipeline
// Chunk-mode batches flow through several HIP streams.  Forwards (MFMA-bound): consecutive batches rotate over a few
// forward lanes (FwdLane: a stream + its own activation tensors), so two independent kernel chains are in flight and the
// partially filled last round of one chain's launch is filled by the other's workgroups.  Beam search (one wave per
// window, latency-bound): the windows of a GROUP of batches are decoded by one launch on a further, high-priority
// stream together with their label copy-out, overlapped with the forwards of the next group.  Grouping matters because
// the decoder's throughput comes from waves per SIMD: 4 x 512 windows decode in about the time of 512.  Two slots of
// probability / metadata / pinned output buffers; a slot is recycled only after its labels were handed to the caller.
namespace {

struct PipeSub {
    int n = 0;          // windows of this submitted batch
    int win0 = 0;       // first window inside the slot
    uint8_t* user_labels = nullptr;
    int32_t* user_lens = nullptr;
};

struct PipeSlot {
    DevBuf probs, meta, labels;
    void* h_meta = nullptr;
    size_t h_meta_cap = 0;
    void* h_out = nullptr;
    size_t h_out_cap = 0;
    hipEvent_t dec_done = nullptr;
    bool busy = false;      // decode launched, labels not yet delivered
    int T = 0, W = 0, nwin = 0;
    int f16 = 0;            // the slot's probability rows are _Float16 (rd_set_logits)
    int64_t rows = 0;       // probability rows produced into this slot so far
    std::vector<PipeSub> subs;
    std::vector<int64_t> off1, off2;   // per window: source rows (see DecodeArgs)
    std::vector<int32_t> split, valid;
    unsigned lane_mask = 0; // forward lanes that produced rows of the open group
};

struct Pipe {
    hipStream_t s_dec = nullptr;
    PipeSlot slot[2];
    int cur = 0;
    int group = 4;          // batches per decode launch
    int lanes = 2;          // forward lanes the submitted batches rotate over
    int next_lane = 0;
};

int pipe_get(rd_ctx* ctx, Pipe** out)
{
    if (!ctx->pipe) {
        Pipe* p = new Pipe();
        int lo = 0, hi = 0;
        RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        RD_HIP(hipStreamCreateWithPriority(&p->s_dec, hipStreamNonBlocking, hi));
        for (int i = 0; i < 2; i++) {
            RD_HIP(hipEventCreateWithFlags(&p->slot[i].dec_done, hipEventDisableTiming));
        }
        ctx->pipe = p;
    }
    *out = (Pipe*)ctx->pipe;
    return RD_OK;
}

void pipe_reset(PipeSlot& s)
{
    s.subs.clear();
    s.off1.clear();
    s.off2.clear();
    s.split.clear();
    s.valid.clear();
    s.nwin = 0;
    s.rows = 0;
    s.lane_mask = 0;
}

int pipe_collect(PipeSlot& s)
{
    if (!s.busy) return RD_OK;
    RD_HIP(hipEventSynchronize(s.dec_done));
    const size_t nT = (size_t)s.nwin * s.T;
    const uint8_t* hl = (const uint8_t*)s.h_out;
    const int32_t* hlen = (const int32_t*)((const char*)s.h_out + align_up(nT, 256));
    s.busy = false;
    int rc = RD_OK;
    for (const PipeSub& sb : s.subs)
        for (int i = 0; i < sb.n; i++) {
            const int w = sb.win0 + i;
            if (hlen[w] < 0 || hlen[w] > s.T) {
                rd_set_error("pipeline: window %d produced an impossible label length %d", w, hlen[w]);
                rc = RD_ERR_STATE;
                continue;
            }
            sb.user_lens[i] = hlen[w];
            if (hlen[w]) memcpy(sb.user_labels + (size_t)i * s.T, hl + (size_t)w * s.T, (size_t)hlen[w]);
        }
    pipe_reset(s);
    return rc;
}

// launch the beam search + copy-out of everything forwarded into the slot so far
int pipe_launch_decode(rd_ctx* ctx, Pipe* p, PipeSlot& s)
{
    if (s.nwin == 0 || s.busy) return RD_OK;
    int rc;
    const size_t n = (size_t)s.nwin, nT = n * s.T;
    // metadata: [off1 | off2 | node_off | label_off] int64, then [seq_len | split] int32, then label_len int32 (device only)
    const size_t a8 = align_up(n * 8, 256), a4 = align_up(n * 4, 256);
    const size_t o_off2 = a8, o_node = 2 * a8, o_lab = 3 * a8, o_len = 4 * a8, o_split = o_len + a4, o_llen = o_split + a4;
    const size_t meta_bytes = o_llen + a4;
    if ((rc = pinned_reserve(&s.h_meta, &s.h_meta_cap, meta_bytes))) return rc;
    if (s.meta.reserve(meta_bytes)) return RD_ERR_NOMEM;
    char* hm = (char*)s.h_meta;
    int64_t* h_node = (int64_t*)(hm + o_node);
    int64_t* h_lab = (int64_t*)(hm + o_lab);
    memcpy(hm, s.off1.data(), n * 8);
    memcpy(hm + o_off2, s.off2.data(), n * 8);
    memcpy(hm + o_len, s.valid.data(), n * 4);
    memcpy(hm + o_split, s.split.data(), n * 4);
    for (int i = 0; i < s.nwin; i++) h_lab[i] = (int64_t)i * s.T;
    std::vector<TrieRun> runs;
    rd_plan_trie_runs(ctx, s.W, 0, s.nwin, [&](int k) { return (int64_t)s.valid[k]; }, h_node, runs);
    if (s.labels.reserve(nT + 16)) return RD_ERR_NOMEM;
    if ((rc = pinned_reserve(&s.h_out, &s.h_out_cap, align_up(nT, 256) + n * 4))) return rc;
    if ((rc = rd_rpipe_drain_decode(ctx))) return rc;   // (beam searches of the other pipeline use the same trie workspace)
    for (int l = 0; l < RD_MAX_LANES; l++)   // every forward that wrote into this slot has finished (a lane's event is its latest forward)
        if (s.lane_mask & (1u << l)) RD_HIP(hipStreamWaitEvent(p->s_dec, ctx->lanes[l].done, 0));
    RD_HIP(hipMemcpyAsync(s.meta.p, s.h_meta, o_llen, hipMemcpyHostToDevice, p->s_dec));
    char* dm = (char*)s.meta.p;
    for (const TrieRun& r : runs) {
        rc = rd_decode_dev(ctx, s.probs.p, s.f16 ? 2 : 0, (const int64_t*)dm + r.k0, (const int32_t*)(dm + o_len) + r.k0, (const int64_t*)(dm + o_node) + r.k0,
                           (const int64_t*)(dm + o_lab) + r.k0, r.k1 - r.k0, r.nodes, s.W, 0, 0.0, 0.0, s.labels.as<uint8_t>(),
                           (int32_t*)(dm + o_llen) + r.k0, nullptr, p->s_dec, (const int64_t*)(dm + o_off2) + r.k0, (const int32_t*)(dm + o_split) + r.k0);
        if (rc) return rc;
    }
    RD_HIP(hipMemcpyAsync(s.h_out, s.labels.p, nT, hipMemcpyDeviceToHost, p->s_dec));
    RD_HIP(hipMemcpyAsync((char*)s.h_out + align_up(nT, 256), dm + o_llen, n * 4, hipMemcpyDeviceToHost, p->s_dec));
    RD_HIP(hipEventRecord(s.dec_done, p->s_dec));
    s.busy = true;
    return RD_OK;
}

// slot that can take `rows` more probability rows for windows of T rows decoded at width W; closes / recycles groups
int pipe_open_slot(rd_ctx* ctx, Pipe* p, int T, int W, int64_t rows, PipeSlot** out, int f16 = 0)
{
    int rc;
    PipeSlot* s = &p->slot[p->cur];
    // a group is homogeneous in chunk_len, beam width and row type and bounded in size; otherwise close it and move on
    if (s->nwin > 0 && (s->T != T || s->W != W || s->f16 != f16 || (int)s->subs.size() >= p->group)) {
        if ((rc = pipe_launch_decode(ctx, p, *s))) return rc;
        p->cur ^= 1;
        s = &p->slot[p->cur];
    }
    if (s->busy && (rc = pipe_collect(*s))) return rc;   // the slot's previous group goes to its callers first
    const size_t need = (size_t)(s->rows + rows) * 20;
    if (need > s->probs.cap) {
        if (s->nwin > 0) {
            // growing would move probabilities already produced: close the group instead
            if ((rc = pipe_launch_decode(ctx, p, *s))) return rc;
            p->cur ^= 1;
            s = &p->slot[p->cur];
            if (s->busy && (rc = pipe_collect(*s))) return rc;
        }
        if (s->probs.reserve((size_t)rows * 20 * (size_t)p->group)) return RD_ERR_NOMEM;
    }
    s->T = T;
    s->W = W;
    s->f16 = f16;
    *out = s;
    return RD_OK;
}

int pipe_close_if_full(rd_ctx* ctx, Pipe* p, PipeSlot* s)
{
    if ((int)s->subs.size() >= p->group) {
        int rc = pipe_launch_decode(ctx, p, *s);
        if (rc) return rc;
        p->cur ^= 1;
    }
    return RD_OK;
}

void pipe_destroy(rd_ctx* ctx)
{
    Pipe* p = (Pipe*)ctx->pipe;
    if (!p) return;
    if (p->s_dec) (void)hipStreamSynchronize(p->s_dec);
    for (int i = 0; i < 2; i++) {
        PipeSlot& s = p->slot[i];
        s.probs.release();
        s.meta.release();
        s.labels.release();
        if (s.h_meta) (void)hipHostFree(s.h_meta);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.dec_done) (void)hipEventDestroy(s.dec_done);
    }
    if (p->s_dec) (void)hipStreamDestroy(p->s_dec);
    delete p;
    ctx->pipe = nullptr;
}

}  // namespace

void rd_pipe_destroy_internal(rd_ctx* ctx) { pipe_destroy(ctx); }

int rd_pipe_drain_decode_internal(rd_ctx* ctx)
{
    Pipe* p = (Pipe*)ctx->pipe;
    if (p && p->s_dec && (p->slot[0].busy || p->slot[1].busy)) RD_HIP(hipStreamSynchronize(p->s_dec));
    return RD_OK;
}

extern "C" int rd_pipe_config(rd_ctx* ctx, int group_batches)
{
    RD_REQUIRE(ctx, "rd_pipe_config: null context");
    RD_REQUIRE(group_batches >= 1 && group_batches <= 64, "rd_pipe_config: group_batches %d out of range [1,64]", group_batches);
    RD_HIP(hipSetDevice(ctx->device));
    Pipe* p = nullptr;
    int rc = pipe_get(ctx, &p);
    if (rc) return rc;
    RD_REQUIRE(p->slot[0].nwin == 0 && p->slot[1].nwin == 0 && !p->slot[0].busy && !p->slot[1].busy && rd_rpipe_idle(ctx),
               "rd_pipe_config: pipeline not empty (call rd_pipe_flush first)");
    p->group = group_batches;
    ctx->pipe_group = group_batches;
    return RD_OK;
}

extern "C" int rd_pipe_set_lanes(rd_ctx* ctx, int lanes)
{
    RD_REQUIRE(ctx, "rd_pipe_set_lanes: null context");
    RD_REQUIRE(lanes >= 1 && lanes <= RD_MAX_LANES, "rd_pipe_set_lanes: %d out of range [1,%d]", lanes, RD_MAX_LANES);
    Pipe* p = nullptr;
    int rc = pipe_get(ctx, &p);
    if (rc) return rc;
    RD_REQUIRE(p->slot[0].nwin == 0 && p->slot[1].nwin == 0 && !p->slot[0].busy && !p->slot[1].busy && rd_rpipe_idle(ctx),
               "rd_pipe_set_lanes: pipeline not empty (call rd_pipe_flush first)");
    p->lanes = lanes;
    p->next_lane = 0;
    ctx->pipe_lanes = lanes;
    return RD_OK;
}

extern "C" int rd_pipe_submit(rd_ctx* ctx, const float* d_windows, int n_windows, int chunk_len, const int32_t* valid_len,
                              int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    RD_REQUIRE(ctx && d_windows && valid_len && labels_out && label_len, "rd_pipe_submit: null argument");
    RD_REQUIRE(n_windows >= 1 && chunk_len >= 1, "rd_pipe_submit: bad shape");
    RD_REQUIRE(beam_width >= 1 && beam_width <= rd_decode_max_width(), "beam_width %d out of range", beam_width);
    RD_REQUIRE(rd_decode_len_ok(beam_width, chunk_len), "rd_pipe_submit: chunk_len %d too long for beam width %d (1 + W * rows < 2^29)", chunk_len, beam_width);
    for (int i = 0; i < n_windows; i++)
        RD_REQUIRE(valid_len[i] >= 0 && valid_len[i] <= chunk_len, "valid_len[%d]=%d out of range", i, valid_len[i]);
    RD_HIP(hipSetDevice(ctx->device));
    Pipe* p = nullptr;
    int rc = pipe_get(ctx, &p);
    if (rc) return rc;
    PipeSlot* s = nullptr;
    const int64_t rows = (int64_t)n_windows * chunk_len;
    if ((rc = pipe_open_slot(ctx, p, chunk_len, beam_width, rows, &s))) return rc;
    const int lane = p->next_lane;
    p->next_lane = (p->next_lane + 1) % p->lanes;
    rc = rd_forward_dev(ctx, d_windows, n_windows, chunk_len, s->probs.as<float>() + (size_t)s->rows * 5, lane);
    if (rc) return rc;
    s->lane_mask |= 1u << lane;
    PipeSub sb;
    sb.n = n_windows;
    sb.win0 = s->nwin;
    sb.user_labels = labels_out;
    sb.user_lens = label_len;
    s->subs.push_back(sb);
    for (int i = 0; i < n_windows; i++) {
        s->off1.push_back(s->rows + (int64_t)i * chunk_len);
        s->off2.push_back(s->rows + (int64_t)i * chunk_len);
        s->split.push_back(0);
        s->valid.push_back(valid_len[i]);
    }
    s->nwin += n_windows;
    s->rows += rows;
    return pipe_close_if_full(ctx, p, s);
}

extern "C" int rd_pipe_flush(rd_ctx* ctx)
{
    RD_REQUIRE(ctx, "rd_pipe_flush: null context");
    RD_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = rd_rpipe_flush(ctx))) return rc;   // the reads-level pipeline (pipe_reads.hip)
    Pipe* p = (Pipe*)ctx->pipe;
    if (!p) return RD_OK;
    // order of completion on the decode stream: the other slot's group (if any) was launched first
    PipeSlot& a = p->slot[p->cur ^ 1];
    PipeSlot& b = p->slot[p->cur];
    if ((rc = pipe_launch_decode(ctx, p, b))) return rc;
    if ((rc = pipe_collect(a))) return rc;
    return pipe_collect(b);
}

// --------------------------------------------------------------------------------------------- reads-level entry points
namespace {

int get_plan(rd_ctx* ctx, const int64_t* read_off, int n_reads, int chunk, int step, int mode, const ReadsPlan** out,
             const TileLists** lists, bool* streamed)
{
    PlanCache* pc = (PlanCache*)ctx->plan_cache[mode];
    if (!pc) {
        pc = new PlanCache();
        ctx->plan_cache[mode] = pc;
    }
    const int halo = rd_model_halo(ctx);
    std::vector<int64_t> lens(n_reads);
    for (int r = 0; r < n_reads; r++) lens[r] = read_off[r + 1] - read_off[r];
    bool hit = pc->chunk == chunk && pc->step == step && pc->halo == halo && pc->mode == mode && pc->lens == lens &&
               read_off[0] == 0 && pc->d_tiles.p && pc->nblocks == ctx->model.nblocks;
    for (int b = 0; hit && b < ctx->model.nblocks; b++) hit = pc->dil[b] == ctx->model.dil[b];
    if (!hit) {
        RD_REQUIRE(read_off[0] == 0, "read_off[0] must be 0");
        pc->chunk = -1;   // the cached key is void from here on: a failure below must not leave a half-built plan reachable
        pc->lens.clear();
        pc->plan = ReadsPlan();
        int rc = mode == 0 ? plan_reads_chunk(ctx->model, read_off, n_reads, chunk, step, halo, pc->plan)
                           : plan_reads_global(ctx->model, read_off, n_reads, chunk, step, halo, pc->plan, &pc->streamed);
        if (rc) return rc;
        ReadsPlan& P = pc->plan;
        auto pad4 = [](std::vector<TileDesc>& v) {          // pad the last workgroup tile with empty sub-tiles (to eight: the bf16x3 kernel's tile)
            while (v.size() % 8) {
                TileDesc e = {};
                e.alt_in = e.alt_res = INT32_MAX;
                v.push_back(e);
            }
        };
        size_t total = 0;
        for (int li = 0; li < P.n_layers; li++) {
            if (P.per_layer || li == 0) {
                pad4(P.tiles[li]);
                total += P.tiles[li].size();
            }
        }
        if (pc->d_tiles.reserve(total * sizeof(TileDesc) + 16)) return RD_ERR_NOMEM;
        // make sure no forward still reads the previous descriptors
        if ((rc = rd_sync_lanes(ctx))) return rc;
        size_t off = 0;
        for (int li = 0; li < RD_MAX_LAYERS; li++) {
            pc->lists.d[li] = nullptr;
            pc->lists.n[li] = 0;
            pc->lists.rows[li] = 0;
        }
        auto upload = [&](const std::vector<TileDesc>& v, const TileDesc** dptr, int* n) -> int {
            if (!v.empty())
                RD_HIP(hipMemcpy(pc->d_tiles.as<TileDesc>() + off, v.data(), v.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
            *dptr = pc->d_tiles.as<TileDesc>() + off;
            *n = (int)(v.size() / 4);
            off += v.size();
            return RD_OK;
        };
        for (int li = 0; li < P.n_layers; li++) {
            if (P.per_layer || li == 0) {
                if ((rc = upload(P.tiles[li], &pc->lists.d[li], &pc->lists.n[li]))) return rc;
                pc->lists.rows[li] = P.rows[li];
            } else {
                pc->lists.d[li] = pc->lists.d[0];
                pc->lists.n[li] = pc->lists.n[0];
                pc->lists.rows[li] = pc->lists.rows[0];
            }
        }
        pc->chunk = chunk;
        pc->step = step;
        pc->halo = halo;
        pc->mode = mode;
        pc->lens = lens;
        pc->nblocks = ctx->model.nblocks;
        for (int b = 0; b < RD_MAX_BLOCKS; b++) pc->dil[b] = b < ctx->model.nblocks ? ctx->model.dil[b] : 0;
    }
    *out = &pc->plan;
    *lists = &pc->lists;
    if (streamed) *streamed = pc->streamed;
    return RD_OK;
}

int check_reads_args(rd_ctx* ctx, const void* signal, const int64_t* read_off, int n_reads, int chunk_len, int step, int W)
{
    RD_REQUIRE(ctx && signal && read_off, "null argument");
    RD_REQUIRE(n_reads >= 1 && chunk_len >= 1, "bad shape");
    RD_REQUIRE(step >= 1 && step <= chunk_len, "step %d must be in [1, chunk_len]", step);
    RD_REQUIRE(W >= 1 && W <= rd_decode_max_width(), "beam_width %d out of range", W);
    if (!ctx->model.loaded) {
        rd_set_error("no weights loaded (rd_load_weights)");
        return RD_ERR_STATE;
    }
    return RD_OK;
}

}  // namespace

void rd_plan_cache_destroy_internal(rd_ctx* ctx)
{
    for (int m = 0; m < 2; m++) {
        PlanCache* pc = (PlanCache*)ctx->plan_cache[m];
        if (pc) {
            pc->d_tiles.release();
            delete pc;
        }
        ctx->plan_cache[m] = nullptr;
    }
}

extern "C" int rd_basecall_reads_chunk_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads,
                                                int chunk_len, int step, int beam_width, uint8_t* labels_out,
                                                int32_t* label_len)
{
    int rc = check_reads_args(ctx, d_signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_len, "rd_basecall_reads_chunk: null output");
    RD_HIP(hipSetDevice(ctx->device));
    const ReadsPlan* P = nullptr;
    const TileLists* tl = nullptr;
    if ((rc = get_plan(ctx, read_off, n_reads, chunk_len, step, 0, &P, &tl, nullptr))) return rc;
    if (ctx->ws_probs.reserve((size_t)P->total_rows * 20)) return RD_ERR_NOMEM;
    const int f16 = ctx->logits_f16;
    rc = rd_forward_tiles_dev(ctx, d_signal, *tl, P->total_rows, ctx->ws_probs.p, 0, f16);
    if (rc) return rc;
    std::vector<int64_t> lab_off(P->n_windows);
    for (int w = 0; w < P->n_windows; w++) lab_off[w] = (int64_t)w * chunk_len;
    return decode_and_fetch(ctx, ctx->ws_probs.p, f16 ? 2 : 0, P->off1.data(), P->valid.data(), P->n_windows, beam_width, 0, 0.0, 0.0,
                            labels_out, lab_off.data(), label_len, nullptr, P->off2.data(), P->split.data());
}

// forward only, at the reads level: the streamed evaluation (every time step once + the window heads in chunk mode) of a batch of
// normalised reads resident in HBM, on forward lane `lane`, asynchronous.  The probability rows land in the context's workspace
// (row layout of the plan: DESIGN.md 4.6); *total_rows (nullable) receives their number.
extern "C" int rd_forward_reads_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len,
                                         int step, int decode_type, int lane, int64_t* total_rows)
{
    int rc = check_reads_args(ctx, d_signal, read_off, n_reads, chunk_len, step, 1);
    if (rc) return rc;
    RD_REQUIRE(decode_type == 0 || decode_type == 1, "rd_forward_reads_resident: decode_type %d (0 = chunk plan, 1 = global plan)", decode_type);
    RD_REQUIRE(lane >= 0 && lane < RD_MAX_LANES, "rd_forward_reads_resident: lane %d out of range [0,%d)", lane, RD_MAX_LANES);
    RD_HIP(hipSetDevice(ctx->device));
    const ReadsPlan* P = nullptr;
    const TileLists* tl = nullptr;
    if ((rc = get_plan(ctx, read_off, n_reads, chunk_len, step, decode_type, &P, &tl, nullptr))) return rc;
    if (ctx->ws_probs.reserve((size_t)P->total_rows * 20)) return RD_ERR_NOMEM;
    if (total_rows) *total_rows = P->total_rows;
    return rd_forward_tiles_dev(ctx, d_signal, *tl, P->total_rows, ctx->ws_probs.p, lane, ctx->logits_f16);
}

// sig_model.predict for the windows of whole reads (basecall.py:83-93) through the streamed evaluation: host signal in, window-shaped
// probabilities out -- bit-identical to rd_forward on the same reads' windows for the rows basecall.py:96 keeps.
extern "C" int rd_forward_reads(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len, int step,
                                float* probs_out, int64_t windows_cap, int64_t* n_windows)
{
    int rc = check_reads_args(ctx, signal, read_off, n_reads, chunk_len, step, 1);
    if (rc) return rc;
    RD_REQUIRE(probs_out && n_windows, "rd_forward_reads: null output");
    RD_REQUIRE(!ctx->logits_f16, "rd_forward_reads: float32 rows only (rd_set_logits 0)");
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)read_off[n_reads];
    if (ctx->ws_in.reserve(n * 4 + 16)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, signal, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const ReadsPlan* P = nullptr;
    const TileLists* tl = nullptr;
    if ((rc = get_plan(ctx, read_off, n_reads, chunk_len, step, 0, &P, &tl, nullptr))) return rc;
    *n_windows = P->n_windows;
    RD_REQUIRE(P->n_windows <= windows_cap, "rd_forward_reads: %d windows, room for %lld", P->n_windows, (long long)windows_cap);
    if (ctx->ws_probs.reserve((size_t)P->total_rows * 20)) return RD_ERR_NOMEM;
    if ((rc = rd_forward_tiles_dev(ctx, ctx->ws_in.as<float>(), *tl, P->total_rows, ctx->ws_probs.p, 0, 0))) return rc;
    const size_t nw = (size_t)P->n_windows, a8 = align_up(nw * 8, 256), a4 = align_up(nw * 4, 256);
    const size_t out_bytes = nw * chunk_len * 20;
    if (ctx->ws_misc.reserve(2 * a8 + 2 * a4) || ctx->ws_mat.reserve(out_bytes + 16)) return RD_ERR_NOMEM;
    char* dm = (char*)ctx->ws_misc.p;
    RD_HIP(hipMemcpyAsync(dm, P->off1.data(), nw * 8, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipMemcpyAsync(dm + a8, P->off2.data(), nw * 8, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipMemcpyAsync(dm + 2 * a8, P->split.data(), nw * 4, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipMemcpyAsync(dm + 2 * a8 + a4, P->valid.data(), nw * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = rd_gather_windows_dev(ctx->stream, ctx->ws_probs.as<float>(), (const int64_t*)dm, (const int64_t*)(dm + a8), (const int32_t*)(dm + 2 * a8),
                                    (const int32_t*)(dm + 2 * a8 + a4), P->n_windows, chunk_len, (float*)ctx->ws_mat.p)))
        return rc;
    RD_HIP(hipMemcpyAsync(probs_out, ctx->ws_mat.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

extern "C" int rd_basecall_reads_chunk(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len,
                                       int step, int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    int rc = check_reads_args(ctx, signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)read_off[n_reads];
    if (ctx->ws_in.reserve(n * 4 + 16)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, signal, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return rd_basecall_reads_chunk_resident(ctx, ctx->ws_in.as<float>(), read_off, n_reads, chunk_len, step, beam_width, labels_out,
                                            label_len);
}

namespace {

// assembly + decode of a batch of reads given the forward's probabilities (streamed or windowed row layout)
int global_finish(rd_ctx* ctx, const void* d_probs, int f16, bool streamed, const ReadsPlan& P, const int64_t* read_off, int n_reads,
                  int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                  const int64_t* label_off, int32_t* label_len)
{
    std::vector<int64_t> off64(n_reads), off32(n_reads);
    std::vector<int32_t> seq_len(n_reads);
    std::vector<int> is64(n_reads);
    int64_t rows64 = 0;
    for (int r = 0; r < n_reads; r++) {
        const int nW = P.read_win_off[r + 1] - P.read_win_off[r];
        const int pad = P.valid[r];
        const int64_t N = assembled_rows(nW, chunk_len, pad, step);
        RD_REQUIRE(N == read_off[r + 1] - read_off[r], "internal: assembled length mismatch for read %d", r);
        seq_len[r] = (int32_t)N;
        is64[r] = assembled_is_f64(nW, chunk_len, pad, step);
        off64[r] = rows64;
        off32[r] = P.read_row[r];   // single coverage: window rows (or stream rows) are consecutive time steps
        if (is64[r]) rows64 += N;
    }
    if (ctx->ws_mat.reserve((size_t)(rows64 + 1) * 40)) return RD_ERR_NOMEM;
    int rc;
    for (int r = 0; r < n_reads; r++) {
        if (!is64[r]) continue;
        const int nW = P.read_win_off[r + 1] - P.read_win_off[r];
        rc = rd_assemble_dev(ctx, (const char*)d_probs + (size_t)P.read_row[r] * 5 * (f16 ? 2 : 4), nW, chunk_len, P.valid[r], step,
                             ctx->ws_mat.as<double>() + off64[r] * 5, seq_len[r], streamed ? 1 : 0, f16);
        if (rc) return rc;
    }
    for (int pass = 0; pass < 2; pass++) {
        std::vector<int64_t> so, lo;
        std::vector<int32_t> sl;
        std::vector<int> idx;
        for (int r = 0; r < n_reads; r++)
            if (is64[r] == (pass == 0)) {
                so.push_back(pass == 0 ? off64[r] : off32[r]);
                sl.push_back(seq_len[r]);
                lo.push_back(label_off[r]);
                idx.push_back(r);
            }
        if (idx.empty()) continue;
        std::vector<int32_t> ll(idx.size());
        rc = decode_and_fetch(ctx, pass == 0 ? (const void*)ctx->ws_mat.p : d_probs, pass == 0 ? 1 : (f16 ? 2 : 0), so.data(), sl.data(),
                              (int)idx.size(), beam_width, use_lm, s_thr, r_thr, labels_out, lo.data(), ll.data(), nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < idx.size(); i++) label_len[idx[i]] = ll[i];
    }
    return RD_OK;
}

}  // namespace

extern "C" int rd_basecall_reads_global_resident(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads,
                                                 int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr,
                                                 uint8_t* labels_out, const int64_t* label_off, int32_t* label_len)
{
    int rc = check_reads_args(ctx, d_signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_off && label_len, "rd_basecall_reads_global: null output");
    RD_REQUIRE_WIDTH_LM(ctx, beam_width, use_lm);
    RD_HIP(hipSetDevice(ctx->device));
    const ReadsPlan* P = nullptr;
    const TileLists* tl = nullptr;
    bool streamed = false;
    if ((rc = get_plan(ctx, read_off, n_reads, chunk_len, step, 1, &P, &tl, &streamed))) return rc;
    if (ctx->ws_probs.reserve((size_t)P->total_rows * 20)) return RD_ERR_NOMEM;
    const int f16 = ctx->logits_f16;
    rc = rd_forward_tiles_dev(ctx, d_signal, *tl, P->total_rows, ctx->ws_probs.p, 0, f16);
    if (rc) return rc;
    return global_finish(ctx, ctx->ws_probs.p, f16, streamed, *P, read_off, n_reads, chunk_len, step, beam_width, use_lm, s_thr,
                         r_thr, labels_out, label_off, label_len);
}

extern "C" int rd_basecall_reads_global(rd_ctx* ctx, const float* signal, const int64_t* read_off, int n_reads, int chunk_len,
                                        int step, int beam_width, int use_lm, double s_thr, double r_thr, uint8_t* labels_out,
                                        const int64_t* label_off, int32_t* label_len)
{
    int rc = check_reads_args(ctx, signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)read_off[n_reads];
    if (ctx->ws_in.reserve(n * 4 + 16)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, signal, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return rd_basecall_reads_global_resident(ctx, ctx->ws_in.as<float>(), read_off, n_reads, chunk_len, step, beam_width, use_lm,
                                             s_thr, r_thr, labels_out, label_off, label_len);
}

// ---- raw int16 reads in: normalisation on the device, then the reads-level paths --------------------------------
namespace {

// uploads raw samples + offsets, runs mad_normalise_kernel into ctx->ws_in, returns per-read status on the host
int normalise_upload(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int clip, int32_t* status)
{
    RD_REQUIRE(ctx && raw && read_off && status, "null argument");
    RD_REQUIRE(n_reads >= 1 && read_off[0] == 0, "bad read offsets");
    for (int r = 0; r < n_reads; r++) RD_REQUIRE(read_off[r + 1] >= read_off[r], "read offsets must be non-decreasing");
    RD_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)read_off[n_reads];
    const size_t o_off = align_up(n * 2 + 16, 256), o_st = o_off + align_up((size_t)(n_reads + 1) * 8, 256);
    if (ctx->ws_raw.reserve(o_st + (size_t)n_reads * 4 + 16) || ctx->ws_in.reserve(n * 4 + 16)) return RD_ERR_NOMEM;
    char* base = (char*)ctx->ws_raw.p;
    if (n) RD_HIP(hipMemcpyAsync(base, raw, n * 2, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipMemcpyAsync(base + o_off, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = rd_normalise_dev(ctx, (const int16_t*)base, (const int64_t*)(base + o_off), n_reads, clip, ctx->ws_in.as<float>(),
                              (int32_t*)(base + o_st));
    if (rc) return rc;
    RD_HIP(hipMemcpyAsync(status, base + o_st, (size_t)n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

}  // namespace

extern "C" int rd_normalise_reads(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                                  float* norm_out, int32_t* status)
{
    int rc = normalise_upload(ctx, raw, read_off, n_reads, outlier_clip, status);
    if (rc) return rc;
    const size_t n = (size_t)read_off[n_reads];
    if (norm_out && n) {
        RD_HIP(hipMemcpyAsync(norm_out, ctx->ws_in.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RD_OK;
}

extern "C" int rd_basecall_raw_chunk(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                                     int chunk_len, int step, int beam_width, uint8_t* labels_out, int32_t* label_len,
                                     int32_t* status)
{
    int rc = normalise_upload(ctx, raw, read_off, n_reads, outlier_clip, status);
    if (rc) return rc;
    for (int r = 0; r < n_reads; r++)
        RD_REQUIRE(status[r] != 2, "rd_basecall_raw_chunk: read %d is empty (the caller skips empty reads, basecall.py:77-82)", r);
    return rd_basecall_reads_chunk_resident(ctx, ctx->ws_in.as<float>(), read_off, n_reads, chunk_len, step, beam_width, labels_out,
                                            label_len);
}

extern "C" int rd_basecall_raw_global(rd_ctx* ctx, const int16_t* raw, const int64_t* read_off, int n_reads, int outlier_clip,
                                      int chunk_len, int step, int beam_width, int use_lm, double s_thr, double r_thr,
                                      uint8_t* labels_out, const int64_t* label_off, int32_t* label_len, int32_t* status)
{
    int rc = normalise_upload(ctx, raw, read_off, n_reads, outlier_clip, status);
    if (rc) return rc;
    for (int r = 0; r < n_reads; r++)
        RD_REQUIRE(status[r] != 2, "rd_basecall_raw_global: read %d is empty (the caller skips empty reads, basecall.py:77-82)", r);
    return rd_basecall_reads_global_resident(ctx, ctx->ws_in.as<float>(), read_off, n_reads, chunk_len, step, beam_width, use_lm,
                                             s_thr, r_thr, labels_out, label_off, label_len);
}

// pipelined chunk-mode batches of whole reads (same overlap scheme as rd_pipe_submit)
extern "C" int rd_pipe_submit_reads(rd_ctx* ctx, const float* d_signal, const int64_t* read_off, int n_reads, int chunk_len,
                                    int step, int beam_width, uint8_t* labels_out, int32_t* label_len)
{
    int rc = check_reads_args(ctx, d_signal, read_off, n_reads, chunk_len, step, beam_width);
    if (rc) return rc;
    RD_REQUIRE(labels_out && label_len, "rd_pipe_submit_reads: null output");
    RD_REQUIRE(rd_decode_len_ok(beam_width, chunk_len), "rd_pipe_submit_reads: chunk_len %d too long for beam width %d (1 + W * rows < 2^29)", chunk_len, beam_width);
    RD_HIP(hipSetDevice(ctx->device));
    const ReadsPlan* P = nullptr;
    const TileLists* tl = nullptr;
    if ((rc = get_plan(ctx, read_off, n_reads, chunk_len, step, 0, &P, &tl, nullptr))) return rc;
    Pipe* p = nullptr;
    if ((rc = pipe_get(ctx, &p))) return rc;
    PipeSlot* s = nullptr;
    const int f16 = ctx->logits_f16;
    if ((rc = pipe_open_slot(ctx, p, chunk_len, beam_width, P->total_rows, &s, f16))) return rc;
    const int lane = p->next_lane;
    p->next_lane = (p->next_lane + 1) % p->lanes;
    rc = rd_forward_tiles_dev(ctx, d_signal, *tl, P->total_rows, (char*)s->probs.p + (size_t)s->rows * 5 * (f16 ? 2 : 4), lane, f16);
    if (rc) return rc;
    s->lane_mask |= 1u << lane;
    PipeSub sb;
    sb.n = P->n_windows;
    sb.win0 = s->nwin;
    sb.user_labels = labels_out;
    sb.user_lens = label_len;
    s->subs.push_back(sb);
    for (int w = 0; w < P->n_windows; w++) {
        s->off1.push_back(P->off1[w] + s->rows);
        s->off2.push_back(P->off2[w] + s->rows);
        s->split.push_back(P->split[w]);
        s->valid.push_back(P->valid[w]);
    }
    s->nwin += P->n_windows;
    s->rows += P->total_rows;
    return pipe_close_if_full(ctx, p, s);
}

// --------------------------------------------------------------------------------------------- device memory
extern "C" int rd_dev_alloc(rd_ctx* ctx, size_t bytes, void** d_ptr)
{
    RD_REQUIRE(ctx && d_ptr, "rd_dev_alloc: null argument");
    RD_HIP(hipSetDevice(ctx->device));
    RD_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return RD_OK;
}
extern "C" int rd_mem_info(rd_ctx* ctx, size_t* free_bytes, size_t* total_bytes)
{
    RD_REQUIRE(ctx && free_bytes && total_bytes, "rd_mem_info: null argument");
    RD_HIP(hipSetDevice(ctx->device));
    RD_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return RD_OK;
}
extern "C" int rd_dev_free(rd_ctx* ctx, void* d_ptr)
{
    RD_REQUIRE(ctx, "rd_dev_free: null context");
    if (d_ptr) RD_HIP(hipFree(d_ptr));
    return RD_OK;
}
extern "C" int rd_memcpy_h2d(rd_ctx* ctx, void* d_dst, const void* src, size_t bytes)
{
    RD_REQUIRE(ctx && d_dst && src, "rd_memcpy_h2d: null argument");
    RD_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}
extern "C" int rd_memcpy_d2h(rd_ctx* ctx, void* dst, const void* d_src, size_t bytes)
{
    RD_REQUIRE(ctx && dst && d_src, "rd_memcpy_d2h: null argument");
    RD_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

extern "C" int rd_split3(rd_ctx* ctx, const float* values, size_t n, uint16_t* terms_out)
{
    RD_REQUIRE(ctx && values && terms_out, "rd_split3: null argument");
    RD_HIP(hipSetDevice(ctx->device));
    if (n == 0) return RD_OK;
    if (ctx->ws_in.reserve(n * 4) || ctx->ws_misc.reserve(n * 6)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(ctx->ws_in.p, values, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rd_split3_dev(ctx, ctx->ws_in.as<float>(), n, ctx->ws_misc.as<uint16_t>());
    if (rc) return rc;
    RD_HIP(hipMemcpyAsync(terms_out, ctx->ws_misc.p, n * 6, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

// --------------------------------------------------------------------------------------------- timers
static KernelTimer* timer_of(rd_ctx* ctx, int which)
{
    switch (which) {
        case RD_TIMER_CONV: return &ctx->timer_conv;
        case RD_TIMER_DECODE: return &ctx->timer_decode;
        case RD_TIMER_HEAD: return &ctx->timer_head;
        case RD_TIMER_IN: return &ctx->timer_in;
    }
    return nullptr;
}

extern "C" int rd_timer_enable(rd_ctx* ctx, int which, int max_launches)
{
    RD_REQUIRE(ctx, "rd_timer_enable: null context");
    KernelTimer* t = timer_of(ctx, which);
    RD_REQUIRE(t, "rd_timer_enable: unknown timer %d", which);
    RD_HIP(hipSetDevice(ctx->device));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    timer_free(*t);
    t->flops = t->bytes = 0.0;
    if (max_launches <= 0) return RD_OK;
    t->starts.resize(max_launches);
    t->stops.resize(max_launches);
    for (int i = 0; i < max_launches; i++) {
        RD_HIP(hipEventCreate(&t->starts[i]));
        RD_HIP(hipEventCreate(&t->stops[i]));
    }
    t->enabled = true;
    return RD_OK;
}

extern "C" int rd_timer_read(rd_ctx* ctx, int which, double* total_ms, int* launches, double* flops, double* bytes)
{
    RD_REQUIRE(ctx, "rd_timer_read: null context");
    KernelTimer* t = timer_of(ctx, which);
    RD_REQUIRE(t, "rd_timer_read: unknown timer %d", which);
    RD_HIP(hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (size_t i = 0; i < t->used; i++) {
        float f = 0.f;
        RD_HIP(hipEventElapsedTime(&f, t->starts[i], t->stops[i]));
        ms += f;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int)t->used;
    if (flops) *flops = t->flops;
    if (bytes) *bytes = t->bytes;
    return RD_OK;
}

extern "C" int rd_timer_read_launches(rd_ctx* ctx, int which, int cap, float* ms_out, double* flops_out, int32_t* tag_out, int* n_out)
{
    RD_REQUIRE(ctx && n_out, "rd_timer_read_launches: null argument");
    KernelTimer* t = timer_of(ctx, which);
    RD_REQUIRE(t, "rd_timer_read_launches: unknown timer %d", which);
    RD_HIP(hipStreamSynchronize(ctx->stream));
    const size_t n = std::min(t->used, (size_t)std::max(0, cap));
    for (size_t i = 0; i < n; i++) {
        float f = 0.f;
        RD_HIP(hipEventElapsedTime(&f, t->starts[i], t->stops[i]));
        if (ms_out) ms_out[i] = f;
        if (flops_out) flops_out[i] = i < t->each_flops.size() ? t->each_flops[i] : 0.0;
        if (tag_out) tag_out[i] = i < t->each_tag.size() ? t->each_tag[i] : 0;
    }
    *n_out = (int)n;
    return RD_OK;
}

// --------------------------------------------------------------------------------------------- RCCL
namespace {

struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.h) return RD_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        rd_set_error("cannot dlopen librccl: %s", dlerror());
        return RD_ERR_RCCL;
    }
    g_rccl.h = h;
#define RD_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                     \
    if (!g_rccl.field) {                                                           \
        rd_set_error("librccl lacks symbol %s", name);                             \
        g_rccl.h = nullptr;                                                        \
        return RD_ERR_RCCL;                                                        \
    }
    RD_SYM(GetUniqueId, "ncclGetUniqueId");
    RD_SYM(CommInitRank, "ncclCommInitRank");
    RD_SYM(CommDestroy, "ncclCommDestroy");
    RD_SYM(CommCount, "ncclCommCount");
    RD_SYM(Broadcast, "ncclBroadcast");
    RD_SYM(AllReduce, "ncclAllReduce");
    RD_SYM(GetErrorString, "ncclGetErrorString");
#undef RD_SYM
    return RD_OK;
}

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    DevBuf scratch;
};

#define RD_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            rd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
            return RD_ERR_RCCL;                                                                    \
        }                                                                                          \
    } while (0)

struct BcastHeader {
    int32_t model_loaded, nblocks, dil[RD_MAX_BLOCKS];
    int32_t lm_loaded, lm_k, lm_order, lm_hashed, lm_sparse;
    int64_t model_floats, lm_doubles;
    float inv_scale[2 * RD_MAX_BLOCKS], inv_scale_d1;
};

}  // namespace

extern "C" int rd_rccl_probe(void) { return rccl_load(); }

extern "C" int rd_rccl_unique_id(uint8_t id_out[128])
{
    RD_REQUIRE(id_out, "rd_rccl_unique_id: null argument");
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RD_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, 128);
    return RD_OK;
}

extern "C" int rd_rccl_init(rd_ctx* ctx, int rank, int nranks, const uint8_t id[128])
{
    RD_REQUIRE(ctx && id, "rd_rccl_init: null argument");
    RD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rd_rccl_init: bad rank %d of %d", rank, nranks);
    int rc = rccl_load();
    if (rc) return rc;
    RD_HIP(hipSetDevice(ctx->device));
    if (ctx->rccl) rd_rccl_finalize(ctx);
    RcclState* st = new RcclState();
    st->rank = rank;
    st->nranks = nranks;
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclResult_t r = g_rccl.CommInitRank(&st->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        rd_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
        delete st;
        return RD_ERR_RCCL;
    }
    ctx->rccl = st;
    return RD_OK;
}

extern "C" int rd_rccl_finalize(rd_ctx* ctx)
{
    if (!ctx || !ctx->rccl) return RD_OK;
    RcclState* st = (RcclState*)ctx->rccl;
    if (st->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(st->comm);
    st->scratch.release();
    delete st;
    ctx->rccl = nullptr;
    return RD_OK;
}

// What a receiver must know before the device images arrive (the sender's side of rd_rccl_bcast_model / rd_clone_artifacts)
static void artifacts_header(const rd_ctx* ctx, BcastHeader& hd)
{
    hd = BcastHeader{};
    hd.model_loaded = 1;
    hd.nblocks = ctx->model.nblocks;
    for (int i = 0; i < RD_MAX_BLOCKS; i++) hd.dil[i] = ctx->model.dil[i];
    hd.model_floats = (int64_t)model_layout(ctx->model.nblocks).total;
    for (int i = 0; i < 2 * RD_MAX_BLOCKS; i++) hd.inv_scale[i] = ctx->model.inv_scale[i];
    hd.inv_scale_d1 = ctx->model.inv_scale_d1;
    hd.lm_loaded = ctx->lm.loaded ? 1 : 0;
    hd.lm_k = ctx->lm.k;
    hd.lm_order = ctx->lm.table_order;
    hd.lm_hashed = ctx->lm.hashed;
    hd.lm_sparse = ctx->lm.sparse;
    hd.lm_doubles = ctx->lm.loaded ? (int64_t)lm_image_doubles(ctx->lm.table_order) : 0;
}

// The receiver's side: geometry and scales from the header, storage reserved and bound; the images are not there yet
// (artifacts_arrived marks them loaded).
static int artifacts_prepare(rd_ctx* ctx, const BcastHeader& hd)
{
    RD_REQUIRE(hd.model_loaded == 1 && hd.nblocks >= 1 && hd.nblocks <= RD_MAX_BLOCKS && hd.model_floats == (int64_t)model_layout(hd.nblocks).total,
               "artefact header: %d blocks, %lld floats do not describe a model of this library", hd.nblocks, (long long)hd.model_floats);
    Model& m = ctx->model;
    m.loaded = false;
    m.nblocks = hd.nblocks;
    for (int i = 0; i < RD_MAX_BLOCKS; i++) m.dil[i] = hd.dil[i];
    for (int i = 0; i < 2 * RD_MAX_BLOCKS; i++) m.inv_scale[i] = hd.inv_scale[i];
    m.inv_scale_d1 = hd.inv_scale_d1;
    if (m.storage.reserve((size_t)hd.model_floats * 4)) return RD_ERR_NOMEM;
    model_bind(m, model_layout(m.nblocks));
    ctx->lm.loaded = false;
    ctx->lm.gate_valid = false;
    if (hd.lm_loaded) {
        RD_REQUIRE(hd.lm_order >= 1 && hd.lm_order <= 13 && hd.lm_doubles == (int64_t)lm_image_doubles(hd.lm_order),
                   "artefact header: LM table of order %d with %lld doubles", hd.lm_order, (long long)hd.lm_doubles);
        ctx->lm.k = hd.lm_k;
        ctx->lm.table_order = hd.lm_order;
        ctx->lm.hashed = hd.lm_hashed;
        ctx->lm.sparse = hd.lm_sparse;
        if (ctx->lm.storage.reserve((size_t)hd.lm_doubles * 8)) return RD_ERR_NOMEM;
        lm_bind(ctx->lm);
    }
    return RD_OK;
}

static void artifacts_arrived(rd_ctx* ctx, const BcastHeader& hd)
{
    ctx->model.loaded = true;
    if (hd.lm_loaded) ctx->lm.loaded = true;
}

extern "C" int rd_rccl_bcast_model(rd_ctx* ctx, int root)
{
    RD_REQUIRE(ctx && ctx->rccl, "rd_rccl_bcast_model: rd_rccl_init not called");
    RcclState* st = (RcclState*)ctx->rccl;
    RD_REQUIRE(root >= 0 && root < st->nranks, "rd_rccl_bcast_model: bad root");
    RD_HIP(hipSetDevice(ctx->device));
    BcastHeader hd = {};
    if (st->rank == root) {
        RD_REQUIRE(ctx->model.loaded, "rd_rccl_bcast_model: root has no weights loaded");
        artifacts_header(ctx, hd);
    }
    if (st->scratch.reserve(sizeof(BcastHeader))) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(st->scratch.p, &hd, sizeof(hd), hipMemcpyHostToDevice, ctx->stream));
    RD_NCCL(g_rccl.Broadcast(st->scratch.p, st->scratch.p, sizeof(hd), ncclUint8, root, st->comm, ctx->stream));
    RD_HIP(hipMemcpyAsync(&hd, st->scratch.p, sizeof(hd), hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    if (st->rank != root) {
        int rc = artifacts_prepare(ctx, hd);
        if (rc) return rc;
    }
    // one broadcast of the packed weights (8.8 MB) and, when present, one of the LM table + entropies
    RD_NCCL(g_rccl.Broadcast(ctx->model.storage.p, ctx->model.storage.p, (size_t)hd.model_floats, ncclFloat32, root, st->comm,
                             ctx->stream));
    if (hd.lm_loaded)
        RD_NCCL(g_rccl.Broadcast(ctx->lm.storage.p, ctx->lm.storage.p, (size_t)hd.lm_doubles, ncclFloat64, root, st->comm,
                                 ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    artifacts_arrived(ctx, hd);
    return RD_OK;
}

// A second context of the same process takes the device images of a loaded one (weights in all three packings, LM table,
// entropies): the receiver's code of rd_rccl_bcast_model with a device-to-device copy as the transport.  The driver's
// extra contexts of a GPU use it instead of parsing and repacking the artefacts again; peer copies make it work across the
// GPUs of one process too.
extern "C" int rd_clone_artifacts(rd_ctx* dst, rd_ctx* src)
{
    RD_REQUIRE(dst && src && dst != src, "rd_clone_artifacts: two distinct contexts are needed");
    RD_REQUIRE(src->model.loaded, "rd_clone_artifacts: the source context has no weights loaded");
    BcastHeader hd;
    artifacts_header(src, hd);
    RD_HIP(hipSetDevice(src->device));
    RD_HIP(hipStreamSynchronize(src->stream));
    RD_HIP(hipSetDevice(dst->device));
    int rc = artifacts_prepare(dst, hd);
    if (rc) return rc;
    RD_HIP(hipMemcpyAsync(dst->model.storage.p, src->model.storage.p, (size_t)hd.model_floats * 4, hipMemcpyDefault, dst->stream));
    if (hd.lm_loaded)
        RD_HIP(hipMemcpyAsync(dst->lm.storage.p, src->lm.storage.p, (size_t)hd.lm_doubles * 8, hipMemcpyDefault, dst->stream));
    RD_HIP(hipStreamSynchronize(dst->stream));
    artifacts_arrived(dst, hd);
    return RD_OK;
}

extern "C" int rd_rccl_allreduce_max(rd_ctx* ctx, double* inout, int n)
{
    RD_REQUIRE(ctx && ctx->rccl && inout && n >= 1, "rd_rccl_allreduce_max: bad argument / rd_rccl_init not called");
    RcclState* st = (RcclState*)ctx->rccl;
    RD_HIP(hipSetDevice(ctx->device));
    if (st->scratch.reserve((size_t)n * 8 + 256)) return RD_ERR_NOMEM;
    RD_HIP(hipMemcpyAsync(st->scratch.p, inout, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    RD_NCCL(g_rccl.AllReduce(st->scratch.p, st->scratch.p, (size_t)n, ncclFloat64, ncclMax, st->comm, ctx->stream));
    RD_HIP(hipMemcpyAsync(inout, st->scratch.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RD_HIP(hipStreamSynchronize(ctx->stream));
    return RD_OK;
}

extern "C" int rd_rccl_comm_count(rd_ctx* ctx, int* nranks)
{
    RD_REQUIRE(ctx && ctx->rccl && nranks, "rd_rccl_comm_count: bad argument / rd_rccl_init not called");
    RcclState* st = (RcclState*)ctx->rccl;
    int n = 0;
    RD_NCCL(g_rccl.CommCount(st->comm, &n));
    *nranks = n;
    return RD_OK;
}

extern "C" int rd_rccl_barrier(rd_ctx* ctx)
{
    double v = 0.0;
    return rd_rccl_allreduce_max(ctx, &v, 1);
}
