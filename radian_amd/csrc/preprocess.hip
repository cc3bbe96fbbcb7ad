// preprocess.hip -- MAD normalisation of raw int16 reads on the device (the step before the hot path).
//
// Replaces `mad_normalise(raw_signal, outlier_clip)` -- radian/preprocess.py:24-49, called at radian/basecall.py:78 --
// followed by the float32 cast `sig_model.predict` applies to its input (basecall.py:91).  Windowing
// (get_windows, preprocess.py:4-22) needs no kernel: the reads-level forward forms windows through tile descriptors.
//
//   median = np.median(signal)                         exact: radix select on the int16 values
//   mad    = np.median(|signal - median|)              exact: radix select on the integers |2x - 2*median|
//   z      = (x - median) / (1.4826 * mad)             float64, IEEE mul/div = NumPy's
//   clip to +-outlier_clip; MAD == 0 -> "MAD is zero" (status 1), empty -> status 2 (the caller skips the read)
//   np.vectorize quirk: if the FIRST sample is clipped the whole result is int64, i.e. every
//   value is truncated toward zero before clipping.
//
// One 256-thread workgroup per read; the read (2 B per sample) is swept five times out of L2: two histogram passes
// per order statistic (high bits, then low bits inside the selected bin) and the final write.  Integer / byte work,
// bound by LDS atomics on a handful of bins for Gaussian-like signals, microseconds per batch.
#include "common.h"

namespace {

struct SelectResult {
    int v1, v2;  // the two middle order statistics (equal ranks when N is odd)
};

// k-th smallest (0-based ranks k1 <= k2) of key(i), keys in [0, 2^(HB+8)); two histogram passes.
template <int HB, typename KeyFn>
__device__ SelectResult radix_select2(int64_t n, int64_t k1, int64_t k2, KeyFn key, unsigned* hist /*[1<<HB] or [256]*/,
                                      int* sh /*[8]*/)
{
    constexpr int NH = 1 << HB;
    const int tid = threadIdx.x;
    // ---- pass 1: high bits
    for (int i = tid; i < NH; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += blockDim.x) atomicAdd(&hist[key(i) >> 8], 1u);
    __syncthreads();
    if (tid == 0) {
        int64_t c = 0;
        int b1 = -1, b2 = -1;
        int64_t r1 = 0, r2 = 0;
        for (int b = 0; b < NH; b++) {
            const int64_t h = hist[b];
            if (b1 < 0 && k1 < c + h) { b1 = b; r1 = k1 - c; }
            if (b2 < 0 && k2 < c + h) { b2 = b; r2 = k2 - c; }
            c += h;
        }
        sh[0] = b1; sh[1] = b2; sh[2] = (int)r1; sh[3] = (int)r2;
    }
    __syncthreads();
    const int b1 = sh[0], b2 = sh[1];
    const int r1 = sh[2], r2 = sh[3];
    // ---- pass 2: low 8 bits inside bin b1 (and b2 when it differs): hist[0..255] and hist[256..511]
    __syncthreads();
    for (int i = tid; i < 512; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += blockDim.x) {
        const int k = key(i);
        const int hb = k >> 8;
        if (hb == b1) atomicAdd(&hist[k & 255], 1u);
        if (hb == b2 && b2 != b1) atomicAdd(&hist[256 + (k & 255)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0, v1 = -1, v2 = -1;
        for (int b = 0; b < 256; b++) {
            c += (int)hist[b];
            if (v1 < 0 && r1 < c) v1 = (b1 << 8) | b;
            if (b2 == b1 && v2 < 0 && r2 < c) v2 = (b1 << 8) | b;
        }
        if (b2 != b1) {
            c = 0;
            for (int b = 0; b < 256; b++) {
                c += (int)hist[256 + b];
                if (v2 < 0 && r2 < c) v2 = (b2 << 8) | b;
            }
        }
        sh[4] = v1; sh[5] = v2;
    }
    __syncthreads();
    SelectResult r;
    r.v1 = sh[4];
    r.v2 = sh[5];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void mad_normalise_kernel(const int16_t* __restrict__ raw, const int64_t* __restrict__ read_off,
                                                            int clip, float* __restrict__ out, int32_t* __restrict__ status)
{
    __shared__ unsigned hist[512];
    __shared__ int sh[8];
    const int r = blockIdx.x;
    const int64_t o = read_off[r];
    const int64_t n = read_off[r + 1] - o;
    const int16_t* x = raw + o;
    float* y = out + o;
    if (n <= 0) {
        if (threadIdx.x == 0) status[r] = 2;  // "Signal must not be empty to normalise" (preprocess.py:25-26)
        return;
    }
    const int64_t k1 = (n - 1) / 2, k2 = n / 2;
    // median of the samples: keys = x + 32768 in [0, 65536)
    const SelectResult m = radix_select2<8>(n, k1, k2, [&](int64_t i) { return (int)x[i] + 32768; }, hist, sh);
    const int s2 = (m.v1 - 32768) + (m.v2 - 32768);           // 2 * median, exact integer
    // MAD: keys = |2x - 2*median| in [0, 131071] -> 9 high bits + 8 low bits
    const SelectResult d = radix_select2<9>(n, k1, k2, [&](int64_t i) { const int v = 2 * (int)x[i] - s2; return v < 0 ? -v : v; }, hist, sh);
    const double median = (double)s2 * 0.5;                     // np.median: mean of the two middle values
    const double mad = ((double)d.v1 * 0.5 + (double)d.v2 * 0.5) * 0.5;
    if (mad == 0.0) {
        if (threadIdx.x == 0) status[r] = 1;                    // "MAD is zero, issue with signal." (preprocess.py:47-48)
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) y[i] = 0.f;
        return;
    }
    const double denom = 1.4826 * mad;
    const double hi = (double)clip, lo = -(double)clip;
    const double z0 = ((double)x[0] - median) / denom;
    const bool int_quirk = (z0 > hi) || (z0 < lo);             // np.vectorize takes the dtype from the first output
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double z = ((double)x[i] - median) / denom;
        if (z > hi) z = hi;
        else if (z < lo) z = lo;
        else if (int_quirk) z = trunc(z);
        y[i] = (float)z;
    }
    if (threadIdx.x == 0) status[r] = 0;
}

}  // namespace

// d_raw [total] int16, d_read_off [n_reads+1] -> d_out [total] float32, d_status [n_reads]
int rd_normalise_dev(rd_ctx* ctx, const int16_t* d_raw, const int64_t* d_read_off, int n_reads, int clip, float* d_out,
                     int32_t* d_status, hipStream_t stream)
{
    if (n_reads == 0) return RD_OK;
    hipLaunchKernelGGL(mad_normalise_kernel, dim3(n_reads), dim3(256), 0, stream ? stream : ctx->stream, d_raw, d_read_off, clip, d_out, d_status);
    RD_HIP(hipGetLastError());
    return RD_OK;
}
