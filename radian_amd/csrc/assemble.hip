// assemble.hip -- global-mode stitch of overlapping window outputs into one [N,5] matrix.
//
// Replaces radian/matrix_assembly.py:6-53 (assemble_matrices) after the pad trim of
// radian/basecall.py:96.  For absolute time step t the reference stacks the rows of every window
// covering t (window i starts at i*step) and "averages" them -- but np.add's result is discarded
// (matrix_assembly.py:52), so the assembled row is the EARLIEST covering window's row,
// L1-normalised in float64 by sklearn.normalize when more than one window covers t, and the
// untouched float32 row otherwise.  This is a pure gather: 20 B read + 40 B written per row, one
// thread per row, fully coalesced on the write side.
#include "common.h"

namespace {

template <typename IT>
__global__ __launch_bounds__(256) void assemble_kernel(const IT* __restrict__ probs, int nW, int T, int pad, int step,
                                                        double* __restrict__ out, int64_t N, int streamed)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    // windows i with i*step <= t < i*step + rows_i ; rows_i = T except the last (T - pad)
    int64_t lo = t - T + 1;
    int i_min = lo <= 0 ? 0 : (int)((lo + step - 1) / step);
    int i_max = (int)(t / step);
    if (i_max > nW - 1) i_max = nW - 1;
    if (i_max == nW - 1 && t >= (int64_t)(nW - 1) * step + (T - pad)) i_max--;  // trimmed rows of the last window
    if (i_min > i_max) i_min = i_max;  // cannot happen for t < N
    // streamed forward: the earliest covering window's row of time step t IS the stream's row t
    const IT* r = streamed ? probs + (size_t)t * 5 : probs + ((size_t)i_min * T + (size_t)(t - (int64_t)i_min * step)) * 5;
    double x[5];
#pragma unroll
    for (int c = 0; c < 5; c++) x[c] = (double)r[c];
    if (i_max > i_min) {
        double norm = (((fabs(x[0]) + fabs(x[1])) + fabs(x[2])) + fabs(x[3])) + fabs(x[4]);
        if (norm == 0.0) norm = 1.0;
#pragma unroll
        for (int c = 0; c < 5; c++) x[c] = x[c] / norm;
    }
#pragma unroll
    for (int c = 0; c < 5; c++) out[t * 5 + c] = x[c];
}

// The same gather for a BATCH of reads in one launch (the pipelined global path): blockIdx.y = read, one record per read.
template <typename IT>
__global__ __launch_bounds__(256) void assemble_batch_kernel(const IT* __restrict__ probs, const AsmRead* __restrict__ reads, int T, int step,
                                                              double* __restrict__ out, int streamed)
{
    const AsmRead rd = reads[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rd.N) return;
    const int nW = rd.nW, pad = rd.pad;
    int64_t lo = t - T + 1;
    int i_min = lo <= 0 ? 0 : (int)((lo + step - 1) / step);
    int i_max = (int)(t / step);
    if (i_max > nW - 1) i_max = nW - 1;
    if (i_max == nW - 1 && t >= (int64_t)(nW - 1) * step + (T - pad)) i_max--;
    if (i_min > i_max) i_min = i_max;
    const IT* base = probs + (size_t)rd.src_row * 5;
    const IT* r = streamed ? base + (size_t)t * 5 : base + ((size_t)i_min * T + (size_t)(t - (int64_t)i_min * step)) * 5;
    double x[5];
#pragma unroll
    for (int c = 0; c < 5; c++) x[c] = (double)r[c];
    if (i_max > i_min) {
        double norm = (((fabs(x[0]) + fabs(x[1])) + fabs(x[2])) + fabs(x[3])) + fabs(x[4]);
        if (norm == 0.0) norm = 1.0;
#pragma unroll
        for (int c = 0; c < 5; c++) x[c] = x[c] / norm;
    }
    double* o = out + (size_t)(rd.out_row + t) * 5;
#pragma unroll
    for (int c = 0; c < 5; c++) o[c] = x[c];
}

}  // namespace

int rd_assemble_batch_dev(hipStream_t st, const void* d_probs, const AsmRead* d_reads, int n_reads, int64_t max_n, int T, int step,
                          double* d_out, int streamed, int in_f16)
{
    if (n_reads <= 0 || max_n <= 0) return RD_OK;
    const int threads = 256;
    // grid.y is capped at 65535 by HIP: a batch of more reads (many short multi-window reads under a large
    // --gpu-batch-windows) goes out in slices of the record array
    for (int r0 = 0; r0 < n_reads; r0 += 65535) {
        const int nr = n_reads - r0 < 65535 ? n_reads - r0 : 65535;
        const dim3 grid((unsigned)((max_n + threads - 1) / threads), (unsigned)nr);
        if (in_f16)
            hipLaunchKernelGGL(assemble_batch_kernel<_Float16>, grid, dim3(threads), 0, st, (const _Float16*)d_probs, d_reads + r0, T, step, d_out,
                               streamed);
        else
            hipLaunchKernelGGL(assemble_batch_kernel<float>, grid, dim3(threads), 0, st, (const float*)d_probs, d_reads + r0, T, step, d_out, streamed);
        RD_HIP(hipGetLastError());
    }
    return RD_OK;
}

// windows of a read out of the streamed evaluation's rows: window w, time step t < valid[w] comes from row off1[w] + t (its head) for
// t < split[w] and from row off2[w] + t (the read's stream) beyond; rows t >= valid[w] are zero
__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ rows, const int64_t* __restrict__ off1, const int64_t* __restrict__ off2,
                                                             const int32_t* __restrict__ split, const int32_t* __restrict__ valid, int T, float* __restrict__ out)
{
    const int w = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float* o = out + ((size_t)w * T + t) * 5;
    if (t >= valid[w]) {
#pragma unroll
        for (int c = 0; c < 5; c++) o[c] = 0.f;
        return;
    }
    const float* r = rows + ((t < split[w] ? off1[w] : off2[w]) + t) * 5;
#pragma unroll
    for (int c = 0; c < 5; c++) o[c] = r[c];
}

int rd_gather_windows_dev(hipStream_t st, const float* d_rows, const int64_t* d_off1, const int64_t* d_off2, const int32_t* d_split,
                          const int32_t* d_valid, int n_windows, int T, float* d_out)
{
    for (int w0 = 0; w0 < n_windows; w0 += 65535) {
        const int nw = n_windows - w0 < 65535 ? n_windows - w0 : 65535;
        hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)nw), dim3(256), 0, st, d_rows, d_off1 + w0, d_off2 + w0,
                           d_split + w0, d_valid + w0, T, d_out + (size_t)w0 * T * 5);
        RD_HIP(hipGetLastError());
    }
    return RD_OK;
}

int rd_assemble_dev(rd_ctx* ctx, const void* d_probs, int nW, int T, int pad, int step, double* d_out, int64_t N, int streamed, int in_f16)
{
    if (N <= 0) return RD_OK;
    const int threads = 256;
    const int64_t blocks = (N + threads - 1) / threads;
    if (in_f16)   // f16 logits mode: the rows were rounded to f16 by the head kernel; widening is exact
        hipLaunchKernelGGL(assemble_kernel<_Float16>, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, (const _Float16*)d_probs, nW, T, pad,
                           step, d_out, N, streamed);
    else
        hipLaunchKernelGGL(assemble_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, (const float*)d_probs, nW, T, pad, step,
                           d_out, N, streamed);
    RD_HIP(hipGetLastError());
    return RD_OK;
}
