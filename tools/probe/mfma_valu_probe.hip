// How fast does a VALU / SALU wave run beside waves that stream MFMAs on the same SIMD?  (The beam search beside the conv kernel:
// DESIGN.md 4.11.)  One workgroup of 12 waves per CU: waves 0-7 (two per SIMD) issue v_mfma_f32_32x32x2_f32 back to back on 8
// independent accumulators, waves 8-11 (one per SIMD) run a dependent v_fma_f64 chain with scalar instructions mixed in, and time
// themselves with s_memtime.  Variants: MFMA waves on / off; s_setprio in either role; s_nop / s_sleep between the MFMA groups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// s_nop k idles the wave for k + 1 issue slots of 4 cycles (measured: "s_nop 15" x 2 after every MFMA -> 136 cycles per MFMA)
template <int QUADS>
__device__ __forceinline__ void nops()
{
    if constexpr (QUADS >= 16) { asm volatile("s_nop 15"); nops<QUADS - 16>(); }
    else if constexpr (QUADS > 0) asm volatile("s_nop %0" ::"n"(QUADS - 1));
}

// (a workgroup barrier cannot be used: the MFMA waves spin until the VALU waves say they are done)
template <int VARIANT, int PACE = 0>
__global__ __launch_bounds__(768) void probe2(double* out, unsigned long long* cyc, unsigned long long* mf, int mfma_on, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < 8) {
        if (!mfma_on) return;
        if ((VARIANT == 6 || VARIANT == 8) && wave >= 4) return;      // one MFMA wave per SIMD
        if (VARIANT == 2) __builtin_amdgcn_s_setprio(0);
        f32x16 acc[8];
        for (int i = 0; i < 8; i++)
            for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
        float a = (float)lane, b = 1.0f;
        unsigned long long n = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    // paced: the wave keeps off the vector issue port for 4 PACE cycles after each MFMA (variants 8, 9, 10)
                    if constexpr (PACE > 0) {   // (the scheduler moves MFMAs across asm statements unless fenced)
                        __builtin_amdgcn_sched_barrier(0);
                        nops<PACE>();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (VARIANT == 3) asm volatile("s_nop 15");
                if (VARIANT == 4) __builtin_amdgcn_s_sleep(1);
            }
            n += 32;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; i++) s += acc[i][0];
        if (lane == 0) { mf[blockIdx.x * 8 + wave] = n; cyc[4096 + blockIdx.x * 8 + wave] = t1 - t0; }
        if (s == 12345.f) out[0] = s;
        return;
    }
    if (VARIANT == 1 || VARIANT == 2 || VARIANT == 9 || VARIANT == 8) __builtin_amdgcn_s_setprio(3);
    double x = 1.0 + lane * 1e-9, c = 0.999999;
    float xf = 1.0f + lane * 1e-3f, cf = 0.999f;
    unsigned xi = lane * 2654435761u;
    int sacc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            // which vector instructions wait for the matrix pipe?  (variants 11..16: the victim's instruction kind)
            if (VARIANT == 7) sacc ^= __builtin_amdgcn_readfirstlane(sacc + k);   // (scalar only)
            else if (VARIANT == 11) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(xf) : "v"(cf));
            else if (VARIANT == 12) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(xi) : "v"(k));
            else if (VARIANT == 13) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            else if (VARIANT == 14) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            else if (VARIANT == 15) asm volatile("v_exp_f32 %0, %0" : "+v"(xf));
            else if (VARIANT == 16) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(xf));
            else x = __builtin_fma(x, c, 1e-12);
            sacc += __builtin_amdgcn_readfirstlane(it + k);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        cyc[blockIdx.x * 4 + (wave - 8)] = t1 - t0;
        __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    out[1 + (size_t)blockIdx.x * 256 + (threadIdx.x - 512)] = x + sacc + xf + xi;
}

template <int V, int PACE = 0>
void run(const char* name, int mfma_on, double* d_out, unsigned long long* d_cyc, unsigned long long* d_mf)
{
    const int nwg = 256, iters = 2000;
    hipMemset(d_cyc, 0, 8192 * 8);
    hipMemset(d_mf, 0, 4096 * 8);
    hipLaunchKernelGGL((probe2<V, PACE>), dim3(nwg), dim3(768), 0, 0, d_out, d_cyc, d_mf, mfma_on, iters);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> c(8192), m(4096);
    hipMemcpy(c.data(), d_cyc, 8192 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), d_mf, 4096 * 8, hipMemcpyDeviceToHost);
    double vs = 0, ms = 0, mc = 0;
    for (int i = 0; i < nwg * 4; i++) vs += (double)c[i];
    for (int i = 0; i < nwg * 8; i++) { ms += (double)m[i]; mc += (double)c[4096 + i]; }
    const double per_instr = vs / (nwg * 4) / (iters * 32.0);   // 16 fma + 16 readfirstlane per iteration
    printf("%-46s %s: VALU-wave %.1f cycles per instruction", name, e == hipSuccess ? "ok" : hipGetErrorString(e), per_instr);
    if (mfma_on) printf("; MFMA waves: %.1f cycles per MFMA per wave", mc / ms);
    printf("\n");
}

int main()
{
    double* d_out; unsigned long long *d_cyc, *d_mf;
    hipMalloc(&d_out, (1 + 256 * 256) * 8);
    hipMalloc(&d_cyc, 8192 * 8);
    hipMalloc(&d_mf, 4096 * 8);
    run<0>("alone (MFMA waves leave at once)", 0, d_out, d_cyc, d_mf);
    run<0>("beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<1>("... VALU wave at s_setprio 3", 1, d_out, d_cyc, d_mf);
    run<2>("... VALU wave prio 3, MFMA waves prio 0", 1, d_out, d_cyc, d_mf);
    run<3>("... MFMA waves: s_nop 15 after every 8 MFMAs", 1, d_out, d_cyc, d_mf);
    run<4>("... MFMA waves: s_sleep 1 after every 8 MFMAs", 1, d_out, d_cyc, d_mf);
    run<6>("beside ONE MFMA wave per SIMD", 1, d_out, d_cyc, d_mf);
    // paced MFMA waves: idle for 4 PACE cycles after EVERY MFMA (the next MFMA then reaches the port about when the pipe is free)
    run<8, 8>("ONE paced MFMA wave (32 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<8, 12>("ONE paced MFMA wave (48 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<8, 14>("ONE paced MFMA wave (56 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<8, 15>("ONE paced MFMA wave (60 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<9, 16>("TWO paced MFMA waves (64 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<9, 24>("TWO paced MFMA waves (96 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<9, 28>("TWO paced MFMA waves (112 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<9, 30>("TWO paced MFMA waves (120 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<9, 31>("TWO paced MFMA waves (124 cyc), VALU prio 3", 1, d_out, d_cyc, d_mf);
    run<10, 16>("TWO paced MFMA waves (64 cyc), VALU prio 0", 1, d_out, d_cyc, d_mf);
    run<10, 28>("TWO paced MFMA waves (112 cyc), VALU prio 0", 1, d_out, d_cyc, d_mf);
    run<10, 30>("TWO paced MFMA waves (120 cyc), VALU prio 0", 1, d_out, d_cyc, d_mf);
    run<11>("v_fma_f32 chain beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<11>("v_fma_f32 chain alone", 0, d_out, d_cyc, d_mf);
    run<12>("v_xad_u32 chain beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<12>("v_xad_u32 chain alone", 0, d_out, d_cyc, d_mf);
    run<13>("v_add_f64 chain beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<13>("v_add_f64 chain alone", 0, d_out, d_cyc, d_mf);
    run<14>("v_mul_f64 chain beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<14>("v_mul_f64 chain alone", 0, d_out, d_cyc, d_mf);
    run<15>("v_exp_f32 chain beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<15>("v_exp_f32 chain alone", 0, d_out, d_cyc, d_mf);
    run<16>("v_cvt_f64_f32 beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<16>("v_cvt_f64_f32 alone", 0, d_out, d_cyc, d_mf);
    run<7>("scalar-only wave beside 2 MFMA waves per SIMD", 1, d_out, d_cyc, d_mf);
    run<7>("scalar-only wave alone", 0, d_out, d_cyc, d_mf);
    return 0;
}
