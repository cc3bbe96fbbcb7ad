// decode_common.h -- what the beam-search kernels share: the score arithmetic (logaddexp on range-specific exp / log1p, or on glibc 2.35's
// operation sequence), log(0) = -inf, and the launch arguments.  Included by decode.hip (the wave-per-sequence kernels, widths <= 51)
// and decode_wide.hip (any wider beam).  Everything is internal linkage: each translation unit gets its own copy of the tables.
#pragma once
#include "common.h"
#include "glibc_math.h"
#include "glibc_tables.h"

#include <math.h>

namespace {


// the "glibc" arithmetic mode's tables (glibc_math.h): exp's is copied into LDS per sequence, log's is read in place
__device__ const uint64_t g_gm_exp_tab[256] = RD_GLIBC_EXP_TAB;
__device__ const uint64_t g_gm_log_tab[256] = RD_GLIBC_LOG_TAB;

constexpr unsigned kHashB = 0x9E3779B1u;   // odd multiplier of the context hash (long contexts)
constexpr double kLogE2 = 0.693147180559945309417232121458176568;

// ---- logaddexp --------------------------------------------------------------------------------------------------------
// numpy's npy_logaddexp (decode.py:172-201 call np.logaddexp on Python floats) is  hi + log1p(exp(lo - hi)).  The two
// transcendentals are evaluated by the routines below instead of the generic device-library ones: the argument ranges are
// known (exp of a non-positive number, log1p of a number in [0, 1]), which makes both a short fma chain -- 75 instructions
// for the pair against ~175 (the generic log1p runs a double-double reduction) -- at the same accuracy class: measured
// against 80-bit references over 2.5e7 arguments, exp_nonpos <= 0.68 ulp, log1p_unit <= 0.63 ulp, their composition
// <= 1.55 ulp (glibc's: 0.51 / 0.82 / 1.54).  So scores keep agreeing with the reference's to a few ulp.
#pragma clang fp contract(off)
// Horner chains with the coefficients in scalar register pairs.  As plain C++ hipcc keeps all ~30 coefficients in VGPRs
// across the time loop and spends a v_mov_b64 + v_fmac_f64 per step (the VOP2 form accumulates into the coefficient's
// copy); here a step is one v_fma_f64 and no vector register.  One asm statement per chain (no hazard padding between
// the steps: each v_fma_f64 only reads the previous one's result, which the hardware interlocks).
__device__ __forceinline__ double horner10(double q, double x, double c0, double c1, double c2, double c3, double c4, double c5,
                                           double c6, double c7, double c8, double c9)
{
    asm("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %5\n\t"
        "v_fma_f64 %0, %0, %1, %6\n\tv_fma_f64 %0, %0, %1, %7\n\tv_fma_f64 %0, %0, %1, %8\n\tv_fma_f64 %0, %0, %1, %9\n\t"
        "v_fma_f64 %0, %0, %1, %10\n\tv_fma_f64 %0, %0, %1, %11"
        : "+v"(q)
        : "v"(x), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7), "s"(c8), "s"(c9));
    return q;
}
__device__ __forceinline__ double horner16(double q, double x, double c0, double c1, double c2, double c3, double c4, double c5,
                                           double c6, double c7, double c8, double c9, double c10, double c11, double c12,
                                           double c13, double c14, double c15)
{
    asm("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %5\n\t"
        "v_fma_f64 %0, %0, %1, %6\n\tv_fma_f64 %0, %0, %1, %7\n\tv_fma_f64 %0, %0, %1, %8\n\tv_fma_f64 %0, %0, %1, %9\n\t"
        "v_fma_f64 %0, %0, %1, %10\n\tv_fma_f64 %0, %0, %1, %11\n\tv_fma_f64 %0, %0, %1, %12\n\tv_fma_f64 %0, %0, %1, %13\n\t"
        "v_fma_f64 %0, %0, %1, %14\n\tv_fma_f64 %0, %0, %1, %15\n\tv_fma_f64 %0, %0, %1, %16\n\tv_fma_f64 %0, %0, %1, %17"
        : "+v"(q)
        : "v"(x), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7), "s"(c8), "s"(c9), "s"(c10), "s"(c11),
          "s"(c12), "s"(c13), "s"(c14), "s"(c15));
    return q;
}
// e^d for d <= 0 (d = -inf and d < -745.2 give 0; subnormal results are rounded by v_ldexp_f64)
__device__ __forceinline__ double exp_nonpos(double d)
{
    constexpr double L2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double dd = fmax(d, -746.0);
    const double kf = rint(dd * L2E);
    const double rh = __builtin_fma(-kf, LN2_HI, dd);   // exact: LN2_HI has 32 significant bits, |kf| < 2^11
    const double tl = kf * LN2_LO;
    const double r = rh - tl;
    const double rl = (rh - r) - tl;                    // r + rl = rh - tl to ~2^-106
    double q = 1.0 / 6227020800.0;                      // Taylor: e^r = 1 + r + r^2 (1/2! + r/3! + ... + r^11/13!), |r| <= ln2/2
    q = horner10(q, r, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
                 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    const double s1 = 1.0 + r;
    const double e1 = (1.0 - s1) + r;                   // exact (|r| < 1)
    const double p = s1 + (__builtin_fma(r * r, q, rl) + e1);
    return ldexp(p, (int)kf);
}
// log(1 + t) for 0 <= t <= 1:  2 atanh(t / (2 + t)) = 2 (s + s^3/3 + s^5/5 + ...), s <= 1/3.  The quotient s comes from
// a Newton-refined reciprocal (2 <= 2 + t <= 3: no scaling cases) and its rounding is compensated to first order, which
// is worth more than a correctly rounded division would be.
__device__ __forceinline__ double log1p_unit(double t)
{
    const double d = 2.0 + t;
    const double dlo = t - (d - 2.0);                   // exact
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);                         // 1 / d to working precision
    const double s = t * y;
    double r = __builtin_fma(-s, d, t);
    r = __builtin_fma(-s, dlo, r);                      // t - s (2 + t)
    const double slo = r * y;
    const double z = s * s;
    double P = 1.0 / 35.0;
    P = horner16(P, z, 1.0 / 33.0, 1.0 / 31.0, 1.0 / 29.0, 1.0 / 27.0, 1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0,
                 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0);
    const double corr = __builtin_fma(s * z, P, slo);
    const double v = 2.0 * (s + corr);
    return t < 0x1p-60 ? t : v;
}
#pragma clang fp contract(on)

// rank += number of k0..k3 that are > key (NaN compares false).  Four compares into four scalar mask pairs, then four
// add-with-carry: 8 vector instructions per 4 keys, and each mask is read >= 3 instructions after the compare that wrote
// it (a VALU-written SGPR needs 2 wait states before a VALU reads it as carry-in; as C++ the compiler chains everything
// through VCC with an s_nop behind every compare and a v_cndmask per pair).
__device__ __forceinline__ int count4_gt(int rank, double k0, double k1, double k2, double k3, double key)
{
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_gt_f64_e64 %1, %5, %9\n\t"
        "v_cmp_gt_f64_e64 %2, %6, %9\n\t"
        "v_cmp_gt_f64_e64 %3, %7, %9\n\t"
        "v_cmp_gt_f64_e64 %4, %8, %9\n\t"
        "v_addc_co_u32_e64 %0, %1, %0, 0, %1\n\t"
        "v_addc_co_u32_e64 %0, %2, %0, 0, %2\n\t"
        "v_addc_co_u32_e64 %0, %3, %0, 0, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, 0, %4"
        : "+v"(rank), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(key));
    return rank;
}

// numpy npy_logaddexp
__device__ __forceinline__ double lae(double x, double y)
{
    const double hi = fmax(x, y), lo = fmin(x, y);
    const double r = hi + log1p_unit(exp_nonpos(lo - hi));
    return x == y ? x + kLogE2 : r;    // (also the -inf / -inf case: lo - hi would be NaN)
}

// numpy npy_logaddexp on glibc 2.35's exp / log1p, operation for operation (glibc_math.h): bit-identical to the reference's
// scores on an x86-64 FMA host.  T = the exp table in LDS.
__device__ __forceinline__ double lae_gx(double x, double y, const uint64_t* T)
{
    const double hi = fmax(x, y), lo = fmin(x, y);
    const double r = hi + gm_log1p_unit(gm_exp(lo - hi, T));      // (lo - hi = -|x - y| exactly, as npy_logaddexp's tmp / -tmp)
    return x == y ? x + kLogE2 : r;
}

// The workgroup IS one wave (launch bounds 64), so LDS hand-offs between lanes need no s_barrier and -- the point -- no
// s_waitcnt vmcnt(0): __syncthreads() would also wait for the acknowledgement of the trie stores to HBM of every time
// step (~1-2 us each on a loaded chip).  LDS operations of one wave execute in order; the fences keep the compiler from
// moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// decode.py:16-17
template <bool GX>
__device__ __forceinline__ double log_m(double x)
{
    if constexpr (GX) return gm_log(x, g_gm_log_tab);
    else return log(x);
}
template <bool GX>
// decode.py:16-17: log(0) = -inf.  A value that is no probability -- negative, NaN: the output of a model with non-finite weights, or a
// caller's own matrix; +inf -- is treated as probability 0 too: math.log raises on the first and sorts NaN scores arbitrarily on the second; here
// a NaN score would leave beam slots unclaimed in the ranking and the trie ids behind them undefined.  (Valid rows: the same bits.)
__device__ __forceinline__ double safe_log(double x) { return (x > 0.0 && x <= 1.79769313486231570815e+308) ? log_m<GX>(x) : -INFINITY; }   // (+inf too: inf - inf later would be a NaN score)

struct DecodeArgs {
    const void* probs;
    const int64_t* seq_off;
    const int64_t* seq_off2;   // nullable: rows t >= seq_split[i] come from row seq_off2[i] + t (streamed forward)
    const int32_t* seq_split;
    const int32_t* seq_len;
    const int64_t* node_off;
    const int64_t* label_off;
    int W;
    int glibc_math;       // rd_set_decode_math: 1 = log / exp / log1p as glibc 2.35 evaluates them (glibc_math.h)
    // LM
    const double* lm_table;
    const uint32_t* lm_gate;
    // sparse models (nullable): bit ctx = the model holds no entry for this context.  The reference looks model[context] up for
    // every kept labeling of at least k labels at every time step, gate or no gate (decode.py:83,161,182), and raises KeyError on
    // an absent one; a labeling's context changes only when it is created, so each labeling that ENTERS the beam before the last
    // time step is checked once, and a hit is reported as label_len = -1 for the sequence.
    const uint32_t* lm_missing;
    int k;                // context length: labels per LM context (decode.py:42-49)
    // long contexts (k > 13, a mode of this library with no reference behaviour -- the reference needs a dict entry per
    // context, decode.py:83): the table row of a context is H(context) & tmask with the polynomial hash
    // H(l_0 .. l_{k-1}) = sum l_i B^(k-1-i) mod 2^32, kept incrementally per beam (window of the last k labels; the label
    // that leaves the window comes from a 256-label ring per beam)
    int hashed;
    unsigned tmask;       // rows of the table - 1
    unsigned bk;          // B^k mod 2^32
    double s_thr;
    // trie in HBM
    int4* childtab;
    int* backptr;
    // out
    uint8_t* labels;
    int32_t* label_len;
    double* best_score;
};

}  // namespace
