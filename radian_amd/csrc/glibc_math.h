// glibc_math.h -- double-precision exp, log, log1p evaluated with EXACTLY the operation sequence of glibc 2.35's x86-64
// FMA build (the libm of this image; the reference's np.logaddexp / math.log run on it), so that results are
// bit-identical to that library's.  Used by the beam search's "glibc" arithmetic mode (rd_set_decode_math) -- with it the
// decoder reproduces the reference's scores bit for bit, and with them its ordering of labelings that are equiprobable
// in exact arithmetic -- and, compiled for the host, by tests/test_glibc_math_cpu.py, which compares
// every routine with the running libm over tens of millions of arguments.
//
// Sources restated (published algorithms; operation order, and which a*b+c are fused, read from `objdump -d libm.so.6`
// of the FMA + AVX2 ifunc variants, see tools/gen_glibc_tables.py):
//   exp   : sysdeps/ieee754/dbl-64/e_exp.c   (ARM optimized-routines exp, N = 128 table, degree-5 polynomial)
//   log   : sysdeps/ieee754/dbl-64/e_log.c   (ARM optimized-routines log, N = 128 table; separate polynomial near 1)
//   log1p : sysdeps/ieee754/dbl-64/s_log1p.c (fdlibm; not an ifunc: plain SSE2 arithmetic, no fused operation)
// Every floating-point operation below is a single IEEE-754 binary64 operation: contraction is off for this header and
// fused multiply-adds are written as gm_fma.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GM_FN __host__ __device__ inline
#else
#define GM_FN static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// K(c): a polynomial / reduction constant materialised at its point of use (two s_mov_b32 into a scalar register pair) instead
// of being hoisted to kernel scope.  Used in gm_log only: the beam search calls log once per probability in a per-tile
// prepass (1/64 of the work), yet hoisted, its 20 constants hold 40 SGPRs through the whole time loop and the compiler spills
// the loop's own scalars around them.  exp / log1p run twice per time step: their constants STAY hoisted -- materialised in
// place they cost two scalar instructions each per use, +0.1 to +0.3 us on a 1.6-2.0 us step (measured).
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ __forceinline__ double gm_kbits()
{
    uint32_t lo, hi;   // (the asm statements have no inputs: nothing for the compiler to hoist; volatile: not hoisted themselves)
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((uint32_t)BITS));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((uint32_t)(BITS >> 32)));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
#define gm_k(x) gm_kbits<__builtin_bit_cast(uint64_t, (double)(x))>()
#define K(x) gm_k(x)
#define GM_CONST constexpr
#else
#define gm_k(x) (x)
#define K(x) (x)
#define GM_CONST const
#endif

GM_FN double gm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
GM_FN uint64_t gm_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
GM_FN double gm_dbl(uint64_t u)
{
    double x;
    memcpy(&x, &u, 8);
    return x;
}

// ---------------------------------------------------------------------------------------------------------------- exp
// T = RD_GLIBC_EXP_TAB (256 words).  e_exp.c: x = k ln2/128 + r; 2^(k/128) = scale (1 + tail); exp(r) - 1 by a polynomial.
GM_FN double gm_exp_special(double tmp, uint64_t sbits, uint64_t ki)
{
    if ((ki & 0x80000000u) == 0) {   // k > 0: the exponent of scale might have overflowed by <= 460
        sbits -= 1009ull << 52;
        const double scale = gm_dbl(sbits);
        return 0x1p1009 * gm_fma(scale, tmp, scale);
    }
    // k < 0: special care in the subnormal range
    sbits += 1022ull << 52;
    const double scale = gm_dbl(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
        // round y to the right precision before scaling it into the subnormal range (avoids double rounding)
        const double hi = 1.0 + y;
        double lo = (scale - y) + st;
        lo = ((1.0 - hi) + y) + lo;
        y = (lo + hi) - 1.0;
        if (y == 0.0) y = 0.0;   // (-0.0 -> +0.0 under downward rounding; no effect in round-to-nearest)
    }
    return 0x1p-1022 * y;
}

GM_FN double gm_exp(double x, const uint64_t* T)
{
    GM_CONST double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3,
                 C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const uint64_t ix = gm_bits(x);
    const uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
    // the main path, evaluated for every argument (one straight instruction stream on a SIMD machine; the range cases below
    // replace its result, and the 512 <= |x| < 1024 case continues from its intermediate values exactly as e_exp.c does)
    double kd = gm_fma(InvLn2N, x, Shift);             // z + Shift, fused
    const uint64_t ki = gm_bits(kd);
    kd = kd - Shift;
    double r = gm_fma(kd, NegLn2hiN, x);
    r = gm_fma(kd, NegLn2loN, r);
    const uint64_t idx = 2 * (ki & 127u);
    const uint64_t top = ki << 45;
    const double tail = gm_dbl(T[idx]);
    const uint64_t sbits = T[idx + 1] + top;
    const double r2 = r * r;
    const double p23 = gm_fma(r, C3, C2);
    const double tr = tail + r;
    const double p45 = gm_fma(r, C5, C4);
    double tmp = gm_fma(p23, r2, tr);
    const double r4 = r2 * r2;
    tmp = gm_fma(r4, p45, tmp);
    const double scale = gm_dbl(sbits);
    double res = gm_fma(scale, tmp, scale);
    if (abstop - 0x3c9u >= 0x3fu) {                    // |x| < 2^-54 or |x| >= 512 or NaN
        if (abstop - 0x3c9u >= 0x80000000u) res = 1.0 + x;   // tiny: exp(x) = 1 + x to within rounding
        else if (abstop >= 0x409u) {                   // |x| >= 1024
            if (ix == 0xfff0000000000000ull) res = 0.0;
            else if (abstop >= 0x7ffu) res = 1.0 + x;  // NaN, +inf
            else res = (ix >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;   // underflow -> 0, overflow -> inf
        } else res = gm_exp_special(tmp, sbits, ki);   // 512 <= |x| < 1024: the result may be subnormal / huge
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------------------- log
// TL = RD_GLIBC_LOG_TAB viewed as 128 x {invc, logc}.  e_log.c: x = 2^k z, z in [OFF, 2 OFF); r = z invc - 1;
// log x = k ln2 + logc + log1p(r).
GM_FN double gm_log(double x, const uint64_t* TL)
{
    GM_CONST double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    GM_CONST double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3, A3 = 0x1.999b324f10111p-3,
                 A4 = -0x1.55575e506c89fp-3;
    GM_CONST double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
                 B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4,
                 B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    uint64_t ix = gm_bits(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {      // 1 - 2^-4 <= x < 1 + 0x1.09p-4: log1p(x - 1) directly
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double q1 = gm_fma(r2, K(B3), gm_fma(K(B2), r, K(B1)));     // B1 + r B2 + r2 B3
        const double q4 = gm_fma(r2, K(B6), gm_fma(K(B5), r, K(B4)));     // B4 + r B5 + r2 B6
        double q7 = gm_fma(r2, K(B9), gm_fma(K(B8), r, K(B7)));           // B7 + r B8 + r2 B9
        q7 = gm_fma(r3, K(B10), q7);
        double inner = gm_fma(q7, r3, q4);
        inner = gm_fma(inner, r3, q1);
        // r - r^2/2 in double-double: rhi = r rounded to 26 bits
        const double rw = gm_fma(r, K(0x1p27), r);               // r + w, w = r 2^27 (fused)
        const double rhi = gm_fma(K(-0x1p27), r, rw);            // (r + w) - w     (fused)
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = gm_fma(rhi2, B0, r);                   // r + w', w' = rhi^2 B0 (fused)
        double lo = gm_fma(rhi2, B0, r - hi);                    // (r - hi) + w'   (fused)
        lo = gm_fma(B0 * rlo, rhi + r, lo);
        const double y = gm_fma(inner, r3, lo);                  // r3 inner + lo   (fused)
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {                   // subnormal, zero, inf, NaN or negative
        if (ix * 2 == 0) return -1.0 / 0.0;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / (x - x);
        ix = gm_bits(x * 0x1p52);                               // subnormal: normalise
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = gm_dbl(TL[2 * i]), logc = gm_dbl(TL[2 * i + 1]);
    const double z = gm_dbl(iz);
    const double r = gm_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = gm_fma(kd, K(Ln2hi), logc);
    const double hi = w + r;
    double lo = (w - hi) + r;
    lo = gm_fma(kd, K(Ln2lo), lo);
    const double r2 = r * r;
    const double p12 = gm_fma(K(A2), r, K(A1));
    const double rr2 = r * r2;
    const double p34 = gm_fma(r, K(A4), K(A3));
    lo = gm_fma(r2, K(A0), lo);
    const double p = gm_fma(p34, r2, p12);
    const double y = gm_fma(rr2, p, lo);
    return y + hi;
}

// -------------------------------------------------------------------------------------------------------------- log1p
GM_FN double gm_log1p(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    const int32_t hx = (int32_t)(gm_bits(x) >> 32);
    const int32_t ax = hx & 0x7fffffff;
    int k = 1;
    int32_t hu = 0;
    double f = 0.0, c = 0.0;
    if (hx < 0x3FDA827A) {                                      // x < 0.41422
        if (ax >= 0x3ff00000) {                                 // x <= -1
            if (x == -1.0) return -0x1p54 / 0.0;
            return (x - x) / (x - x);
        }
        if (ax < 0x3e200000) {                                  // |x| < 2^-29
            if (ax < 0x3c900000) return x;                      // |x| < 2^-54
            return x - (x * x) * 0.5;
        }
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) {              // -0.2929 < x < 0.41422
            k = 0;
            f = x;
            hu = 1;
        }
    } else if (hx >= 0x7ff00000) {
        return x + x;
    }
    if (k != 0) {
        double u;
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = (int32_t)(gm_bits(u) >> 32);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);        // correction term
            c = c / u;
        } else {
            u = x;
            hu = (int32_t)(gm_bits(u) >> 32);
            k = (hu >> 20) - 1023;
            c = 0.0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = gm_dbl((gm_bits(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));   // normalise u
        } else {
            k += 1;
            u = gm_dbl((gm_bits(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));   // normalise u / 2
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    const double hfsq = (0.5 * f) * f;
    const double kd = (double)k;
    if (hu == 0) {                                              // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c = kd * ln2_lo + c;
            return c + kd * ln2_hi;
        }
        const double R = (1.0 - 0.66666666666666666 * f) * hfsq;
        if (k == 0) return f - R;
        return kd * ln2_hi - ((R - (kd * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * Lp1;
    const double z2 = z * z;
    const double R2 = Lp2 + z * Lp3;
    const double z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5;
    const double z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
    const double t = s * (hfsq + R);
    if (k == 0) return f - (hfsq - t);
    return kd * ln2_hi - ((hfsq - (t + (kd * ln2_lo + c))) - f);
}

// log1p on [0, 1] (and NaN): the same operations as gm_log1p for such arguments, arranged for a SIMD machine -- the
// 1 + x >= sqrt(2) reduction is the only divergent block, the rest is one instruction stream with selects.  (The decoder's
// argument is exp of a non-positive number.)
GM_FN double gm_log1p_unit(double t)
{
    GM_CONST double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    GM_CONST double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    const uint32_t hx = (uint32_t)(gm_bits(t) >> 32);
    int k = 0;
    int32_t hu = 1;
    double f = t, cn = 0.0, cd = 1.0;
    if (hx >= 0x3FDA827Au && hx < 0x7ff00000u) {              // 0.41422 <= t (<= 1): u = 1 + t in [sqrt 2, 2]
        double u = 1.0 + t;
        hu = (int32_t)(gm_bits(u) >> 32);
        k = (hu >> 20) - 1023;                                  // 0, or 1 for u = 2
        cn = (k > 0) ? 1.0 - (u - t) : t - (u - 1.0);          // correction term c = cn / u, divided below
        cd = u;
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = gm_dbl((gm_bits(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));
        } else {
            k += 1;
            u = gm_dbl((gm_bits(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    // (outside the block, so that the two divisions of the routine are independent chains of one instruction stream; lanes
    // that skipped the block divide 0 by 1)
    const double c = cn / cd;
    const double hfsq = (0.5 * f) * f;
    const double kd = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * Lp1;
    const double z2 = z * z;
    const double R2 = Lp2 + z * Lp3;
    const double z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5;
    const double z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
    const double tt = s * (hfsq + R);
    const double kl = kd * ln2_lo + c, kh = kd * ln2_hi;
    double res = (k == 0) ? f - (hfsq - tt) : kh - ((hfsq - (tt + kl)) - f);
    if (hu == 0) {                                              // |f| < 2^-20 (t within 2^-19 of 1)
        if (f == 0.0) res = (k == 0) ? 0.0 : kl + kh;
        else {
            const double Rs = (1.0 - 0.66666666666666666 * f) * hfsq;
            res = (k == 0) ? f - Rs : kh - ((Rs - kl) - f);
        }
    }
    if (hx < 0x3e200000u) res = (hx < 0x3c900000u) ? t : t - (t * t) * 0.5;   // t < 2^-29 (< 2^-54: t)
    return res;
}

#undef K
#undef GM_CONST

#if defined(__clang__)
#pragma clang fp contract(on)
#endif
