/* Host check of radian_amd/csrc/glibc_math.h: every routine against the running libm, bit for bit, over random arguments of
 * the ranges the beam search uses plus the edges.  Test infrastructure (tests/test_glibc_math_cpu.py compiles and runs it).
 * usage: glibc_math_check <millions of arguments per routine>;  exit status 0 = all identical. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../radian_amd/csrc/glibc_math.h"
#include "../radian_amd/csrc/glibc_tables.h"

static const uint64_t TE[256] = RD_GLIBC_EXP_TAB;
static const uint64_t TL[256] = RD_GLIBC_LOG_TAB;

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void)
{   /* xorshift128+ */
    uint64_t a = s[0], b = s[1];
    s[0] = b;
    a ^= a << 23;
    s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
}
static double unif(void) { return (double)(rnd() >> 11) * 0x1p-53; }

static long bad_exp, bad_log, bad_log1p, bad_lae;
static void ck_exp(double x)
{
    const double a = gm_exp(x, TE), b = exp(x);
    if (gm_bits(a) != gm_bits(b) && !(a != a && b != b) && bad_exp++ < 5) printf("exp(%a): %a vs libm %a\n", x, a, b);
}
static void ck_log(double x)
{
    const double a = gm_log(x, TL), b = log(x);
    if (gm_bits(a) != gm_bits(b) && !(a != a && b != b) && bad_log++ < 5) printf("log(%a): %a vs libm %a\n", x, a, b);
}
static void ck_log1p(double x)
{
    const double a = gm_log1p(x), b = log1p(x);
    if (gm_bits(a) != gm_bits(b) && !(a != a && b != b) && bad_log1p++ < 5) printf("log1p(%a): %a vs libm %a\n", x, a, b);
    if ((x >= 0.0 && x <= 1.0) || x != x) {   /* the SIMD arrangement for [0, 1] */
        const double u = gm_log1p_unit(x);
        if (gm_bits(u) != gm_bits(b) && !(u != u && b != b) && bad_log1p++ < 5) printf("log1p_unit(%a): %a vs libm %a\n", x, u, b);
    }
}

int main(int argc, char** argv)
{
    const long n = (argc > 1 ? atol(argv[1]) : 5) * 1000000L;
    const double edges[] = {0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 0x1p-54, -0x1p-54, 0x1p-55, 0x1p-29, 0x1p-30, 0.41421356237309503, 0.4142135623730951,
                            -0.2928932188134524, -0.29289321881345254, 0x1p-1022, 0x1p-1074, 0x1.fffffffffffffp-1023, 1e-300, 1e300, 511.9, 512.0,
                            -511.9, -512.0, -708.3, -708.4, -744.9, -745.13, -745.14, -746.0, -1023.9, -1024.0, -1e5, 709.7, 709.8, 1024.0,
                            1.0 - 0x1p-4, 1.0 + 0x1.09p-4, 1.0 - 0x1p-19, 1.0 - 0x1p-20, 1.0 - 0x1p-21, 1.0 - 0x1p-30, 0x1.fffffffffffffp-1, 0x1.0000000000001p+0, 0.9375, 1.0647, INFINITY, -INFINITY, NAN};
    for (size_t i = 0; i < sizeof edges / sizeof *edges; i++) {
        ck_exp(edges[i]);
        ck_exp(nextafter(edges[i], 0.0));
        ck_exp(nextafter(edges[i], -INFINITY));
        ck_log(edges[i]);
        ck_log(nextafter(edges[i], INFINITY));
        ck_log(nextafter(edges[i], 0.0));
        ck_log1p(edges[i]);
        ck_log1p(nextafter(edges[i], INFINITY));
        ck_log1p(nextafter(edges[i], -1.0));
    }
    for (long i = 0; i < n; i++) {
        /* exp: the decoder's arguments are differences of log-probabilities, <= 0 */
        ck_exp(-unif() * 40.0);
        ck_exp(-unif() * 800.0);
        ck_exp(-ldexp(unif(), -(int)(rnd() % 70)));
        ck_exp((unif() - 0.5) * 1500.0);
        /* log: probabilities (float32 and float64 rows), LM mixtures, and anything else */
        ck_log(unif());
        ck_log((double)(float)unif());
        ck_log(ldexp(unif(), -(int)(rnd() % 150)));
        ck_log(0.9 + 0.2 * unif());
        ck_log(gm_dbl(rnd() & 0x7fffffffffffffffull));
        /* log1p of exp(<= 0) in [0, 1], and beyond */
        ck_log1p(unif());
        ck_log1p(ldexp(unif(), -(int)(rnd() % 80)));
        ck_log1p(0.41 + 0.01 * unif());
        ck_log1p(1.0 - ldexp(unif(), -(int)(rnd() % 40)));
        ck_log1p(-unif() * 0.999);
        ck_log1p(ldexp(unif(), (int)(rnd() % 70)));
        /* the composition numpy's logaddexp evaluates: hi + log1p(exp(lo - hi)) */
        {
            const double hi = -unif() * 50.0, lo = hi - unif() * 60.0;
            const double a = hi + gm_log1p_unit(gm_exp(lo - hi, TE)), b = hi + log1p(exp(lo - hi));
            if (gm_bits(a) != gm_bits(b)) bad_lae++;
        }
    }
    printf("arguments per routine: %ld x 4-5;  differing results: exp %ld, log %ld, log1p %ld, logaddexp %ld\n", n, bad_exp, bad_log, bad_log1p,
           bad_lae);
    return (bad_exp || bad_log || bad_log1p || bad_lae) ? 1 : 0;
}
