// glibc_trig.h -- bit-exact restatement of glibc 2.35's sin(), cos() and acos() (sysdeps/ieee754/dbl-64 s_sin.c / e_asin.c, the
// x86_64 FMA variants the dynamic loader selects on every AVX2 host) for host and device, on the domain the simulator needs.
//
// Why: the inter-node sprouting (greenhouse.py:259-306) turns Murray's angle phi_2 = degrees(arccos(c)), c in (0, 1], into the
// rotation cos / sin(radians(phi_2)) that places the new node; the reference evaluates the three with libm (numpy hands float64
// scalars and small arrays to it), the oracle likewise. glibc's results are NOT always the correctly rounded ones (about 0.1 % of
// the inputs differ from them), so neither ROCm's functions nor a correctly rounded evaluation reproduce every last bit -- and a
// last-bit difference in a node position occasionally crosses a rounding boundary of the printed CSV (1 sample in 4352,
// profiles/r02_validate_final.log). Hence the same route as gpow.h: the published algorithms with the same operation order and
// the same fused multiply-adds the FMA build executes (read off the image's libm.so.6), tables generated from that libm
// (glibc_trig_tables.h, tools/gen_glibc_trig_tables.py). tests/test_sim_core.py compares all three with math.sin / cos / acos on
// millions of inputs.
// Domain: sin / cos 0 <= x < 2.426 (|x| beyond the table range returns NaN so a misuse is loud); acos 0 < c <= 1.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef OCTA_HD
#if defined(__HIP__) || defined(__HIPCC__)
#define OCTA_HD __host__ __device__
#else
#define OCTA_HD
#endif
#endif
#ifndef OCTA_CONST
#define OCTA_CONST static constexpr
#endif
#include "glibc_trig_tables.h"

namespace octa_gtrig {

OCTA_HD inline uint64_t asu64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
OCTA_HD inline double fma_(double a, double b, double c) { return ::fma(a, b, c); }

OCTA_CONST double BIG = 0x1.8p+45, HP0 = 0x1.921fb54442d18p+0, HP1 = 0x1.1a62633145c07p-54;
OCTA_CONST double SN3 = -0x1.5555555555515p-3, SN5 = 0x1.11110e829872fp-7;
OCTA_CONST double CS2 = 0x1.0p-1, CS4 = -0x1.5555555555535p-5, CS6 = 0x1.6c16bedd9e239p-10;
OCTA_CONST double S1 = -0x1.5555555555555p-3, S2 = 0x1.1111111110ecep-7, S3 = -0x1.a01a019db08b8p-13, S4 = 0x1.71de27b9a7ed9p-19,
                  S5 = -0x1.addffc2fcdf59p-26;

// s_sin.c do_sin: sin(x + dx) for |x| < 0.855 (x, dx the head and tail of the argument)
OCTA_HD inline double do_sin(double x, double dx) {
    const double xold = x;
    if (fabs(x) < 0.126) {                                     // TAYLOR_SIN
        const double xx = x * x;
        double p = fma_(xx, S5, S4);
        p = fma_(xx, p, S3);
        p = fma_(xx, p, S2);
        p = fma_(xx, p, S1);
        const double t = fma_(xx, fma_(p, x, -(0.5 * dx)), dx);
        return x + t;
    }
    if (x <= 0) dx = -dx;
    const double u = BIG + fabs(x);
    x = fabs(x) - (u - BIG);
    const int k = (int)(uint32_t)asu64(u) * 4;
    const double sn = SINCOSTAB[k], ssn = SINCOSTAB[k + 1], cs = SINCOSTAB[k + 2], ccs = SINCOSTAB[k + 3];
    const double xx = x * x;
    const double s = x + fma_(x * xx, fma_(xx, SN5, SN3), dx);
    const double c = fma_(x, dx, xx * fma_(xx, fma_(xx, CS6, CS4), CS2));
    const double cor = fma_(s, cs, fma_(-c, sn, fma_(s, ccs, ssn)));
    return copysign(sn + cor, xold);
}
// s_sin.c do_cos: cos(x + dx) for |x| < 0.855
OCTA_HD inline double do_cos(double x, double dx) {
    if (x < 0) dx = -dx;
    const double u = BIG + fabs(x);
    x = fabs(x) - (u - BIG) + dx;
    const int k = (int)(uint32_t)asu64(u) * 4;
    const double sn = SINCOSTAB[k], ssn = SINCOSTAB[k + 1], cs = SINCOSTAB[k + 2], ccs = SINCOSTAB[k + 3];
    const double xx = x * x;
    const double s = fma_(x * xx, fma_(xx, SN5, SN3), x);
    const double c = xx * fma_(xx, fma_(xx, CS6, CS4), CS2);
    const double cor = fma_(-s, sn, fma_(-c, cs, fma_(-s, ssn, ccs)));
    return cs + cor;
}
OCTA_HD inline double gsin(double x) {
    const uint32_t k = (uint32_t)(asu64(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                             // |x| < 2^-26
    if (k < 0x3feb6000u) return do_sin(x, 0.0);                 // |x| < 0.855469
    if (k < 0x400368fdu) return copysign(do_cos(HP0 - fabs(x), HP1), x);   // |x| < 2.426265
    return NAN;
}
OCTA_HD inline double gcos(double x) {
    const uint32_t k = (uint32_t)(asu64(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;                           // |x| < 2^-27
    if (k < 0x3feb6000u) return do_cos(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = HP0 - fabs(x), a = y + HP1, da = (y - a) + HP1;
        return do_sin(a, da);
    }
    return NAN;
}


// ---- e_asin.c __ieee754_acos, 0 < x <= 1 ---------------------------------------------------------------------------------
OCTA_CONST double F1 = 0x1.55555555554f9p-3, F2 = 0x1.333333336127dp-4, F3 = 0x1.6db6dae42c0e4p-5, F4 = 0x1.f1c7e04f4ad99p-6,
                  F5 = 0x1.6e442c822d419p-6, F6 = 0x1.292d80f453c72p-6;
OCTA_CONST double RT0 = 0x1.fffffffecc1ddp-1, RT1 = 0x1.fffffff757304p-2, RT2 = 0x1.800496769c91ap-2, RT3 = 0x1.4006318d1dab9p-2;
OCTA_CONST double T27 = 0x1.0p+27;

// table branches: x near the knot ASNCS[n]; M coefficients of the local polynomial, then the tail and acos of the knot
template <int M>
OCTA_HD inline double acos_tab(double x, int n) {
    const double xx = x - ASNCS[n];
    double p = ASNCS[n + M];
#pragma unroll
    for (int j = M - 1; j >= 2; j--) p = fma_(xx, p, ASNCS[n + j]);
    p = fma_(xx * xx, p, ASNCS[n + M + 1]);
    const double t = fma_(xx, ASNCS[n + 1], p);
    const double y = HP0 - ASNCS[n + M + 2];
    return (HP1 - t) + y;
}

OCTA_HD inline double gacos(double x) {
    const uint64_t bits = asu64(x);
    const uint32_t k = (uint32_t)(bits >> 32);
    if (k & 0x80000000u) return NAN;                           // negative arguments: outside the restated domain
    if (k < 0x3c880000u) return HP0;                           // x < 2^-55
    if (k < 0x3fc00000u) {                                     // x < 0.125
        const double x2 = x * x;
        double p = fma_(x2, F6, F5);
        p = fma_(x2, p, F4);
        p = fma_(x2, p, F3);
        p = fma_(x2, p, F2);
        p = fma_(x2, p, F1);
        const double r = HP0 - x;
        const double cor = fma_(-p, x * x2, ((HP0 - r) - x) + HP1);
        return r + cor;
    }
    if (k < 0x3fd00000u) return acos_tab<6>(x, 11 * (int)((k >> 15) & 0x1fu));            // < 0.25
    if (k < 0x3fe00000u) return acos_tab<6>(x, 352 + 11 * (int)((k >> 14) & 0x3fu));      // < 0.5
    if (k < 0x3fe80000u) return acos_tab<7>(x, 1056 + 3 * (int)((k >> 11) & 0x1fcu));     // < 0.75
    if (k < 0x3fed8000u) return acos_tab<8>(x, 992 + 13 * (int)((k >> 13) & 0x7fu));      // < 0.921875
    if (k < 0x3fee8000u) return acos_tab<9>(x, 884 + 14 * (int)((k >> 13) & 0x7fu));      // < 0.953125
    if (k < 0x3fef0000u) return acos_tab<10>(x, 768 + 15 * (int)((k >> 13) & 0x7fu));     // < 0.96875
    if (k < 0x3ff00000u) {                                     // < 1: acos x = 2 asin(sqrt((1 - x) / 2)), square root by table + Newton
        const double z = (1.0 - x) * 0.5;
        const uint64_t zb = asu64(z);
        double t = INROOT[(zb >> 46) & 0x7f] * ::ldexp(1.0, (int)(0x1ff - (int)(zb >> 53)));   // powtwo[i] = 2^i
        const double r = fma_(-(t * t), z, 1.0);
        double q = fma_(r, RT3, RT2);
        q = fma_(r, q, RT1);
        q = fma_(r, q, RT0);
        t = q * t;
        const double c = z * t;
        const double e = fma_(-c, t * 0.5, 1.5);
        const double w = fma_(c, T27, c);
        const double y = fma_(-T27, c, w);
        const double den = fma_(e, c, y);
        const double cc = fma_(-y, y, z) / den;
        double p = fma_(z, F6, F5);
        p = fma_(z, p, F4);
        p = fma_(z, p, F3);
        p = fma_(z, p, F2);
        p = fma_(z, p, F1);
        p = p * z;
        const double res = (cc + p * (y + cc)) + y;
        return res + res;
    }
    if (bits == 0x3ff0000000000000ull) return 0.0;
    return NAN;
}

}  // namespace octa_gtrig
