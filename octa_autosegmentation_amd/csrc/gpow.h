// gpow.h -- bit-exact restatement of glibc's pow() (x86_64 FMA variant) for host and device.
//
// Why: vessel radii are written with 17 significant digits (generate_vessel_graph.py:66) and are
// produced by CPython float `**` = glibc pow in Murray's law (arterial_tree.py:180,
// greenhouse.py:206,264). ROCm's pow is not bit-identical to glibc's, so the simulator kernels
// evaluate the published glibc algorithm (ARM optimized-routines pow: log via a 128-entry table and
// a degree-8 polynomial, exp via a 128-entry 2^(k/128) table and a degree-5 polynomial) with the
// same operation order and the same fused multiply-adds the FMA build of glibc executes. Constants:
// glibc_pow_tables.h (generated from the image's libm by tools/gen_pow_tables.py).
// Domain: finite x > 0 (normal), finite y with the result neither overflowing nor underflowing --
// all the simulator needs (radii ~1e-4..1e-1, exponents kappa, 1/kappa, 2, 4, 5). Outside that
// domain the function returns NaN so a misuse is loud. tests/test_sim_core.py compares it with
// math.pow over millions of inputs.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#ifndef OCTA_HD
#if defined(__HIP__) || defined(__HIPCC__)
#define OCTA_HD __host__ __device__
#else
#define OCTA_HD
#endif
#endif
#define OCTA_CONST static constexpr
#ifndef OCTA_UNLIKELY
#define OCTA_UNLIKELY(c) __builtin_expect(!!(c), 0)      // cold blocks leave the hot path's cache lines (the simulator kernel is short of instruction cache: DESIGN.md 4.1)
#endif
#include "glibc_pow_tables.h"

namespace octa_gpow {

OCTA_HD inline uint64_t asu64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
OCTA_HD inline double asf64(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
OCTA_HD inline double fma_(double a, double b, double c) { return ::fma(a, b, c); }

// log_tab / exp_tab: LOG_TAB / EXP_TAB or copies of them (the ordered pass keeps copies in LDS)
// UNI: every lane of the wave passes the same (x, y) -- the table indices are declared wave-uniform, so that with the constant tables
// (LOG_TAB / EXP_TAB) the entries come through the scalar data cache instead of an LDS / vector-memory round trip per lookup.
template <bool UNI>
OCTA_HD inline double gpow_impl(double x, double y, const double *log_tab, const uint64_t *exp_tab) {
    const uint64_t ix = asu64(x), iy = asu64(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
    // glibc's special-case gate: x subnormal/zero/negative/inf/nan, or |y| tiny/huge/inf/nan
    if (OCTA_UNLIKELY(topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu)) return NAN;
    // ---- log_inline
    const uint64_t OFF = 0x3fe6955500000000ULL;
    uint64_t tmp = ix - OFF;
    int i = (int)((tmp >> (52 - 7)) % 128);
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNI) i = __builtin_amdgcn_readfirstlane(i);
#endif
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & (0xfffULL << 52));
    double z = asf64(iz), kd = (double)k;
    double invc = log_tab[3 * i], logc = log_tab[3 * i + 1], logctail = log_tab[3 * i + 2];
    double r = fma_(z, invc, -1.0);
    double t1 = fma_(kd, LN2HI, logc);
    double t2 = t1 + r;
    double lo1 = fma_(kd, LN2LO, logctail);
    double lo2 = t1 - t2 + r;
    double ar = LOG_POLY[0] * r;
    double ar2 = r * ar;
    double ar3 = r * ar2;
    double hi = t2 + ar2;
    double lo3 = fma_(ar, r, -ar2);
    double lo4 = t2 - hi + ar2;
    double q3 = fma_(r, LOG_POLY[6], LOG_POLY[5]);
    double q2 = fma_(ar2, q3, fma_(r, LOG_POLY[4], LOG_POLY[3]));
    double q1 = fma_(ar2, q2, fma_(r, LOG_POLY[2], LOG_POLY[1]));
    double p = ar3 * q1;
    double lo = lo1 + lo2 + lo3 + lo4 + p;
    double ylog = hi + lo;
    double tail = hi - ylog + lo;
    // ---- y * log(x) in double-double
    double ehi = y * ylog;
    double elo = fma_(y, tail, fma_(y, ylog, -ehi));
    // ---- exp_inline
    uint32_t abstop = (uint32_t)(asu64(ehi) >> 52) & 0x7ffu;
    // |ehi| in [2^-54, 2^9): the only branch the simulator can reach
    if (OCTA_UNLIKELY(abstop - 0x3c9u >= 0x408u - 0x3c9u)) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + ehi;  // tiny exponent: glibc returns 1 + x (WANT_ROUNDING)
        return NAN;
    }
    double kd2 = fma_(INVLN2N, ehi, SHIFT);
    uint64_t ki = asu64(kd2);
    kd2 -= SHIFT;
    double rr = fma_(kd2, NEGLN2LON, fma_(kd2, NEGLN2HIN, ehi));
    rr += elo;
    uint64_t idx = 2 * (ki % 128);
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNI) idx = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)idx);
#endif
    uint64_t top = ki << (52 - 7);
    double etail = asf64(exp_tab[idx]);
    uint64_t sbits = exp_tab[idx + 1] + top;
    double r2 = rr * rr;
    double s1 = etail + rr;
    double s2 = fma_(r2, fma_(rr, EXP_POLY[1], EXP_POLY[0]), s1);
    double tm = fma_(r2 * r2, fma_(rr, EXP_POLY[3], EXP_POLY[2]), s2);
    double scale = asf64(sbits);
    return fma_(scale, tm, scale);
}

OCTA_HD inline double gpow_t(double x, double y, const double *log_tab, const uint64_t *exp_tab) { return gpow_impl<false>(x, y, log_tab, exp_tab); }
OCTA_HD inline double gpow(double x, double y) { return gpow_t(x, y, LOG_TAB, EXP_TAB); }
// wave-uniform arguments (device: all lanes of the calling wave active and equal)
OCTA_HD inline double gpow_u(double x, double y) { return gpow_impl<true>(x, y, LOG_TAB, EXP_TAB); }

}  // namespace octa_gpow
