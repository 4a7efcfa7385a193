/* oracle/smc_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Portable, libm-free double-precision exp/log built only from IEEE-754
 * + - * / (no FMA contraction: compile with -ffp-contract=off), so that the
 * HIP kernels (which carry their own, independently written copy of the same
 * published algorithms) produce bit-identical values.  Algorithms: the classic
 * Sun fdlibm e_exp.c / e_log.c argument reductions and minimax polynomials
 * (public domain; constants are the published fdlibm constants).
 *
 * fastexp()/exp_digamma() restate the reference's helpers:
 *   /root/reference/src/particle.cpp:30-40  (fastexp)
 *   /root/reference/src/particle.cpp:65-74  (exp_digamma)
 */
#ifndef SMC_ORACLE_MATH_H
#define SMC_ORACLE_MATH_H

#include <cstdint>
#include <cstring>

namespace smco {

static inline uint64_t d2u(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }

/* exp(x) for finite x.  Result within ~1 ulp of libm. */
static inline double smc_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (x > 709.782712893383973096) return u2d(0x7ff0000000000000ULL);  /* +inf */
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    int k = 0;
    double hi = x, lo = 0.0;
    if (ax > 0.34657359027997264) {                 /* |x| > 0.5 ln2 */
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {        /* |x| < 2^-28 */
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) {
        return u2d(d2u(y) + ((uint64_t)(int64_t)k << 52));
    } else {
        y = u2d(d2u(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
        return y * 9.33263618503218878990e-302;     /* 2^-1000 */
    }
}

/* log(x) for finite x > 0. */
static inline double smc_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    int k = 0;
    uint64_t ux = d2u(x);
    int32_t hx = (int32_t)(ux >> 32);
    if (hx < 0x00100000) {                           /* subnormal: scale up */
        k -= 54;
        x *= 18014398509481984.0;                    /* 2^54 */
        ux = d2u(x);
        hx = (int32_t)(ux >> 32);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    ux = (ux & 0x00000000ffffffffULL) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    x = u2d(ux);
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {               /* |f| < 2^-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}

/* reference: particle.cpp:30-40 */
/* particle.cpp:45-55: one-division continued-fraction form for x^2 < 2.099166 (rel. error < 1e-2), exp otherwise */
static inline double fastexp_approx(double x) {
    double xx = x * x;
    if (xx < 2.099166) return 1 + 2 * x / (2 - x + xx * (1.0 / 6));
    return smc_exp(x);
}

static inline double fastexp(double x) {
    double xx = x * x;
    if (xx < 0.516167859) {
        return 1 + 2 * x / (2 - x + xx / (6 + xx * 0.1));
    } else {
        return smc_exp(x);
    }
}

/* reference: particle.cpp:65-74 */
static inline double exp_digamma(double x) {
    if (x > 10) return x - 0.5 + (x + 0.5) / (24 * x * x);
    double f = 0.0;
    while (x < 6) {
        f = f + 1.0 / x;
        x = x + 1.0;
    }
    double psi = smc_log(x) - 1 / (2 * x) - 1 / (12 * x * x);
    return smc_exp(psi - f);
}

/* ---- counter-based RNG: Philox4x32-10 (Salmon et al. 2011, published constants) ---- */
struct Philox {
    static inline void round(uint32_t c[4], const uint32_t k[2]) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c[1] ^ k[0];
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c[3] ^ k[1];
        uint32_t n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static inline void block(uint32_t c[4], uint32_t k0, uint32_t k1) {
        uint32_t k[2] = {k0, k1};
        for (int r = 0; r < 10; ++r) {
            round(c, k);
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
    }
};

/* One uniform in (0,1) per (seed, slot, stream, draw index): 53 random bits + 0.5, * 2^-53. */
static inline double philox_uniform(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t draw) {
    uint32_t c[4] = {(uint32_t)draw, (uint32_t)(draw >> 32), slot, stream};
    Philox::block(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 11;
    return ((double)bits + 0.5) * 1.1102230246251565e-16;   /* 2^-53 */
}

/* Both halves of the Philox block at one draw index: the first equals philox_uniform, the second comes from the
 * block's other two words.  A genealogy update takes its four uniforms from two blocks this way. */
static inline void philox_pair(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t draw, double* u0, double* u1) {
    uint32_t c[4] = {(uint32_t)draw, (uint32_t)(draw >> 32), slot, stream};
    Philox::block(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t b0 = (((uint64_t)c[0] << 32) | c[1]) >> 11;
    uint64_t b1 = (((uint64_t)c[2] << 32) | c[3]) >> 11;
    *u0 = ((double)b0 + 0.5) * 1.1102230246251565e-16;
    *u1 = ((double)b1 + 0.5) * 1.1102230246251565e-16;
}

}  // namespace smco
#endif
