/* dhmc_detmath.h — the numerical contract of the dhmc C ABI.
 *
 * The reference (tpapp/DynamicHMC.jl) calls Julia's libm-class functions for the handful of
 * scalar transcendentals on the NUTS hot path:
 *   - logaddexp           trees.jl:145, NUTS.jl:70   (LogExpFunctions, not vendored)
 *   - exp                 NUTS.jl:87 (acceptance_rate), stepsize.jl:163,170 (current_ϵ, final_ϵ)
 *   - log, sqrt, ^        stepsize.jl:136-137,153-154 (dual averaging)
 *   - randn / randexp     hamiltonian.jl:124, NUTS.jl:44 (Julia stdlib ziggurats, not vendored)
 * None of those pin a bit pattern (the reference's own tests are tolerance-based), and the
 * vendor device libm and glibc disagree in the last ulp.  So that the CPU oracle and the HIP
 * kernels can be compared bit-for-bit over whole runs, the ABI fixes these functions to the
 * algorithms below, built only from IEEE-754 correctly rounded +,-,*,/,sqrt,fma and integer
 * bit operations.  Every function is within 1–2 ulp of the correctly rounded result (checked
 * against libm in tests/test_detmath.py), i.e. inside the tolerance class of the reference.
 *
 * Plain C++ (no HIP types): compiled by g++ for the oracle and by hipcc for the device code.
 * Compile BOTH sides with -ffp-contract=off; every fused multiply-add below is explicit.
 */
#ifndef DHMC_DETMATH_H
#define DHMC_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define DHMC_HD __host__ __device__ inline
#else
#define DHMC_HD inline
#endif

namespace dhmc {

DHMC_HD double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

DHMC_HD uint64_t dm_bits(double x) {
    union { double d; uint64_t u; } c; c.d = x; return c.u;
}
DHMC_HD double dm_from_bits(uint64_t u) {
    union { double d; uint64_t u; } c; c.u = u; return c.d;
}
DHMC_HD double dm_inf() { return dm_from_bits(0x7ff0000000000000ull); }
DHMC_HD double dm_nan() { return dm_from_bits(0x7ff8000000000000ull); }
DHMC_HD bool dm_isnan(double x) { return x != x; }
DHMC_HD bool dm_isfinite(double x) {
    return (dm_bits(x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
}
/* 2^k for -1022 <= k <= 1023 */
DHMC_HD double dm_pow2(int k) { return dm_from_bits((uint64_t)(k + 1023) << 52); }

/* exp(x): Cody–Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial
 * (remainder < 5e-18 relative, Estrin evaluation), result scaled by 2^k in two exact-or-once-rounded steps. */
DHMC_HD double det_exp(double x) {
    if (dm_isnan(x)) return x;
    if (x > 709.782712893384) return dm_inf();
    if (x < -745.2) return 0.0;
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
    const double LN2_LO = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
    double kd = __builtin_floor(x * INV_LN2 + 0.5);
    int k = (int)kd;
    double r = dm_fma(-kd, LN2_HI, x);
    r = dm_fma(-kd, LN2_LO, r);
    /* Σ_{n<=13} r^n/n! by Estrin's scheme (4 dependent levels instead of Horner's 14: these scalar chains sit on
     * the critical path of every tree merge) — the evaluation order below IS the definition. */
    const double r2 = r * r;
    const double r4 = r2 * r2;
    const double r8 = r4 * r4;
    const double a0 = dm_fma(1.0, r, 1.0);
    const double a1 = dm_fma(1.0 / 6.0, r, 0.5);
    const double a2 = dm_fma(1.0 / 120.0, r, 1.0 / 24.0);
    const double a3 = dm_fma(1.0 / 5040.0, r, 1.0 / 720.0);
    const double a4 = dm_fma(1.0 / 362880.0, r, 1.0 / 40320.0);
    const double a5 = dm_fma(1.0 / 39916800.0, r, 1.0 / 3628800.0);
    const double a6 = dm_fma(1.0 / 6227020800.0, r, 1.0 / 479001600.0);
    const double b0 = dm_fma(a1, r2, a0);
    const double b1 = dm_fma(a3, r2, a2);
    const double b2 = dm_fma(a5, r2, a4);
    const double c0 = dm_fma(b1, r4, b0);
    const double c1 = dm_fma(a6, r4, b2);
    const double p = dm_fma(c1, r8, c0);
    int k1 = k >> 1;          /* arithmetic shift: floor(k/2) */
    int k2 = k - k1;
    return (p * dm_pow2(k1)) * dm_pow2(k2);
}

/* log(x): x = 2^e m, m in [sqrt(1/2), sqrt(2)); f = m-1, s = f/(2+f);
 * log(m) = f - f^2/2 + s (f^2/2 + R(s^2)), R(z) = sum_{n>=1} 2/(2n+1) z^n truncated at n=12
 * (|z| <= 0.0295, remainder < 3e-20); e ln2 added in hi/lo parts. */
DHMC_HD double det_log(double x) {
    if (dm_isnan(x)) return x;
    if (x < 0.0) return dm_nan();
    if (x == 0.0) return -dm_inf();
    if (!dm_isfinite(x)) return x;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    int e = 0;
    uint64_t u = dm_bits(x);
    if ((u >> 52) == 0) {            /* subnormal: scale by 2^54 (exact) */
        x = x * 18014398509481984.0;
        u = dm_bits(x);
        e = -54;
    }
    e += (int)(u >> 52) - 1023;
    uint64_t mant = u & 0x000fffffffffffffull;
    double m = dm_from_bits(mant | 0x3ff0000000000000ull);   /* [1,2) */
    if (mant > 0x6a09e667f3bcdull) {                         /* m > sqrt(2) */
        m = m * 0.5;
        e += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* R(z) = z (2/3 + 2/5 z + ... + 2/25 z^11), Estrin */
    const double z2 = z * z;
    const double z4 = z2 * z2;
    const double z8 = z4 * z4;
    const double q0 = dm_fma(2.0 / 5.0, z, 2.0 / 3.0);
    const double q1 = dm_fma(2.0 / 9.0, z, 2.0 / 7.0);
    const double q2 = dm_fma(2.0 / 13.0, z, 2.0 / 11.0);
    const double q3 = dm_fma(2.0 / 17.0, z, 2.0 / 15.0);
    const double q4 = dm_fma(2.0 / 21.0, z, 2.0 / 19.0);
    const double q5 = dm_fma(2.0 / 25.0, z, 2.0 / 23.0);
    const double t0 = dm_fma(q1, z2, q0);
    const double t1 = dm_fma(q3, z2, q2);
    const double t2 = dm_fma(q5, z2, q4);
    const double u0 = dm_fma(t1, z4, t0);
    double R = dm_fma(t2, z8, u0);
    R = R * z;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

/* log1p(u) for u >= 0 (the only use is logaddexp, u = exp(-|d|) in [0,1]). */
DHMC_HD double det_log1p_nonneg(double u) {
    double w = 1.0 + u;
    if (w == 1.0) return u;
    if (!dm_isfinite(w)) return w;
    return det_log(w) + (u - (w - 1.0)) / w;
}

/* logaddexp(x, y) = max(x,y) + log1p(exp(-|x-y|)), with Δ := 0 when x == y so that
 * (-Inf, -Inf) -> -Inf; restates LogExpFunctions.logaddexp (call sites trees.jl:145,
 * NUTS.jl:70). */
DHMC_HD double det_logaddexp(double x, double y) {
    double d = (x == y) ? 0.0 : __builtin_fabs(x - y);
    double mx = (x > y) ? x : y;
    if (dm_isnan(x) || dm_isnan(y)) return dm_nan();
    return mx + det_log1p_nonneg(det_exp(-d));
}

/* sin and cos of 2*pi*a for a in [0,1): t = 4a, quadrant n = floor(t+1/2), f = t-n exact,
 * angle x = f*pi/2 in [-pi/4, pi/4]; Taylor polynomials to x^17 / x^16. */
DHMC_HD void det_sincos2pi(double a, double* sn, double* cs) {
    double t = 4.0 * a;
    double nd = __builtin_floor(t + 0.5);
    int n = (int)nd;
    double f = t - nd;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double x = f * PIO2_HI;
    double xlo = dm_fma(f, PIO2_HI, -x) + f * PIO2_LO;
    double z = x * x;
    /* sin(x) = x + x^3 S(z) */
    double S = 1.0 / 355687428096000.0;              /*  1/17! */
    S = dm_fma(S, z, -1.0 / 1307674368000.0);        /* -1/15! */
    S = dm_fma(S, z, 1.0 / 6227020800.0);            /*  1/13! */
    S = dm_fma(S, z, -1.0 / 39916800.0);             /* -1/11! */
    S = dm_fma(S, z, 1.0 / 362880.0);                /*  1/9!  */
    S = dm_fma(S, z, -1.0 / 5040.0);                 /* -1/7!  */
    S = dm_fma(S, z, 1.0 / 120.0);                   /*  1/5!  */
    S = dm_fma(S, z, -1.0 / 6.0);                    /* -1/3!  */
    /* cos(x) = 1 - z/2 + z^2 C(z) */
    double C = -1.0 / 6402373705728000.0;            /* -1/18! */
    C = dm_fma(C, z, 1.0 / 20922789888000.0);        /*  1/16! */
    C = dm_fma(C, z, -1.0 / 87178291200.0);          /* -1/14! */
    C = dm_fma(C, z, 1.0 / 479001600.0);             /*  1/12! */
    C = dm_fma(C, z, -1.0 / 3628800.0);              /* -1/10! */
    C = dm_fma(C, z, 1.0 / 40320.0);                 /*  1/8!  */
    C = dm_fma(C, z, -1.0 / 720.0);                  /* -1/6!  */
    C = dm_fma(C, z, 1.0 / 24.0);                    /*  1/4!  */
    double s0 = dm_fma(x * z, S, xlo) + x;           /* cos(x)*xlo ~ xlo to first order */
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    double c0 = w + (((1.0 - w) - hz) + dm_fma(z * z, C, -x * xlo));
    switch (n & 3) {
    case 0: *sn = s0;  *cs = c0;  break;
    case 1: *sn = c0;  *cs = -s0; break;
    case 2: *sn = -s0; *cs = -c0; break;
    default: *sn = -c0; *cs = s0; break;
    }
}

/* x^y for x > 0 through exp(y log x); used only for m^(-κ) in dual averaging
 * (stepsize.jl:154), where |y log x| <= ~7 so the result is within ~8 ulp. */
DHMC_HD double det_pow_pos(double x, double y) { return det_exp(y * det_log(x)); }

/* Uniform doubles from 64 random bits. */
DHMC_HD double u01_open_closed(uint64_t r) { /* (0,1] */
    return (double)((r >> 11) + 1ull) * 1.1102230246251565404e-16; /* 2^-53 */
}
DHMC_HD double u01_closed_open(uint64_t r) { /* [0,1) */
    return (double)(r >> 11) * 1.1102230246251565404e-16;
}

/* Exp(1) draw; stands in for Random.randexp (NUTS.jl:44). */
DHMC_HD double det_randexp(uint64_t r) { return -det_log(u01_open_closed(r)); }

/* Two independent N(0,1) draws by Box–Muller; stands in for Random.randn
 * (hamiltonian.jl:124). */
DHMC_HD void det_randn2(uint64_t r1, uint64_t r2, double* z0, double* z1) {
    double u1 = u01_open_closed(r1);
    double u2 = u01_closed_open(r2);
    double rad = __builtin_sqrt(-2.0 * det_log(u1));
    double sn, cs;
    det_sincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

} /* namespace dhmc */
#endif
