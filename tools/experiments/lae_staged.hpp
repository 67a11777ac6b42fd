// The pair of logaddexp's of a tree merge (trees.jl:145,249) as a STAGED computation.
//
// A wave-per-chain kernel has one wave per SIMD, so a dependent chain of scalar fp64 operations (exp, log, two
// divisions: ≈45 levels) runs at the pipeline latency with nothing to fill the gaps, and the compiler's scheduler keeps
// source order in these register-bound kernels.  The merge's vector arithmetic is independent of the chain, so the
// chain is cut into stages and the stages are called from inside the vector loops (one or two levels per loop
// iteration, the tail between the steps of the butterfly reduction).
//
// The arithmetic is det_logaddexp's (include/dhmc_detmath.h), operation for operation, restricted to what can occur
// here: the exponential's argument is -|x - y| <= 0 (or NaN), the logarithm's 1 + u with u in [0, 1].  The divisions are
// the IEEE-correct sequence the compiler emits for `/` (v_div_scale, v_rcp, Newton steps, v_div_fmas, v_div_fixup),
// written out so that it can be cut too; a correctly rounded quotient has one value, the CPU's.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dhmc_detmath.h"
#include "wave.hpp"

namespace dhmc {

// x / y in three parts
struct DivStaged {
    double x, y, ds0, rcp, f1, f3, ds1, mul, f4, q;
    bool flag;
    __device__ __forceinline__ void p0(double x_, double y_) {
        x = x_; y = y_;
        bool dummy;
        ds0 = __builtin_amdgcn_div_scale(x, y, false, &dummy);   // denominator, scaled
        ds1 = __builtin_amdgcn_div_scale(x, y, true, &flag);     // numerator, scaled
        rcp = __builtin_amdgcn_rcp(ds0);
    }
    __device__ __forceinline__ void p1() {
        const double f0 = __builtin_fma(-ds0, rcp, 1.0);
        f1 = __builtin_fma(rcp, f0, rcp);
    }
    __device__ __forceinline__ void p2() {
        const double f2 = __builtin_fma(-ds0, f1, 1.0);
        f3 = __builtin_fma(f1, f2, f1);
    }
    __device__ __forceinline__ void p3() {
        mul = ds1 * f3;
        f4 = __builtin_fma(-ds0, mul, ds1);
    }
    __device__ __forceinline__ void p4() {
        const double fm = __builtin_amdgcn_div_fmas(f4, f3, mul, flag);
        q = __builtin_amdgcn_div_fixup(fm, y, x);
    }
};

// no side job
struct NoSide {
    static constexpr int kLoopStages = 0;
    __device__ __forceinline__ void stage(int) {}
    __device__ __forceinline__ void post(int) {}
};

// r1 = logaddexp(a1, b1), r2 = logaddexp(a2, b2): even lanes evaluate the first, odd lanes the second (the values are
// wave-uniform; two different ones in different lanes cost one chain).  16 loop stages + 7 post stages; results valid
// after post(6).
struct LaePairJob {
    static constexpr int kLoopStages = 16;
    double a1, b1, a2, b2;
    int lane;
    double r1, r2;
    // state
    double x, y, mx, xr, kd, r, e2, e4, e8, A0, A1, A2, A3, A4, A5, A6, B0, B1, B2, C0, C1, pw1, pw2, u, w, wr, m, f, hfsq, cn;
    double z, z2, z4, z8, q0, q1, q2, q3, q4, q5, t0, t1, t2, u0, R, dk, inner, lw, res;
    bool nanf, exnan, under, tiny, bad;
    DivStaged d1, d2;

    __device__ __forceinline__ LaePairJob(double a1_, double b1_, double a2_, double b2_, int lane_)
        : a1(a1_), b1(b1_), a2(a2_), b2(b2_), lane(lane_) {}

    __device__ __forceinline__ void stage(int s) {
        const double INV_LN2 = 1.44269504088896338700e+00;
        const double LN2_HI = 6.93147180369123816490e-01;
        const double LN2_LO = 1.90821492927058770002e-10;
        switch (s) {
        case 0: {
            const bool odd = (lane & 1) != 0;
            x = odd ? a2 : a1;
            y = odd ? b2 : b1;
            const double d = dm_sel(x == y, 0.0, __builtin_fabs(x - y));
            mx = dm_sel(x > y, x, y);
            nanf = dm_isnan(x) || dm_isnan(y);
            const double xe = -d;                              // det_exp(xe), xe <= 0 or NaN
            exnan = dm_isnan(xe);
            under = xe < -745.2;
            xr = dm_sel(exnan || under, 0.0, xe);
            kd = __builtin_floor(xr * INV_LN2 + 0.5);
        } break;
        case 1: {
            r = dm_fma(-kd, LN2_HI, xr);
            r = dm_fma(-kd, LN2_LO, r);
            const int k = (int)kd;
            const int k1 = k >> 1;
            const int k2 = k - k1;
            pw1 = dm_pow2(k1);
            pw2 = dm_pow2(k2);
        } break;
        case 2: {
            e2 = r * r;
            A0 = dm_fma(1.0, r, 1.0);
            A1 = dm_fma(1.0 / 6.0, r, 0.5);
            A2 = dm_fma(1.0 / 120.0, r, 1.0 / 24.0);
            A3 = dm_fma(1.0 / 5040.0, r, 1.0 / 720.0);
        } break;
        case 3: {
            A4 = dm_fma(1.0 / 362880.0, r, 1.0 / 40320.0);
            A5 = dm_fma(1.0 / 39916800.0, r, 1.0 / 3628800.0);
            A6 = dm_fma(1.0 / 6227020800.0, r, 1.0 / 479001600.0);
            e4 = e2 * e2;
            B0 = dm_fma(A1, e2, A0);
            B1 = dm_fma(A3, e2, A2);
        } break;
        case 4: {
            B2 = dm_fma(A5, e2, A4);
            e8 = e4 * e4;
            C0 = dm_fma(B1, e4, B0);
            C1 = dm_fma(A6, e4, B2);
        } break;
        case 5: {
            const double p = dm_fma(C1, e8, C0);
            const double ex = (p * pw1) * pw2;
            u = dm_sel(exnan, -__builtin_fabs(x - y), dm_sel(under, 0.0, ex));   // NaN in -> that NaN out (det_exp returns x)
            w = 1.0 + u;
        } break;
        case 6: {
            tiny = w == 1.0;
            bad = !dm_isfinite(w);
            wr = dm_sel(bad, 1.0, w);                           // in [1, 2]: det_log's regular path
            const uint64_t ub = dm_bits(wr);
            int e = (int)(ub >> 52) - 1023;
            const uint64_t mant = ub & 0x000fffffffffffffull;
            const double m1 = dm_from_bits(mant | 0x3ff0000000000000ull);
            const bool big = mant > 0x6a09e667f3bcdull;
            m = dm_sel(big, m1 * 0.5, m1);
            e += big ? 1 : 0;
            dk = (double)e;
        } break;
        case 7: {
            f = m - 1.0;
            cn = u - (wr - 1.0);
            d1.p0(f, 2.0 + f);                                  // s = f / (2 + f)
        } break;
        case 8: {
            d2.p0(cn, wr);                                      // (u - (w - 1)) / w
            d1.p1();
            hfsq = 0.5 * f * f;
        } break;
        case 9: { d1.p2(); d2.p1(); } break;
        case 10: { d1.p3(); d2.p2(); } break;
        case 11: { d1.p4(); d2.p3(); } break;
        case 12: {
            z = d1.q * d1.q;
            d2.p4();
        } break;
        case 13: {
            z2 = z * z;
            q0 = dm_fma(2.0 / 5.0, z, 2.0 / 3.0);
            q1 = dm_fma(2.0 / 9.0, z, 2.0 / 7.0);
            q2 = dm_fma(2.0 / 13.0, z, 2.0 / 11.0);
            q3 = dm_fma(2.0 / 17.0, z, 2.0 / 15.0);
        } break;
        case 14: {
            q4 = dm_fma(2.0 / 21.0, z, 2.0 / 19.0);
            q5 = dm_fma(2.0 / 25.0, z, 2.0 / 23.0);
            z4 = z2 * z2;
            t0 = dm_fma(q1, z2, q0);
            t1 = dm_fma(q3, z2, q2);
        } break;
        case 15: {
            t2 = dm_fma(q5, z2, q4);
            z8 = z4 * z4;
            u0 = dm_fma(t1, z4, t0);
        } break;
        default: break;
        }
    }
    // between the steps of the reduction that follows the loop
    __device__ __forceinline__ void post(int s) {
        const double LN2_HI = 6.93147180369123816490e-01;
        const double LN2_LO = 1.90821492927058770002e-10;
        switch (s) {
        case 0: R = dm_fma(t2, z8, u0); break;
        case 1: R = R * z; break;
        case 2: inner = d1.q * (hfsq + R) + dk * LN2_LO; break;
        case 3: lw = dk * LN2_HI - ((hfsq - inner) - f); break;
        case 4: res = lw + d2.q; break;
        case 5: res = mx + dm_sel(tiny, u, dm_sel(bad, w, res)); break;
        case 6: {
            res = dm_sel(nanf, dm_nan(), res);
            r1 = readlane_f64(res, 0);
            r2 = readlane_f64(res, 1);
        } break;
        default: break;
        }
    }
    // everything at once (callers without a loop to hide it in)
    __device__ __forceinline__ void run_all() {
#pragma unroll
        for (int s = 0; s < kLoopStages; ++s) stage(s);
#pragma unroll
        for (int s = 0; s < 7; ++s) post(s);
    }
};

// the stages that belong to loop iteration k of NPL
template <int NPL, class SIDE>
__device__ __forceinline__ void side_stages(SIDE& side, int k) {
    constexpr int NS = SIDE::kLoopStages;
#pragma unroll
    for (int s = (k * NS) / NPL; s < ((k + 1) * NS) / NPL; ++s) side.stage(s);
}

// wave_allreduce with the side job's tail between its steps
template <int N, class SIDE>
__device__ __forceinline__ void wave_allreduce_side(double (&v)[N], SIDE& side) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0xB1>(v[i]);
    side.post(0);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x4E>(v[i]);
    side.post(1);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x141>(v[i]);
    side.post(2);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x140>(v[i]);
    side.post(3);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_rows_f64<0x142, 0xA>(v[i]);
    side.post(4);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_rows_f64<0x143, 0xC>(v[i]);
    side.post(5);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = readlane_f64(v[i], 63);
    side.post(6);
}

}  // namespace dhmc
