// Device policies of the ABI's scalar math (include/dhmc_detmath.h).  One chain per wavefront means every scalar operation
// of the tree logic costs a full vector issue slot, and a lone wave per SIMD is bound by the NUMBER of instructions it issues
// (docs/DESIGN_history_rounds1-4.md §6a) — so these policies keep every IEEE operation and its order (same bits as the oracle's dm_generic
// instantiation, checked by tests/test_gpu_detmath.py) and change only where operands live:
//
//   dm_uniform   the arguments are wave-uniform (logaddexp pairs of a merge, acceptance rate, step size, the funnel's exp(-v)):
//                a table index goes through v_readfirstlane, so the row arrives by ONE scalar load (s_load_dwordx16 for the
//                eight coefficients of a softplus cell: no vector registers, no vector-memory latency) and the Horner chain
//                runs with its coefficients in scalar registers;
//   dm_vector    the arguments differ per lane (log and sin/cos of the momentum refresh, Exp(1) draws): table rows are
//                gathered per lane, compile-time polynomial coefficients sit in scalar registers.
//
// How the Horner chains get one instruction per step: for fma(s, x, constant) the compiler's own choice is two v_mov_b32 of the
// constant into a vector register pair followed by v_fmac_f64 — three vector instructions instead of one.  Passing each
// coefficient through an EMPTY asm statement with an "s" constraint pins it to a scalar register pair without emitting
// anything, and the compiler then selects v_fma_f64 with the scalar operand (the constant-bus limit of gfx9 allows one per
// instruction: the leading coefficient is the one it moves to the accumulator).  No instruction is written in asm: round 4's
// first version spelled the chains (and Philox's v_mad_u64_u32) out as asm blocks, and gfx950's software-managed hazards —
// two wait states between a VALU write of an SGPR (a v_readlane reload of a spilled SGPR is one) and a VALU read of it, one
// between a VALU write of a VGPR and a v_readfirstlane of it — are tracked by the compiler for its own instructions only:
// stale operands, wild scalar loads, and with the wait states added by hand still two funnel cases at D = 1000 in the fuzz
// sweep whose trees differed from the oracle's.  tools/isa_hazard_verify.py checks the built library for those hazards.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dhmc_detmath.h"

namespace dhmc {

__device__ __forceinline__ double dm_in_sgpr(double c) {
    asm("" : "+s"(c));
    return c;
}

struct dm_sgpr_horner {
    template <int N>
    static __device__ __forceinline__ double run(double x, const double (&c)[N]) {
        double s = dm_in_sgpr(c[N - 1]);
#pragma unroll
        for (int k = N - 2; k >= 0; --k) s = __builtin_fma(s, x, dm_in_sgpr(c[k]));
        return s;
    }
};

struct dm_uniform {
    // The index is uniform and the compiler can prove it — and then keeps it, and the address arithmetic behind it, in vector
    // registers.  The empty asm makes the value opaque, so v_readfirstlane stays and everything derived from it is scalar code.
    // (v_readfirstlane itself comes from the builtin, see above.)
    static __device__ __forceinline__ int idx(int i) {
        asm volatile("" : "+v"(i));
        return (int)__builtin_amdgcn_readfirstlane((unsigned)i);
    }
    template <int N>
    static __device__ __forceinline__ double horner_const(double x, const double (&c)[N]) { return dm_sgpr_horner::run<N>(x, c); }
    template <int N>
    static __device__ __forceinline__ double horner_row(double x, const double (&c)[N]) { return dm_sgpr_horner::run<N>(x, c); }
    static __device__ __forceinline__ double max_nonnan(double x, double y) { return __builtin_fmax(x, y); }   // v_max_f64
    typedef double v8d __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void row8(const double* row, double (&c)[8]) {  // uniform, 64-byte aligned: s_load_dwordx16
        const v8d v = *reinterpret_cast<const v8d*>(row);
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = v[k];
    }
};

struct dm_vector {
    static __device__ __forceinline__ int idx(int i) { return i; }
    template <int N>
    static __device__ __forceinline__ double horner_const(double x, const double (&c)[N]) { return dm_sgpr_horner::run<N>(x, c); }
    template <int N>
    static __device__ __forceinline__ double horner_row(double x, const double (&c)[N]) { return dm_generic::horner_const<N>(x, c); }
    static __device__ __forceinline__ double max_nonnan(double x, double y) { return dm_uniform::max_nonnan(x, y); }
    static __device__ __forceinline__ void row8(const double* row, double (&c)[8]) { dm_generic::row8(row, c); }
};

// Debugging switches (tools/experiments/build_variant_fast.sh): the compiler's own code in place of a policy, per use.
#ifdef DHMC_UNI_GENERIC
typedef dm_generic dm_u;
#else
typedef dm_uniform dm_u;
#endif
#ifdef DHMC_VEC_GENERIC
typedef dm_generic dm_v;
#else
typedef dm_vector dm_v;
#endif
#ifdef DHMC_REXP_GENERIC
typedef dm_generic dm_rexp;
#else
typedef dm_v dm_rexp;
#endif

// wave-uniform arguments
__device__ __forceinline__ double det_exp_u(double x) { return det_exp_t<dm_u>(x); }
__device__ __forceinline__ double det_log_u(double x) { return det_log_t<dm_u>(x); }
__device__ __forceinline__ double det_logaddexp_u(double x, double y) { return det_logaddexp_t<dm_u>(x, y); }
__device__ __forceinline__ double det_pow_pos_u(double x, double y) { return det_pow_pos_t<dm_u>(x, y); }
// The two logaddexp's of a tree merge (visited statistic and ω) together: when both |x - y| < 16 — the common case — the two
// softplus cells are requested back to back and the two Horner chains run interleaved (one basic block); each value is
// computed by exactly the operations of det_logaddexp_t, so the bits are those of two separate calls.
__device__ __forceinline__ void det_logaddexp_pair_u(double x1, double y1, double x2, double y2, double& r1, double& r2) {
    const double d1 = __builtin_fabs(x1 - y1), d2 = __builtin_fabs(x2 - y2);
    if (__builtin_expect((d1 < 16.0) & (d2 < 16.0), 1)) {
        const int i1 = dm_u::idx((int)(d1 * 16.0)), i2 = dm_u::idx((int)(d2 * 16.0));
        double c1[8], c2[8];
        dm_u::row8(DM_SOFTPLUS_TBL[i1], c1);
        dm_u::row8(DM_SOFTPLUS_TBL[i2], c2);
        const double t1 = d1 - (double)(2 * i1 + 1) * 0.03125, t2 = d2 - (double)(2 * i2 + 1) * 0.03125;
        const double m1 = dm_u::max_nonnan(x1, y1), m2 = dm_u::max_nonnan(x2, y2);
        const double s1 = dm_u::template horner_row<8>(t1, c1), s2 = dm_u::template horner_row<8>(t2, c2);
        r1 = m1 + s1;
        r2 = m2 + s2;
    } else {
        r1 = det_logaddexp_t<dm_u>(x1, y1);
        r2 = det_logaddexp_t<dm_u>(x2, y2);
    }
}
// per-lane arguments
__device__ __forceinline__ double det_exp_v(double x) { return det_exp_t<dm_v>(x); }
__device__ __forceinline__ double det_log_v(double x) { return det_log_t<dm_v>(x); }
__device__ __forceinline__ double det_randexp_v(uint64_t r) { return det_randexp_t<dm_rexp>(r); }
__device__ __forceinline__ void det_randn2_v(uint64_t r1, uint64_t r2, double* z0, double* z1) { det_randn2_t<dm_v>(r1, r2, z0, z1); }

// ------------------------------------------------------------------------------------------------------------------------------
// The Bernoulli-logit link of NE observations at once (the logistic round engine, logistic_rounds.hpp): sig = det_logistic_sigma(η),
// l1pe = det_log1pexp(η) (include/dhmc_detmath.h), value by value the operations of det_exp_t<dm_v>(-|η|), det_log1p_nonneg_t<dm_v>(t)
// and the two divisions by w = 1 + t — but
//   * in PHASES over the NE arguments: all reductions, then all table gathers in flight together (one round trip per table instead
//     of one per argument), then the polynomials;
//   * with the functions' special cases taken out of the common path by ONE ballot: an argument with |η| > 707 or NaN anywhere in the
//     wave (e^{-|η|} subnormal or zero) sends the whole wave through the functions themselves; for every other argument t is a normal
//     number in [2^-1020, 1], w = 1 + t is in [1, 2] and none of det_exp_t's / det_log1p_nonneg_t's selects can fire except w == 1;
//   * with both quotients n / w of an argument — (η >= 0 ? 1 : t) / w and (t - (w - 1)) / w — taken from ONE refined reciprocal of w:
//     a correctly rounded fp64 division on this hardware IS  r₀ = v_rcp_f64(w), two Newton steps r ← fma(r, fma(-w, r, 1), r),
//     q₀ = n·r, q = fma(fma(-w, q₀, n), r, q₀)  wrapped in v_div_scale / v_div_fmas / v_div_fixup, which pre-scale operands whose exponents
//     are extreme and patch Inf / NaN / 0 — and are the identity for w in [1, 2] and a numerator that is 0 or of magnitude >= 2^-969.
//     Smaller numerators exist only for t < 2^-53, where w == 1 exactly, r == 1 exactly and q = n exactly.  So the explicit sequence
//     returns the bits of the IEEE quotient (the oracle's `/`), the reciprocal's five instructions are spent once, and the compiler
//     can interleave the chains of the NE arguments (v_div_fmas reads VCC: its expansions run one after the other).
//     Checked value by value against the CPU side: dhmc_detmath_selftest kinds 10 / 11 (tests/test_gpu_detmath.py).
// ------------------------------------------------------------------------------------------------------------------------------
struct dm_shared_recip {
    double w, r;
    __device__ __forceinline__ explicit dm_shared_recip(double w_) : w(w_) {
        const double r0 = __builtin_amdgcn_rcp(w_);
        const double r1 = __builtin_fma(r0, __builtin_fma(-w_, r0, 1.0), r0);
        r = __builtin_fma(r1, __builtin_fma(-w_, r1, 1.0), r1);
    }
    __device__ __forceinline__ double quotient(double n) const {        // n / w, correctly rounded (see above for the operand ranges)
        const double q0 = n * r;
        return __builtin_fma(__builtin_fma(-w, q0, n), r, q0);
    }
};

template <int NE>
__device__ __forceinline__ void logistic_link_batch(const double (&eta)[NE], double (&sig)[NE], double (&l1pe)[NE]) {
    double x[NE], r[NE], t[NE];
    int j[NE], e[NE];
    bool rare = false;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        x[i] = -__builtin_fabs(eta[i]);
        dm_exp_reduce(x[i], &j[i], &e[i], &r[i]);
        rare = rare || !(x[i] >= -707.0);                  // NaN, or e^x below 2^-1020 (x >= -707 has e >= -1020: no subnormal path)
    }
    if (__builtin_expect(__ballot(rare) != 0ull, 0)) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            sig[i] = det_logistic_sigma_t<dm_v>(eta[i]);
            l1pe[i] = det_log1pexp_t<dm_v>(eta[i]);
        }
        return;
    }
    double T[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) T[i] = DM_EXP2_TBL[j[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const uint64_t yb = dm_bits(dm_exp_poly<dm_v>(r[i], T[i]));
        // 2^e by an exponent-field add (det_exp_t's e >= -1021 branch): (e << 52) has a zero low word, so only the high word moves
        const uint32_t hi = (uint32_t)(yb >> 32) + ((uint32_t)e[i] << 20);
        t[i] = dm_from_bits(((uint64_t)hi << 32) | (uint32_t)yb);
    }
    double w[NE], m[NE], row[NE][3];
    int jl[NE], el[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        w[i] = 1.0 + t[i];
        dm_log_reduce(dm_bits(w[i]), 0, &jl[i], &el[i], &m[i]);                       // w in [1, 2]
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const double* __restrict__ lp = DM_LOG_TBL[jl[i]];
        row[i][0] = lp[0]; row[i][1] = lp[1]; row[i][2] = lp[2];
    }
    // the general values for EVERY argument first (opaque to the compiler below: written as one expression per argument it would put each
    // argument's logarithm and quotients under that argument's own exec mask, one chain after the other), then the selects
    double lg[NE], sg[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const dm_shared_recip rw(w[i]);
        lg[i] = dm_log_finish<dm_v>(m[i], el[i], row[i][0], row[i][1], row[i][2]) + rw.quotient(t[i] - (w[i] - 1.0));
        sg[i] = rw.quotient(eta[i] >= 0 ? 1.0 : t[i]);
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) asm volatile("" : "+v"(lg[i]));
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const double l1p = (w[i] == 1.0) ? t[i] : lg[i];       // det_log1p_nonneg_t's early return (w is finite here)
        sig[i] = sg[i];
        l1pe[i] = (eta[i] > 0 ? eta[i] : 0.0) + l1p;
    }
}

}  // namespace dhmc
