// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar math used by the oracle, in two interchangeable flavours:
//   DET  — the deterministic functions of include/dhmc_detmath.h (the ABI's numerical
//          contract); oracle and HIP kernels then agree bit for bit.
//   LIBM — glibc's exp/log/log1p/sincos/pow, i.e. the class of functions the reference itself
//          calls through Julia; used to show that DET stays inside the reference's tolerance.
// Also the wave-ordered reductions: the ABI fixes the summation order of every dot product
// (64 interleaved partial sums accumulated with fma, combined by an adjacent-pairs binary
// tree), because LinearAlgebra.dot's order (src/hamiltonian.jl:103, src/NUTS.jl:130) is
// BLAS-dependent and unpinned.
#pragma once
#include <cmath>
#include <cstdint>
#include "../include/dhmc_detmath.h"

namespace oracle {

struct MathOps {
    bool det = true;
    double exp(double x) const { return det ? dhmc::det_exp(x) : std::exp(x); }
    double log(double x) const { return det ? dhmc::det_log(x) : std::log(x); }
    double pow_pos(double x, double y) const {
        return det ? dhmc::det_pow_pos(x, y) : std::pow(x, y);
    }
    // LogExpFunctions.logaddexp (call sites src/trees.jl:145, src/NUTS.jl:70)
    double logaddexp(double x, double y) const {
        if (det) return dhmc::det_logaddexp(x, y);
        double d = (x == y) ? 0.0 : std::fabs(x - y);
        return std::fmax(x, y) + std::log1p(std::exp(-d));
    }
    double randexp(uint64_t r) const {
        return det ? dhmc::det_randexp(r) : -std::log(dhmc::u01_open_closed(r));
    }
    void randn2(uint64_t r1, uint64_t r2, double* z0, double* z1) const {
        if (det) {
            dhmc::det_randn2(r1, r2, z0, z1);
            return;
        }
        double u1 = dhmc::u01_open_closed(r1), u2 = dhmc::u01_closed_open(r2);
        double rad = std::sqrt(-2.0 * std::log(u1));
        const double TWO_PI = 6.283185307179586476925286766559;
        *z0 = rad * std::cos(TWO_PI * u2);
        *z1 = rad * std::sin(TWO_PI * u2);
    }
};

// Σ_e a[e]*b[e] in wave order: lane l (of 64) accumulates e = l, l+64, ... with fma, then the
// 64 partials are combined by the xor-butterfly 1,2,4,8,16,32 (adjacent pairs first).
inline double wave_tree(double* partial) {
    for (int off = 1; off < 64; off <<= 1)
        for (int l = 0; l < 64; l += 2 * off) partial[l] = partial[l] + partial[l + off];
    return partial[0];
}
inline double wave_dot(const double* a, const double* b, int n) {
    double partial[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int e = l; e < n; e += 64) acc = __builtin_fma(a[e], b[e], acc);
        partial[l] = acc;
    }
    return wave_tree(partial);
}

}  // namespace oracle
