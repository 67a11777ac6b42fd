// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar math used by the oracle, in two interchangeable flavours:
//   DET  — the deterministic functions of include/dhmc_detmath.h (the ABI's numerical
//          contract); oracle and HIP kernels then agree bit for bit.
//   LIBM — glibc's exp/log/log1p/sincos/pow, i.e. the class of functions the reference itself
//          calls through Julia; used to show that DET stays inside the reference's tolerance.
// Also the wave-ordered reductions: the ABI fixes the summation order of every dot product
// (256-coordinate blocks of 64 interleaved fma chains, combined per lane by an adjacent-pairs
// tree over the blocks and then over the lanes), because LinearAlgebra.dot's order (src/hamiltonian.jl:103, src/NUTS.jl:130) is
// BLAS-dependent and unpinned.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
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

// Σ_e a[e]*b[e] in the ABI's order (include/dhmc.h "Summation order"; device: csrc/wave.hpp LaneAcc +
// wave_allreduce).  The row is cut into blocks of 256 coordinates; in block B lane l (of 64) accumulates
// e = 256 B + l, + 64, + 128, + 192 with fma; the blocks' partial sums are combined per lane by an adjacent-pairs
// binary tree over the power-of-two number of blocks of the padded row (missing blocks are +0.0); the 64 lane
// values are combined by the xor-butterfly 1,2,4,8,16,32 (adjacent pairs first).  For n <= 256 this is one fma chain
// per lane followed by the butterfly.
// A second flavour of every sum, for tolerance tests only (tests/test_gpu_tolerance.py): plain left-to-right accumulation
// with separately rounded products and no blocks — the textbook loop, the order closest to what a generic
// `LinearAlgebra.dot` / `*` does on the reference's side (src/hamiltonian.jl:103,110; src/NUTS.jl:130), whose BLAS order
// is unpinned.  Process-wide switch (oracle_set_sequential_sums); the ABI's order is the default and the only one the
// device implements.
inline bool& sequential_sums() {
    static bool on = false;
    return on;
}
inline double wave_tree(double* partial) {
    for (int off = 1; off < 64; off <<= 1)
        for (int l = 0; l < 64; l += 2 * off) partial[l] = partial[l] + partial[l + off];
    return partial[0];
}
inline double wave_dot(const double* a, const double* b, int n) {
    if (sequential_sums()) {
        double s = 0.0;
        for (int e = 0; e < n; ++e) s = s + a[e] * b[e];
        return s;
    }
    int nblk = 1;
    while (256 * nblk < n) nblk *= 2;
    std::vector<double> blk((size_t)nblk * 64, 0.0);
    for (int B = 0; B < nblk; ++B)
        for (int l = 0; l < 64; ++l) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) {
                const int e = 256 * B + 64 * k + l;
                if (e < n) acc = __builtin_fma(a[e], b[e], acc);
            }
            blk[(size_t)B * 64 + l] = acc;
        }
    for (int w = 1; w < nblk; w *= 2)
        for (int B = 0; B + w < nblk; B += 2 * w)
            for (int l = 0; l < 64; ++l) blk[(size_t)B * 64 + l] = blk[(size_t)B * 64 + l] + blk[(size_t)(B + w) * 64 + l];
    return wave_tree(blk.data());
}

}  // namespace oracle
