// Device policies of the ABI's scalar math (include/dhmc_detmath.h).  One chain per wavefront means every scalar operation
// of the tree logic costs a full vector issue slot, and a lone wave per SIMD is bound by the NUMBER of instructions it issues
//  — so these policies keep every IEEE operation and its order (same bits as the oracle's dm_generic
// instantiation, checked by tests/test_gpu_detmath.py) and change only where operands live:
//
//   dm_uniform   the arguments are wave-uniform (logaddexp pairs of a merge, acceptance rate, step size, the funnel's exp(-v)):
//                a table index goes through v_readfirstlane, so the row arrives by ONE scalar load (s_load_dwordx16 for the
//                eight coefficients of a softplus cell: no vector registers, no vector-memory latency) and the Horner chain
//                runs with its coefficients in scalar registers;
//   dm_vector    the arguments differ per lane (log and sin/cos of the momentum refresh, Exp(1) draws): table rows are
//                gathered per lane, compile-time polynomial coefficients sit in scalar registers.
//
// Why the Horner chains are inline assembly: for fma(s, x, constant) the compiler's own choice is two v_mov_b32 of the
// constant into a vector register pair followed by v_fmac_f64 — three vector instructions per step instead of one — and
// around single-instruction asm statements it inserts a wait state per step; a whole chain per asm block has neither.
// v_fma_f64 reads at most one scalar operand (constant-bus limit of gfx9), so the leading coefficient is moved to the
// accumulator first (v_mov_b64).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dhmc_detmath.h"

namespace dhmc {

struct dm_asm_horner {
    template <int N>
    static __device__ __forceinline__ double run(double x, const double (&c)[N]) {
        static_assert(N == 4 || N == 5 || N == 6 || N == 8, "Horner chains of the ABI's polynomials");
        double s;
        if constexpr (N == 4) {
            asm("v_mov_b64 %0, %5\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %2"
                : "=&v"(s) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]));
        } else if constexpr (N == 5) {
            asm("v_mov_b64 %0, %6\n\tv_fma_f64 %0, %0, %1, %5\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %3\n\t"
                "v_fma_f64 %0, %0, %1, %2"
                : "=&v"(s) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]));
        } else if constexpr (N == 6) {
            asm("v_mov_b64 %0, %7\n\tv_fma_f64 %0, %0, %1, %6\n\tv_fma_f64 %0, %0, %1, %5\n\tv_fma_f64 %0, %0, %1, %4\n\t"
                "v_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %2"
                : "=&v"(s) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]));
        } else {
            asm("v_mov_b64 %0, %9\n\tv_fma_f64 %0, %0, %1, %8\n\tv_fma_f64 %0, %0, %1, %7\n\tv_fma_f64 %0, %0, %1, %6\n\t"
                "v_fma_f64 %0, %0, %1, %5\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %2"
                : "=&v"(s) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6]), "s"(c[7]));
        }
        return s;
    }
};

struct dm_uniform {
    static __device__ __forceinline__ int idx(int i) {      // opaque to the optimiser: everything derived from the index is scalar code
        int s;
        asm("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(i));
        return s;
    }
    template <int N>
    static __device__ __forceinline__ double horner_const(double x, const double (&c)[N]) { return dm_asm_horner::run<N>(x, c); }
    template <int N>
    static __device__ __forceinline__ double horner_row(double x, const double (&c)[N]) { return dm_asm_horner::run<N>(x, c); }
    static __device__ __forceinline__ double max_nonnan(double x, double y) {       // v_max_f64: one instruction where a select is three
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        return r;
    }
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
    static __device__ __forceinline__ double horner_const(double x, const double (&c)[N]) { return dm_asm_horner::run<N>(x, c); }
    template <int N>
    static __device__ __forceinline__ double horner_row(double x, const double (&c)[N]) { return dm_generic::horner_const<N>(x, c); }
    static __device__ __forceinline__ double max_nonnan(double x, double y) { return dm_uniform::max_nonnan(x, y); }
    static __device__ __forceinline__ void row8(const double* row, double (&c)[8]) { dm_generic::row8(row, c); }
};

// wave-uniform arguments
__device__ __forceinline__ double det_exp_u(double x) { return det_exp_t<dm_uniform>(x); }
__device__ __forceinline__ double det_log_u(double x) { return det_log_t<dm_uniform>(x); }
__device__ __forceinline__ double det_logaddexp_u(double x, double y) { return det_logaddexp_t<dm_uniform>(x, y); }
__device__ __forceinline__ double det_pow_pos_u(double x, double y) { return det_pow_pos_t<dm_uniform>(x, y); }
// per-lane arguments
__device__ __forceinline__ double det_exp_v(double x) { return det_exp_t<dm_vector>(x); }
__device__ __forceinline__ double det_log_v(double x) { return det_log_t<dm_vector>(x); }
__device__ __forceinline__ double det_randexp_v(uint64_t r) { return det_randexp_t<dm_vector>(r); }
__device__ __forceinline__ void det_randn2_v(uint64_t r1, uint64_t r2, double* z0, double* z1) { det_randn2_t<dm_vector>(r1, r2, z0, z1); }

}  // namespace dhmc
