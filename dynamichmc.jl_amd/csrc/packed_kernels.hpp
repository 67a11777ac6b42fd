// Device side of the packed small-D engine (packed_core.hpp): the group's cross-lane operations on gfx950 and the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "detmath_dev.hpp"
#include "packed_core.hpp"
#include "nuts_kernels.hpp"
#include "targets.hpp"
#include "wave.hpp"

namespace dhmc {

// DHMC_TARGET_* of a functor of targets.hpp that has a packed evaluator (-1: none)
template <class T> struct PackedId { static constexpr int value = -1; };
template <> struct PackedId<StdNormalT> { static constexpr int value = DHMC_TARGET_STD_NORMAL; };
template <> struct PackedId<DiagNormalT> { static constexpr int value = DHMC_TARGET_DIAG_NORMAL; };
template <> struct PackedId<TridiagNormalT> { static constexpr int value = DHMC_TARGET_TRIDIAG_NORMAL; };
template <> struct PackedId<DenseNormalT> { static constexpr int value = DHMC_TARGET_DENSE_NORMAL; };
template <> struct PackedId<FunnelT> { static constexpr int value = DHMC_TARGET_FUNNEL; };
template <> struct PackedId<AlwaysDivergentT> { static constexpr int value = DHMC_TARGET_ALWAYS_DIVERGENT; };

// A chain's group: L adjacent lanes, aligned to L.  Every operation is called with all lanes of the group active (the group's
// lanes hold identical control state), possibly under a divergent exec mask of the wave.
//   sum      the top log2(L) levels of the ABI's summation tree: xor butterfly 1, 2, 4, 8 inside the group by DPP (quad_perm,
//            quad_perm, row_half_mirror, row_mirror — equal to the xor pairing because the lanes of an already reduced subgroup
//            hold identical values); every lane of the group gets the total
//   first    the value of the group's lane 0, bit for bit (a DPP broadcast, not a sum: -0 stays -0)
//   pick     the value of the group's lane `src` (run-time index: ds_bpermute)
//   all      a predicate over the group's lanes (ballot of the wave's active lanes, masked to the group)
template <int L>
struct PackedGroup {
    template <int N>
    static __device__ __forceinline__ void sum_n(double (&v)[N]) {
        if constexpr (L >= 2) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0xB1>(v[i]);   // quad_perm [1,0,3,2]
        }
        if constexpr (L >= 4) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x4E>(v[i]);   // quad_perm [2,3,0,1]
        }
        if constexpr (L >= 8) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x141>(v[i]);  // row_half_mirror
        }
        if constexpr (L >= 16) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x140>(v[i]);  // row_mirror
        }
    }
    static __device__ __forceinline__ double sum(double x) {
        double v[1] = {x};
        sum_n<1>(v);
        return v[0];
    }
    static __device__ __forceinline__ int first_i(int x) { return __shfl(x, (int)(threadIdx.x & 63u & ~(unsigned)(L - 1))); }
    static __device__ __forceinline__ double first(double x) {
        if constexpr (L == 1) return x;
        else if constexpr (L == 2) return dpp_f64<0xA0>(x);                   // quad_perm [0,0,2,2]
        else {
            const int sub = (int)(threadIdx.x & (L - 1));
            double b = dpp_f64<0x00>(x);                                      // quad_perm [0,0,0,0]: every quad's lane 0
            if constexpr (L >= 8) {
                const double hm = dpp_f64<0x141>(b);                          // lanes 4..7 of a half row read lanes 3..0
                b = (sub & 4) ? hm : b;
            }
            if constexpr (L >= 16) {
                const double mr = dpp_f64<0x140>(b);                          // lanes 8..15 of a row read lanes 7..0
                b = (sub & 8) ? mr : b;
            }
            return b;
        }
    }
    // the value of the group's lane before / after this one (row_shr:1 / row_shl:1: a group never crosses a row of 16 lanes); the
    // group's first / last lane gets a lane of the neighbouring group, or 0 at the row's edge — its caller does not use it
    static __device__ __forceinline__ double prev(double x) {
        if constexpr (L == 1) return 0.0;
        else {
            const uint64_t b = (uint64_t)__double_as_longlong(x);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x111, 0xF, 0xF, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x111, 0xF, 0xF, true);
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        }
    }
    static __device__ __forceinline__ double next(double x) {
        if constexpr (L == 1) return 0.0;
        else {
            const uint64_t b = (uint64_t)__double_as_longlong(x);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x101, 0xF, 0xF, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x101, 0xF, 0xF, true);
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        }
    }
    static __device__ __forceinline__ double pick(double x, int src) {
        if constexpr (L == 1) return x;
        else return __shfl(x, (int)((threadIdx.x & ~(unsigned)(L - 1)) + (unsigned)src));
    }
    static __device__ __forceinline__ bool all(bool pred) {
        if constexpr (L == 1) return pred;
        else {
            const unsigned long long b = __builtin_amdgcn_ballot_w64(pred);
            const unsigned long long gm = ((L >= 64 ? 0ull : (1ull << L)) - 1ull) << (threadIdx.x & ~(unsigned)(L - 1));
            return (b & gm) == gm;
        }
    }
    static __device__ __forceinline__ bool wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }
};

// Phase timing (tools/experiments/phase_timing.sh, -DDHMC_PHASE_TIMING builds only): the clocks a wave spends in each region of
// the packed body, added to g_phase (nuts_kernels.hpp) — [14] counts trips of the main loop, [15] waves.
#ifdef DHMC_PHASE_TIMING
#define PK_PH_DECL unsigned long long ph_a0 = 0, ph_a1 = 0, ph_a2 = 0, ph_a3 = 0, ph_a4 = 0, ph_a5 = 0, ph_a6 = 0, ph_a7 = 0, ph_a8 = 0; unsigned long long ph_t0 = __builtin_readcyclecounter();
#define PK_PH_END(i) { const unsigned long long t_ = __builtin_readcyclecounter(); ph_a##i += t_ - ph_t0; ph_t0 = t_; }
#define PK_PH_FLUSH(trips) { PK_PH_END(0) if (threadIdx.x == 0) { const unsigned long long a_[9] = {ph_a0, ph_a1, ph_a2, ph_a3, ph_a4, ph_a5, ph_a6, ph_a7, ph_a8}; for (int i_ = 0; i_ < 9; ++i_) atomicAdd(&g_phase[i_], a_[i_]); atomicAdd(&g_phase[14], (unsigned long long)(trips)); atomicAdd(&g_phase[15], 1ull); } }
#elif defined(DHMC_PHASE_MARK)   // region boundaries as comments in the assembly (static instruction counts per region)
#define PK_PH_DECL
#define PK_PH_END(i) asm volatile("; PKPH " #i);
#define PK_PH_FLUSH(trips) asm volatile("; PKPH 0");
#else
#define PK_PH_DECL
#define PK_PH_END(i)
#define PK_PH_FLUSH(trips)
#endif

// A caller's device functor (or any functor of targets.hpp) as a packed evaluator, for the functors whose ℓ is a sum of per-coordinate
// terms (kElementwise && kDeferred: "eval takes any element base as its lane argument", targets.hpp): the functor is called once per
// coordinate with ONE slot — its partial sum is then exactly the leaf of the ABI's summation tree that the packed engine adds up
// (lane-local adjacent pairs, then the group's DPP butterfly), and `finish` makes ℓ of the total.  Same bits as the functor through
// the wave-per-chain kernel at one slot per lane.
template <class T>
struct PackedFunctor {
    static constexpr bool kFiniteLqImpliesFiniteQ = T::kFiniteLqImpliesFiniteQ;
    static constexpr bool kEligible = T::kElementwise && T::kDeferred && T::kRecomputeGrad && T::kFiniteLqImpliesFiniteGrad && !T::kBigDims;
    T f;
    __device__ explicit PackedFunctor(const TargetParams& p) : f(p) {}
    template <int CPL, class Grp, class Pol>
    __device__ __forceinline__ double eval(const double (&q)[CPL], double (&g)[CPL], int e0, int D) const {
        double t[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const double q1[1] = {q[k]};
            double g1[1];
            t[k] = f.template eval<1>(q1, g1, e0 + k, D);
            g[k] = g1[0];
        }
        return f.finish(Grp::sum(pk::Tree<CPL>::sum(t)));
    }
};

// One wavefront per workgroup, 64 / L chains in it: group `grp` of workgroup b runs the chain in place b·(64/L) + grp of the launch
// order (RunParams::launch_order: the chains sorted by the previous launch's work, longest first — so that chains with
// persistently deep trees share waves instead of each holding a wave of finished chains open).
// (two coordinates per lane: TWO waves per SIMD — 256 registers each; the launch plans its LDS and its queue for that, capi_run.hip
// plan_packed_launch — four per lane: one)
template <class PT, int L, int CPL>
__global__ __launch_bounds__(64, CPL == 2 ? 2 : 1) void nuts_run_packed_kernel(RunParams P) {
    constexpr int GPW = 64 / L;
    const int sub = (int)(threadIdx.x & (L - 1));
    const int grp = (int)(threadIdx.x / L);
    const int place = P.pk_order_base + (int)blockIdx.x * GPW + grp;
    extern __shared__ double pk_lds[];
    double* const lds_cold = pk_lds;
    double* const lds_rows = pk_lds + (size_t)6 * 64 * CPL;
    double* const lds_sc = lds_rows + (size_t)P.pk_lds_levels * 4 * 64 * CPL;
    typedef PackedGroup<L> Grp;
    typedef dm_vector Pol;
#define PK_ATOMIC_ADD_ULL(ptr, v) atomicAdd((ptr), (v))
#define PK_QUEUE_NEXT(ptr) atomicAdd((ptr), 1u)
#define PK_LOAD_UINT(ptr) __atomic_load_n((ptr), __ATOMIC_RELAXED)
#define PK_ATOMIC_DEC_UINT(ptr) atomicSub((ptr), 1u)
#include "packed_body.inc"
#undef PK_ATOMIC_DEC_UINT
#undef PK_LOAD_UINT
#undef PK_QUEUE_NEXT
#undef PK_ATOMIC_ADD_ULL
}

#ifndef __HIPCC_RTC__      // (the host side; a caller's functor is compiled with hiprtc: kernels only)
template <class T>
int launch_run_packed(const RunParams& P, hipStream_t s) {
    constexpr int TGT = PackedId<T>::value;
    if constexpr (TGT < 0) {
        return DHMC_ERR_UNSUPPORTED;
    } else {
        typedef pk::PackedTarget<TGT> PT;
        RunParams Q;
        int L = 0;
        unsigned waves = 0;
        size_t lds = 0;
        if (int rc = packed_launch_prepare(P, s, &Q, &L, &waves, &lds)) return rc;
        const int cpl = P.pk_cpl;
        const dim3 grid(waves), block(64);
#define DHMC_PK_LAUNCH(LL, CC)                                                                                                 \
    if (L == LL && cpl == CC) {                                                                                                \
        static bool once = [] {                                                                                                \
            (void)hipFuncSetAttribute((const void*)nuts_run_packed_kernel<PT, LL, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk::kMaxLdsPerWave); \
            return true;                                                                                                       \
        }();                                                                                                                   \
        (void)once;                                                                                                            \
        hipLaunchKernelGGL((nuts_run_packed_kernel<PT, LL, CC>), grid, block, lds, s, Q);                                      \
        return DHMC_OK;                                                                                                        \
    }
        DHMC_PK_LAUNCH(1, 2) DHMC_PK_LAUNCH(2, 2) DHMC_PK_LAUNCH(4, 2) DHMC_PK_LAUNCH(8, 2) DHMC_PK_LAUNCH(16, 2)
        DHMC_PK_LAUNCH(1, 4) DHMC_PK_LAUNCH(2, 4) DHMC_PK_LAUNCH(4, 4) DHMC_PK_LAUNCH(8, 4) DHMC_PK_LAUNCH(16, 4)
#undef DHMC_PK_LAUNCH
        return DHMC_ERR_UNSUPPORTED;
    }
}
#endif

}  // namespace dhmc
