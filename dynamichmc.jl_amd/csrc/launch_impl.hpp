// Definitions behind launch.hpp: included only by family.hip, which instantiates dispatch_family<T> for ONE
// target family per translation unit.
#pragma once
#include "dense_rounds_k3b.hpp"
#include "launch.hpp"
#include "nuts_pipeline_kernel.hpp"
#include "packed_kernels.hpp"

namespace dhmc {

// Level-1 summaries go to LDS when four Dpad-rows per wave still leave one wave per SIMD
// (4 waves per CU) resident: always for Dpad <= 512, and at Dpad = 1024 (33.5 KB per wave).
template <class T, int NPL>
void launch_run(const RunParams& P, hipStream_t s) {
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)nuts_run_kernel<T, NPL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    if (P.l1_in_lds)
        hipLaunchKernelGGL((nuts_run_kernel<T, NPL, true>), dim3(P.C), dim3(WAVE), lds_bytes(P.Dpad, true, lds_extra_levels(NPL)), s, P);
    else
        hipLaunchKernelGGL((nuts_run_kernel<T, NPL, false>), dim3(P.C), dim3(WAVE), lds_bytes(P.Dpad, false, 0), s, P);
}
template <class T, int NPL>
void launch_init(const InitParams& P, hipStream_t s) {
    hipLaunchKernelGGL((init_kernel<T, NPL>), dim3(P.C), dim3(WAVE), 0, s, P);
}
template <class T, int NPL>
void launch_search(const SearchParams& P, hipStream_t s) {
    hipLaunchKernelGGL((stepsize_search_kernel<T, NPL>), dim3(P.C), dim3(WAVE), sizeof(double) * P.Dpad, s, P);
}

template <class T, int NPL>
void launch_run_dense(const RunParams& P, const DenseMetric& M, hipStream_t s) {
    hipLaunchKernelGGL((nuts_run_dense_kernel<T, NPL>), dim3(P.C), dim3(WAVE), lds_bytes_dense(), s, P, M);
}
template <class T, int NPL>
void launch_search_dense(const SearchParams& P, const DenseMetric& M, hipStream_t s) {
    hipLaunchKernelGGL((stepsize_search_dense_kernel<T, NPL>), dim3(P.C), dim3(WAVE), 0, s, P, M);
}

template <class T, int NPL>
void launch_round_op(int which, const RoundArgs& a, hipStream_t s) {
    switch (which) {
    case 0: hipLaunchKernelGGL((rounds_start_kernel<NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R); break;
    case 1: hipLaunchKernelGGL((rounds_k0_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R); break;
    case 2: hipLaunchKernelGGL((rounds_k2_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R); break;
    default:
        // chains of 512+ coordinates: a workgroup (4+ waves) per chain; DHMC_DENSE=k3_block=0 (read at dhmc_create) keeps the one-wave kernel
        if constexpr (NPL >= 8) {
            if (a.P.k3_block || NPL > 16) {
                hipLaunchKernelGGL((rounds_k3b_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE * k3b_waves(NPL)), 0, s, a.P, a.R);
                break;
            }
        }
        if constexpr (NPL <= 16) hipLaunchKernelGGL((rounds_k3_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R);
        break;
    }
}

// 32 / 64 slots per lane (D <= 4096): only the streaming round-engine kernels exist at these widths
template <class T, int NPL>
int dispatch_op_big(Op op, const void* P, hipStream_t s) {
    const RoundArgs& a = *(const RoundArgs*)P;
    switch (op) {
    case Op::RoundStart: hipLaunchKernelGGL((rounds_start_kernel<NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R); return DHMC_OK;
    case Op::RoundK0: hipLaunchKernelGGL((rounds_k0_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE), 0, s, a.P, a.R); return DHMC_OK;
    case Op::RoundK3: hipLaunchKernelGGL((rounds_k3b_kernel<T, NPL>), dim3(a.P.C), dim3(WAVE * k3b_waves(NPL)), 0, s, a.P, a.R); return DHMC_OK;
    default: return DHMC_ERR_UNSUPPORTED;
    }
}

template <class T, int NPL>
void dispatch_op(Op op, const void* P, hipStream_t s, const DenseMetric* M) {
    if (op == Op::RoundStart || op == Op::RoundK0 || op == Op::RoundK2 || op == Op::RoundK3) {
        launch_round_op<T, NPL>((int)op - (int)Op::RoundStart, *(const RoundArgs*)P, s);
        return;
    }
    if (op == Op::ProbeTrajectory || op == Op::ProbeRatios) {
        const ProbeParams& Q = *(const ProbeParams*)P;
        const dim3 g(Q.C), b(WAVE);
        if (M) {
            if (op == Op::ProbeTrajectory) hipLaunchKernelGGL((probe_kernel<T, NPL, true, 0>), g, b, 0, s, Q, *M);
            else hipLaunchKernelGGL((probe_kernel<T, NPL, true, 1>), g, b, 0, s, Q, *M);
        } else {
            const size_t lds = sizeof(double) * Q.Dpad;
            if (op == Op::ProbeTrajectory) hipLaunchKernelGGL((probe_kernel<T, NPL, false, 0>), g, b, lds, s, Q, DenseMetric{});
            else hipLaunchKernelGGL((probe_kernel<T, NPL, false, 1>), g, b, lds, s, Q, DenseMetric{});
        }
        return;
    }
    if (M && op == Op::Run) { launch_run_dense<T, NPL>(*(const RunParams*)P, *M, s); return; }
    if (M && op == Op::Search) { launch_search_dense<T, NPL>(*(const SearchParams*)P, *M, s); return; }
    switch (op) {
    case Op::Run: launch_run<T, NPL>(*(const RunParams*)P, s); break;
    case Op::Init: launch_init<T, NPL>(*(const InitParams*)P, s); break;
    case Op::Search: launch_search<T, NPL>(*(const SearchParams*)P, s); break;
    default: break;
    }
}
template <class T>
int dispatch_family(int npl, Op op, const void* P, hipStream_t s, const DenseMetric* M) {
    if (op == Op::RunPacked) return launch_run_packed<T>(*(const RunParams*)P, s);   // several chains per wave (packed_kernels.hpp)
    if (op == Op::RunPipeline) return launch_run_pipeline<T>(*(const RunParams*)P, s);   // integrator ‖ turn statistics ‖ visited statistic ‖ proposals (nuts_pipeline_kernel.hpp)
    switch (npl) {
    case 1: dispatch_op<T, 1>(op, P, s, M); return DHMC_OK;
    case 2: dispatch_op<T, 2>(op, P, s, M); return DHMC_OK;
    case 4: dispatch_op<T, 4>(op, P, s, M); return DHMC_OK;
    case 8: dispatch_op<T, 8>(op, P, s, M); return DHMC_OK;
    case 16: dispatch_op<T, 16>(op, P, s, M); return DHMC_OK;
    case 32: if constexpr (T::kBigDims) return dispatch_op_big<T, 32>(op, P, s); else return DHMC_ERR_UNSUPPORTED;
    case 64: if constexpr (T::kBigDims) return dispatch_op_big<T, 64>(op, P, s); else return DHMC_ERR_UNSUPPORTED;
    default: return DHMC_ERR_UNSUPPORTED;
    }
}

}  // namespace dhmc
