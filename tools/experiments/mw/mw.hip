// The multi-wave per-draw kernels (nuts_mw_kernel.hpp), one instantiation per coordinate-wise target family and
// chain width (NW = 2: 512 coordinates, NW = 4: 1024).  Own translation unit: see launch.hpp.
#include "mw_launch.hpp"
#include "nuts_mw_kernel.hpp"

namespace dhmc {

template <class T, int NW>
void launch_run_mw(const RunParams& P, hipStream_t s) {
    static bool once = [] {   // 53 KB of dynamic LDS per chain
        (void)hipFuncSetAttribute((const void*)nuts_run_mw_kernel<T, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((nuts_run_mw_kernel<T, NW>), dim3(P.C), dim3(WAVE * (NW + 1)), mw_lds_bytes(P.Dpad, NW), s, P);
}

template void launch_run_mw<StdNormalT, 2>(const RunParams&, hipStream_t);
template void launch_run_mw<StdNormalT, 4>(const RunParams&, hipStream_t);
template void launch_run_mw<DiagNormalT, 2>(const RunParams&, hipStream_t);
template void launch_run_mw<DiagNormalT, 4>(const RunParams&, hipStream_t);

}  // namespace dhmc

#ifdef MW_TRACE
extern "C" int dhmc_debug_mw_trace(unsigned long long* out, unsigned int* counts, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(dhmc::g_mw_trace), sizeof(unsigned long long) * 8 * 4096) != hipSuccess) return 1;
    if (counts && hipMemcpyFromSymbol(counts, HIP_SYMBOL(dhmc::g_mw_trace_n), sizeof(unsigned int) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned int z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dhmc::g_mw_trace_n), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

