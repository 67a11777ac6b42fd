// The multi-wave per-draw kernels (nuts_mw_kernel.hpp), one instantiation per coordinate-wise target family and
// chain width (NW = 2: 512 coordinates, NW = 4: 1024).  Own translation unit: see launch.hpp.
#include "mw_launch.hpp"
#include "nuts_mw_kernel.hpp"

namespace dhmc {

template <class T, int NW>
void launch_run_mw(const RunParams& P, hipStream_t s) {
    static bool once = [] {   // up to 55 KB of dynamic LDS per chain (mw_lds_bytes at max_depth = 32)
        (void)hipFuncSetAttribute((const void*)nuts_run_mw_kernel<T, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((nuts_run_mw_kernel<T, NW>), dim3(P.C), dim3(WAVE * (NW + 1)), mw_lds_bytes(P.Dpad, NW, P.max_depth), s, P);
}

template void launch_run_mw<StdNormalT, 2>(const RunParams&, hipStream_t);
template void launch_run_mw<StdNormalT, 4>(const RunParams&, hipStream_t);
template void launch_run_mw<DiagNormalT, 2>(const RunParams&, hipStream_t);
template void launch_run_mw<DiagNormalT, 4>(const RunParams&, hipStream_t);

}  // namespace dhmc
