// The multi-wave per-draw kernel (nuts_mw_kernel.hpp) lives in its own translation unit (mw.hip: compiled with machine
// LICM off, which keeps the loop-invariant fp64 polynomial constants out of long-lived VGPRs — no scratch at 128 VGPRs).
#pragma once
#include "nuts_kernels.hpp"

namespace dhmc {

template <class T, int NW>
void launch_run_mw(const RunParams& P, hipStream_t s);

}  // namespace dhmc
