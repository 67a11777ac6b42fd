// Kernel dispatch over (target family, slots per lane).  Every target family's kernels are instantiated in
// their own translation unit (family.hip compiled once per family, see Makefile), so the library builds in
// parallel; this header only declares the per-family entry point that dhmc_capi.hip (capi_internal.hpp capi::dispatch) calls.
#pragma once
#include "../../include/dhmc.h"
#include "dense_rounds.hpp"
#include "nuts_dense_kernel.hpp"
#include "nuts_kernels.hpp"
#include "packed_core.hpp"
#include "probe_kernels.hpp"

namespace dhmc {

struct RoundArgs {
    RunParams P;
    RoundBuffers R;
};

enum class Op { Run, Init, Search, RoundStart, RoundK0, RoundK2, RoundK3, ProbeTrajectory, ProbeRatios, RunPacked, RunPipeline };

// launches `op` of family T with NPL = npl slots per lane on stream s (M: the dense metric, or null)
template <class T>
int dispatch_family(int npl, Op op, const void* P, hipStream_t s, const DenseMetric* M);

}  // namespace dhmc
