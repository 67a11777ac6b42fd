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

// The host side of a packed launch, whichever module the kernel comes from: the grid (more places than the lane groups of
// pk_max_waves waves: that many waves start — the GPU holds them all at once — and the rest of the launch order waits in the queue:
// a group takes the next place when its chain is done, so that the lanes of chains with little work do not idle behind the longest
// chain of their wave), the queue's and the live-group counters, the LDS size.  Q: the parameter block as launched.
inline int packed_launch_prepare(const RunParams& P, hipStream_t s, RunParams* Qout, int* L_out, unsigned* waves_out, size_t* lds_out) {
    const int cpl = P.pk_cpl;
    const int L = pk::lanes_per_chain(P.D, cpl);
    if (L == 0 || (cpl != 2 && cpl != 4)) return DHMC_ERR_UNSUPPORTED;
    const int gpw = 64 / L;
    const int places = P.C - P.pk_order_base;
    int waves = (places + gpw - 1) / gpw;
    RunParams Q = P;
    if (Q.pk_queue && Q.pk_max_waves > 0 && waves > Q.pk_max_waves) {
        waves = Q.pk_max_waves;
        if (hipMemsetD32Async((hipDeviceptr_t)Q.pk_queue, P.pk_order_base + waves * gpw, 1, s) != hipSuccess) return DHMC_ERR_HIP;
    } else {
        Q.pk_queue = nullptr;
    }
    if (Q.pk_live && hipMemsetD32Async((hipDeviceptr_t)Q.pk_live, waves * gpw, 1, s) != hipSuccess) return DHMC_ERR_HIP;   // every group of the launch
    *Qout = Q; *L_out = L; *waves_out = (unsigned)waves;
    *lds_out = pk::lds_bytes_per_wave(L, cpl, P.max_depth, P.pk_lds_levels);
    return DHMC_OK;
}

// launches `op` of family T with NPL = npl slots per lane on stream s (M: the dense metric, or null)
template <class T>
int dispatch_family(int npl, Op op, const void* P, hipStream_t s, const DenseMetric* M);

}  // namespace dhmc
