// One target family's kernels: compiled once per family with -DDHMC_FAMILY=<functor of targets.hpp>.
#include "launch_impl.hpp"

namespace dhmc {
template int dispatch_family<DHMC_FAMILY>(int, Op, const void*, hipStream_t, const DenseMetric*);
}

#ifdef DHMC_PHASE_TIMING
// tools/experiments/phase_timing.py: read (reset != 0: clear) the per-region clock totals of nuts_run_kernel
extern "C" int dhmc_debug_phase(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(dhmc::g_phase), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dhmc::g_phase), sizeof(z));
}
#endif
