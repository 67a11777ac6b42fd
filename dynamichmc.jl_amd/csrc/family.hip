// One target family's kernels: compiled once per family with -DDHMC_FAMILY=<functor of targets.hpp>.
#include "launch_impl.hpp"

namespace dhmc {
template int dispatch_family<DHMC_FAMILY>(int, Op, const void*, hipStream_t, const DenseMetric*);
}
