// One instantiation of the per-draw kernel and nothing else: seconds to compile, for tools/isa_regions.py,
// tools/isa_liveness.py and register / spill experiments.  -DKO_T=StdNormalT -DKO_NPL=16
#include "../../dynamichmc.jl_amd/csrc/nuts_kernels.hpp"
#ifndef KO_T
#define KO_T StdNormalT
#endif
#ifndef KO_NPL
#define KO_NPL 16
#endif
namespace dhmc {
template __global__ void nuts_run_kernel<KO_T, KO_NPL, true>(RunParams);
}
