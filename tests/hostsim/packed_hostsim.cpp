// TEST INFRASTRUCTURE — a CPU simulation of the packed per-draw kernel (dynamichmc.jl_amd/csrc/packed_body.inc).
//
// The kernel's body is written against a small environment (the chain's lane group, LDS, an atomic add), so the same text that
// nuts_run_packed_kernel includes compiles here with g++ for L = 1: one "lane" holds all 64 padded coordinates of a chain, the
// group operations are identities, and the lane-local summation tree over 64 leaves is the ABI's tree (packed_core.hpp).  The
// CPU suite (tests/test_packed_hostsim.py) runs it against the oracle bit for bit — the tree logic, the gate, the RNG
// bookkeeping and the per-family arithmetic of the packed engine are checked without a GPU; what is left for the GPU suite is
// the DPP group operations and the launch geometry.  Nothing here is linked into libdhmc_amd.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../dynamichmc.jl_amd/csrc/packed_core.hpp"

namespace dhmc {

struct HostGroup {
    template <int N> static void sum_n(double (&)[N]) {}
    static double sum(double x) { return x; }
    static double first(double x) { return x; }
    static double prev(double) { return 0.0; }
    static double next(double) { return 0.0; }
    static int first_i(int x) { return x; }
    static double pick(double x, int) { return x; }
    static bool all(bool p) { return p; }
    static bool wave_any(bool p) { return p; }
};

template <int TGT>
static void run_chain(const RunParams& P, int place, double* lds_cold, double* lds_rows, double* lds_sc) {
    constexpr int L = 1, CPL = 64, GPW = 1;
    const int sub = 0, grp = 0;
    typedef HostGroup Grp;
    typedef dm_generic Pol;
    typedef pk::PackedTarget<TGT> PT;
#define PK_ATOMIC_ADD_ULL(ptr, v) (*(ptr) += (v))
#define PK_QUEUE_NEXT(ptr) ((*(ptr))++)
#define PK_LOAD_UINT(ptr) (*(ptr))
#define PK_ATOMIC_DEC_UINT(ptr) ((*(ptr))--)
#define PK_PH_DECL
#define PK_PH_END(i)
#define PK_PH_FLUSH(t)
#include "../../dynamichmc.jl_amd/csrc/packed_body.inc"
#undef PK_ATOMIC_DEC_UINT
#undef PK_LOAD_UINT
#undef PK_QUEUE_NEXT
#undef PK_ATOMIC_ADD_ULL
}

}  // namespace dhmc

// use_queue: one call of the body walks every place of the launch order through the kernel's queue of places (the only lane group
// of this "launch" takes place after place); else one call per place, as a launch without a queue assigns them.
extern "C" int hostsim_packed_run(int target, const dhmc::RunParams* Pin, int use_queue, int total_chains) {
    using namespace dhmc;
    RunParams P = *Pin;
    if (P.Dpad != 64 || P.D > 64) return 1;
    std::vector<double> ws((size_t)total_chains * P.nvec * P.Dpad, 0.0);   // (P.C: the places of this launch)
    P.st.ws = ws.data();
    std::vector<double> coldv(6 * 64, 0.0), rows((size_t)(P.pk_lds_levels > 0 ? P.pk_lds_levels : 1) * 4 * 64, 0.0), sc((size_t)P.max_depth * 4 + 4, 0.0);
    unsigned queue = (unsigned)P.pk_order_base + 1u;
    P.pk_queue = use_queue ? &queue : nullptr;
    for (int chain = P.pk_order_base; chain < (use_queue ? P.pk_order_base + 1 : P.C); ++chain) {
        switch (target) {
        case DHMC_TARGET_STD_NORMAL: run_chain<DHMC_TARGET_STD_NORMAL>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        case DHMC_TARGET_DIAG_NORMAL: run_chain<DHMC_TARGET_DIAG_NORMAL>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        case DHMC_TARGET_TRIDIAG_NORMAL: run_chain<DHMC_TARGET_TRIDIAG_NORMAL>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        case DHMC_TARGET_DENSE_NORMAL: run_chain<DHMC_TARGET_DENSE_NORMAL>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        case DHMC_TARGET_FUNNEL: run_chain<DHMC_TARGET_FUNNEL>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        case DHMC_TARGET_ALWAYS_DIVERGENT: run_chain<DHMC_TARGET_ALWAYS_DIVERGENT>(P, chain, coldv.data(), rows.data(), sc.data()); break;
        default: return 2;
        }
    }
    return 0;
}
extern "C" int hostsim_ws_nvec(int max_depth) { return dhmc::ws_nvec(max_depth); }
extern "C" int hostsim_sizeof_runparams() { return (int)sizeof(dhmc::RunParams); }
