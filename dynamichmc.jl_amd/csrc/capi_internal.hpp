// Internal header of the C ABI's translation units that work on a context.  Not part of the ABI: include/dhmc.h is.
//   dhmc_capi.hip     contexts (create / destroy / init / positions / step sizes / step-size search) and dhmc_run: the per-draw
//                     kernel, the round engines' loops and the staging of host outputs
//   capi_metric.hip   metric setters / getters, the warmup windows' updates, the dense factorisation (dense_factor.hpp)
//   capi_probes.hip   the Diagnostics probes
//   capi_state.hip    checkpoint / resume blobs
// and without a context (capi_util.hpp only): capi_rtc.hip (the caller's device functor through hiprtc), capi_diag.hip (ESS /
// R-hat / tree-statistics summaries), capi_detmath.hip (the scalar math's self-test).
#pragma once
#include "capi_util.hpp"
#include "dense_rounds_k3b.hpp"
#include "external_rounds.hpp"
#include "logistic_rounds.hpp"
#include "launch.hpp"
#include "util_kernels.hpp"

using namespace dhmc;

struct dhmc_ctx {
    dhmc_config cfg{};
    int Dpad = 0, NPL = 0, nvec = 0;
    hipStream_t stream = nullptr;
    ChainArrays st{};
    TargetParams tp{};
    void* d_tp_a = nullptr;
    void* d_tp_b = nullptr;
    unsigned long long* d_counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    unsigned long long last_leapfrogs = 0;
    int l1_in_lds = 1;
    int k3_block = 1;
    DenseMetric dm{};          // DHMC_METRIC_DENSE only
    double* d_Minv = nullptr;
    double* d_WT = nullptr;
    double* d_fwork = nullptr;   // 4 × Dpad² doubles: work space of the device factorisation (dense_factor.hpp)
    int* d_fflags = nullptr;     // [2]: non-finite input, not positive definite
    RoundBuffers rb{};         // round-based dense engine (dense_rounds.hpp)
    RoundBuffers rbp[4]{};     // dense round engine: the batch is run as up to 4 parts on as many streams; every part has its own
                               // list and counters, the vectors are shared
    hipStream_t streams[4] = {};
    int dense_parts = 2;       // DHMC_DENSE=parts=
    int dense_row_lists = 1;   // DHMC_DENSE=row_lists=: products over the running chains only once some have finished
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_joins[4] = {};
    int dense_rounds = 1;
    int fuse_k2 = 1;           // DHMC_DENSE=fuse_k2=0: K3b leaves the next position update / density evaluation to K2 (dense_rounds_k3b.hpp)
    int dense_products = 2;    // dhmc_set_dense_products: 2 = the reference's recurrence; 1 = one M⁻¹ product per leapfrog (either dense engine)
    int per_chain_dense = 0;   // cfg.dense_per_chain: every chain has its own M⁻¹ / Wᵀ ([C][Dpad][Dpad]); wave-per-chain kernels only
    int logistic_rounds = 0;   // GEMM-gradient round engine for DHMC_TARGET_LOGISTIC with a diagonal metric
    int logistic_batched = 0;  // … and with it (or beyond 1024 coefficients) ℓ, ∇ℓ of all chains by the same GEMMs wherever they are needed
                               // outside a round: initialisation, step-size search, the Diagnostics probes (external_eval)
    LogisticRound lr{};
    int external = 0;          // DHMC_TARGET_EXTERNAL: density from the host's callback, round engine always
    int builtin_big = 0;       // a built-in family whose density the LIBRARY evaluates for all chains between kernels, where an external model's
                               // callback stands (more than 1024 coordinates; the logistic regression with a dense metric): the same engine
    int* d_all_rows = nullptr; // logistic_batched: the row list 0..C-1 and its length, for the GEMMs of the batched gradient
    double* d_big[2] = {};     // DHMC_TARGET_DENSE_NORMAL beyond 1024 coordinates: q − μ and P(q − μ) of all chains ([C][Dpad] each)
    dhmc_logdensity_fn ext_fn = nullptr;
    void* ext_user = nullptr;
    ExtSearchState* d_ss = nullptr;
    uint32_t* d_sflags = nullptr;   // [C][4]: ℓ(q′) (a double) and the position flag between two search kernels (dense)
    unsigned long long last_rounds = 0;
    uint64_t ws_bytes = 0;
    // host outputs of dhmc_run: persistent device staging (two buffers per field, grown on demand — no hipMalloc per call),
    // and a copy stream
    struct StageBuf { void* p = nullptr; size_t cap = 0; };
    StageBuf stage[2][10];
    hipStream_t copy_stream = nullptr;
    dhmc_allreduce_fn metric_allreduce = nullptr;   // dhmc_set_metric_allreduce: the shared dense metric adapted from the draws of all ranks
    void* metric_allreduce_user = nullptr;
    hipEvent_t ev_k0[2] = {}, ev_k1[2] = {}, ev_copy[2] = {};
    int* h_done = nullptr;     // page-locked [2][8]: the dense round engine's done-counters, read without draining the streams
    hipEvent_t ev_done[2] = {};
    int64_t host_chunk = 0;    // DHMC_HOST_CHUNK: transitions per chunk of a call with host outputs (0: ≈1 GiB of draws per chunk)
    const UserKernels* user = nullptr;   // target >= DHMC_TARGET_USER_BASE: the run-time compiled kernels of the caller's functor
    hipFunction_t user_packed = nullptr; // … and its packed per-draw kernel (D <= 64, a functor whose ℓ is a sum of per-coordinate terms)
    int user_packed_cpl = 0;             //     for this many coordinates per lane (the module holds one lane-group shape)
    hipFunction_t user_eval = nullptr;   // … beyond 1024 coordinates: its batched evaluation for the streaming round engine (user stays null)
    void* d_user_params = nullptr;
    // the per-draw kernels' launch order (nuts_kernels.hpp RunParams::launch_order): chains sorted by the leapfrog steps of the previous
    // launch, longest first, when one of them did more than a few percent above the mean (run_call)
    unsigned* d_chain_work = nullptr;      // [C]
    int* d_launch_order = nullptr;         // [C]
    std::vector<unsigned> h_chain_work;
    std::vector<int> h_launch_order;
    bool launch_order_valid = false;
    int launch_order_on = 1;               // DHMC_LAUNCH_ORDER=0: workgroup b takes chain b, always
    // the packed small-D engine (packed_core.hpp): several chains per wavefront for diagonal-metric chains of at most 64 coordinates
    int packed = 0;                        // the context's chains can run packed; DHMC_PACKED=0: never, =1: always, unset: by the previous launch's work
    int packed_force = 0;
    int pipeline = 0;                      // the context's chains can run as four-wave pipelines (nuts_pipeline_kernel.hpp): launches that the
                                           // previous launch showed to be held open by a few chains; DHMC_PIPELINE=0: never, =1: always
    int pipeline_force = 0;
    double mean_leapfrogs_per_transition = 0.0;   // of the previous call
    int* d_prog = nullptr;                 // [C] the end game of a packed launch: transitions of the call a chain has behind it …
    int* d_evicted = nullptr;              // [C] … and the chains the packed launch gave up, in the order it gave them up
    int pk_handover = -1;                  // DHMC_PK=handover=: the end game of a tail-bound packed launch starts at this many live lane groups (0: off; -1: what the pipeline kernel keeps resident)
    int many_chains_min = 0;               // DHMC_PK=many_chains=
    int pk_queue = 1;                      // DHMC_PK=queue=0: a packed launch starts a lane group per place (no queue of places)
    int pk_max_waves = 0;                  // DHMC_PK=max_waves=: the waves a queued packed launch starts (0: what the GPU holds at once)
    int tail_count = 0;                    // places at the head of the launch order whose work was far above the median's
    int pk_align = 0;                      // DHMC_PK=align=: transitions start on trips that are multiples of it (a power of two; 0: from the tree sizes)
    int pk_cpl = 0;                        // DHMC_PK=cpl=: coordinates per lane, 2 or 4 (0: by chain count, dhmc_run)
    int pk_lds_levels = -1;                // DHMC_PK=lds_levels=: suspended levels kept in LDS (-1: what the launch's occupancy leaves room for)
    int num_cus = 256;
    bool tail_bound = false;               // the previous launch was held open by a few chains with many times the mean's leapfrog steps
                                           // (dhmc_run: such launches go to the wave-per-chain kernel, whose leapfrog latency is lower)
    double* d_win = nullptr;   // dhmc_metric_window_begin: [2][C][Dpad] running mean / sum of squared deviations of every chain's draws
    int64_t win_n = -1;        // draws in the open metric window (-1: none open)
    bool poisoned = false;     // an external callback failed in the middle of dhmc_run: (q, ℓq, ∇ℓ) are inconsistent until dhmc_init / dhmc_import_state
    std::string err;
    std::vector<void*> allocs;
};

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return DHMC_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define DHMC_CHECK_USABLE(ctx)                                                                                     \
    do {                                                                                                          \
        if ((ctx)->poisoned) {                                                                                    \
            (ctx)->err = "the context's chain state is inconsistent after a failed log-density callback: call dhmc_init or dhmc_import_state"; \
            return DHMC_ERR_CALLBACK;                                                                             \
        }                                                                                                         \
    } while (0)


// one launch of KERNEL<slots per lane> for the context `c` in scope, on its stream
#define DHMC_EXT_NPL(KERNEL, GRID, ...)                                                                        \
    switch (c->NPL) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    case 32: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    default: hipLaunchKernelGGL((KERNEL<64>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    }

// ---- helpers shared by the translation units (defined in dhmc_capi.hip unless noted) --------------------------------------
namespace capi {

template <class Tp>
int dev_alloc(dhmc_ctx* c, Tp** p, size_t count) {
    void* v = nullptr;
    HIP_TRY(c, hipMalloc(&v, count * sizeof(Tp)));
    c->allocs.push_back(v);
    c->ws_bytes += count * sizeof(Tp);
    *p = (Tp*)v;
    return DHMC_OK;
}

// Stage a host array onto the device (returns a temp the caller frees), or pass through.
struct Staged {
    const void* dev = nullptr;
    void* temp = nullptr;
    ~Staged() { if (temp) { (void)hipDeviceSynchronize(); (void)hipFree(temp); } }   // error paths; stage_free is the normal one
};
int stage_in(dhmc_ctx* c, const void* p, size_t bytes, int on_device, Staged* s);
void stage_free(dhmc_ctx* c, Staged* s);

int npl_for_dim(int D, bool big);
// launches one operation (launch.hpp Op) of the context's target family, or of the caller's run-time compiled functor
int dispatch(const dhmc_ctx* c, Op op, const void* P, hipStream_t stream_override = nullptr, bool use_override = false);
// the logistic round engine's own kernels (logistic_rounds.hpp): 0 momentum, 1 position update, 3 gradient fold / second half step, 4 row list
void launch_logistic_op(int which, int npl, const RoundArgs& a, const LogisticRound& L, hipStream_t s);
int read_status(dhmc_ctx* c, std::vector<uint32_t>& st);
int status_code(dhmc_ctx* c);
int copy_out_padded(dhmc_ctx* c, const double* padded, double* dst, int on_device);
int copy_out_scalar(dhmc_ctx* c, const void* src, void* dst, size_t bytes, int on_device);
// ℓ and ∇ℓ of all chains between kernels: the host's callback, or the library's own batched evaluation (external_rounds.hpp)
int external_eval(dhmc_ctx* c, const double* q, bool active = false);
// capi_metric.hip
int upload_dense_metric(dhmc_ctx* c, const std::vector<double>& S, const std::vector<double>& W);
int device_dense_metric(dhmc_ctx* c, const double* src, int lsrc, int slot = -1);   // slot: a chain of a per-chain dense context, -1: all
void launch_metric(const dhmc_ctx* c, const double* draws, int64_t N);

}  // namespace capi
