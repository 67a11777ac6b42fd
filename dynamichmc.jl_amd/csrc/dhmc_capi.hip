// libdhmc_amd.so — host side of the C ABI declared in include/dhmc.h.
//
// Owns the opaque context (device-resident chain state + workspace), validates arguments where
// the reference uses @argcheck, launches the HIP kernels of nuts_kernels.hpp on the caller's
// stream, and maps per-chain failure words onto return codes.  There is NO CPU path in this
// library: without a HIP device every entry point that computes returns DHMC_ERR_NO_DEVICE.
#include "capi_internal.hpp"

using namespace capi;



namespace capi {

// slots per lane: the register/LDS-resident kernels go up to 16 (D <= 1024); the streaming round-engine kernels
// of an external model up to 64 (D <= 4096)
int npl_for_dim(int D, bool big) {
    int npl = (D + WAVE - 1) / WAVE;
    for (int cand : {1, 2, 4, 8, 16, 32, 64})
        if (npl <= cand) return (cand <= 16 || big) ? cand : 0;
    return 0;
}


void launch_logistic_op(int which, int npl, const RoundArgs& a, const LogisticRound& L, hipStream_t s) {
    const dim3 g(a.P.C), b(WAVE);
#define DHMC_NPL_SWITCH(KERNEL, ...)                                                              \
    switch (npl) {                                                                                \
    case 1: hipLaunchKernelGGL((KERNEL<1>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 2: hipLaunchKernelGGL((KERNEL<2>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 4: hipLaunchKernelGGL((KERNEL<4>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 8: hipLaunchKernelGGL((KERNEL<8>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 16: hipLaunchKernelGGL((KERNEL<16>), g, b, 0, s, __VA_ARGS__); break;                    \
    case 32: hipLaunchKernelGGL((KERNEL<32>), g, b, 0, s, __VA_ARGS__); break;                    \
    default: hipLaunchKernelGGL((KERNEL<64>), g, b, 0, s, __VA_ARGS__); break;                    \
    }
    switch (which) {
    case 0: DHMC_NPL_SWITCH(rounds_momentum_diag_kernel, a.P, a.R) break;
    case 1: DHMC_NPL_SWITCH(rounds_k1_diag_kernel, a.P, a.R) break;
    case 2:
        hipLaunchKernelGGL(logistic_link_kernel, dim3((unsigned)L.nz, a.P.C), b, 0, s, a.P, a.R, L);
        break;
    case 4:
        (void)hipMemsetAsync(L.act_count, 0, sizeof(int), s);
        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((a.P.C + 255) / 256), dim3(256), 0, s, a.P, a.R, L);
        break;
    default: DHMC_NPL_SWITCH(rounds_k2_logistic_kernel, a.P, a.R, L) break;
    }
#undef DHMC_NPL_SWITCH
}

int dispatch(const dhmc_ctx* c, Op op, const void* P, hipStream_t stream_override, bool use_override) {
    hipStream_t cs = use_override ? stream_override : c->stream;
    const DenseMetric* M = c->cfg.metric == DHMC_METRIC_DENSE ? &c->dm : nullptr;
    if (c->user) {     // the caller's functor: the same kernels, from the run-time compiled modules
        const UserKernels& U = *c->user;
        // the grid is the op's own chain count: the dense round engine runs half-batches (RoundArgs::P.C chains from P.chain_base)
        // on two streams, and the kernels index chain_base + blockIdx.x without a bounds guard
        unsigned grid = (unsigned)c->cfg.chains;
        auto launch = [&](hipFunction_t f, unsigned block, unsigned lds, void** args) {
            return f && grid > 0 && hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, cs, args, nullptr) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
        };
        DenseMetric dm = M ? *M : DenseMetric{};
        switch (op) {
        case Op::Run: {
            const RunParams& R = *(const RunParams*)P;
            grid = (unsigned)R.C;
            void* args[] = {const_cast<void*>(P), &dm};
            if (M) return launch(U.run_dense, WAVE, (unsigned)lds_bytes_dense(), args);
            return launch(R.l1_in_lds ? U.run_lds : U.run, WAVE,
                          (unsigned)(R.l1_in_lds ? lds_bytes(R.Dpad, true, lds_extra_levels(c->NPL)) : lds_bytes(R.Dpad, false, 0)), args);
        }
        case Op::RunPipeline: {
            const RunParams& R = *(const RunParams*)P;
            if (M || !U.pipeline || c->NPL > 2) return DHMC_ERR_UNSUPPORTED;
            grid = (unsigned)R.C;
            void* args[] = {const_cast<void*>(P)};
            return launch(U.pipeline, 4 * WAVE, (unsigned)pipeline_lds_bytes(c->NPL), args);
        }
        case Op::Init: { grid = (unsigned)((const InitParams*)P)->C; void* args[] = {const_cast<void*>(P)}; return launch(U.init, WAVE, 0, args); }
        case Op::Search: {
            grid = (unsigned)((const SearchParams*)P)->C;
            void* args[] = {const_cast<void*>(P), &dm};
            return M ? launch(U.search_dense, WAVE, 0, args) : launch(U.search, WAVE, (unsigned)(sizeof(double) * c->Dpad), args);
        }
        case Op::ProbeTrajectory: case Op::ProbeRatios: {     // Diagnostics.leapfrog_trajectory / explore_log_acceptance_ratios
            grid = (unsigned)((const ProbeParams*)P)->C;
            void* args[] = {const_cast<void*>(P), &dm};
            const bool traj = op == Op::ProbeTrajectory;
            if (M) return launch(traj ? U.probe_traj_dense : U.probe_ratio_dense, WAVE, 0, args);
            return launch(traj ? U.probe_traj : U.probe_ratio, WAVE, (unsigned)(sizeof(double) * c->Dpad), args);
        }
        case Op::RoundStart: return dispatch_family<StdNormalT>(c->NPL, op, P, cs, M);      // no density in this kernel
        case Op::RoundK0: case Op::RoundK2: case Op::RoundK3: {
            RoundArgs a = *(const RoundArgs*)P;
            grid = (unsigned)a.P.C;
            void* args[] = {&a.P, &a.R};
            if (op == Op::RoundK0) return launch(U.k0, WAVE, 0, args);
            if (op == Op::RoundK2) return launch(U.k2, WAVE, 0, args);
            return launch(U.k3, c->NPL >= 8 ? WAVE * k3b_waves(c->NPL) : WAVE, 0, args);   // a workgroup per chain from 512 coordinates
        }
        default: return DHMC_ERR_UNSUPPORTED;
        }
    }
    if (c->builtin_big) return dispatch_family<ExternalT>(c->NPL, op, P, cs, M);
    switch (c->cfg.target) {
    case DHMC_TARGET_STD_NORMAL: return dispatch_family<StdNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_DIAG_NORMAL: return dispatch_family<DiagNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_TRIDIAG_NORMAL: return dispatch_family<TridiagNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_FUNNEL: return dispatch_family<FunnelT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_LOGISTIC: return dispatch_family<LogisticT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_DENSE_NORMAL: return dispatch_family<DenseNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_ALWAYS_DIVERGENT: return dispatch_family<AlwaysDivergentT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_EXTERNAL: return dispatch_family<ExternalT>(c->NPL, op, P, cs, M);
    default: return DHMC_ERR_UNSUPPORTED;
    }
}

// Stage a host array onto the device (returns a temp the caller frees), or pass through.
int stage_in(dhmc_ctx* c, const void* p, size_t bytes, int on_device, Staged* s) {
    if (on_device) { s->dev = p; return DHMC_OK; }
    HIP_TRY(c, hipMalloc(&s->temp, bytes));
    HIP_TRY(c, hipMemcpyAsync(s->temp, p, bytes, hipMemcpyHostToDevice, c->stream));
    s->dev = s->temp;
    return DHMC_OK;
}
void stage_free(dhmc_ctx* c, Staged* s) {
    if (s->temp) { (void)hipStreamSynchronize(c->stream); (void)hipFree(s->temp); s->temp = nullptr; }
}

int read_status(dhmc_ctx* c, std::vector<uint32_t>& st) {
    st.resize(c->cfg.chains);
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->st.status, sizeof(uint32_t) * st.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
int status_code(dhmc_ctx* c) {
    std::vector<uint32_t> st;
    int rc = read_status(c, st);
    if (rc != DHMC_OK) return rc;
    for (uint32_t s : st)
        if (s) return DHMC_ERR_CHAIN_FAILURE;
    return DHMC_OK;
}

int copy_out_padded(dhmc_ctx* c, const double* padded, double* dst, int on_device) {
    int D = c->cfg.dim, C = c->cfg.chains;
    size_t n = (size_t)C * D;
    double* d = dst;
    DevBuf temp;
    if (!on_device) {
        HIP_TRY(c, hipMalloc(&temp.p, n * sizeof(double)));
        d = (double*)temp.p;
    }
    hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, D, c->Dpad, C, padded, d);
    HIP_TRY(c, hipGetLastError());
    if (!on_device) {
        HIP_TRY(c, hipMemcpyAsync(dst, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return DHMC_OK;
}
int copy_out_scalar(dhmc_ctx* c, const void* src, void* dst, size_t bytes, int on_device) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}

}  // namespace capi

extern "C" {

const char* dhmc_version(void) { return "dhmc_amd 0.3.0 (gfx950; detmath 2)"; }
int dhmc_detmath_version(void) { return DHMC_DETMATH_VERSION; }

const char* dhmc_last_error(const dhmc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dhmc_create(const dhmc_config* cfg, dhmc_ctx** out) {
    if (!cfg || !out) return DHMC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (cfg->dim <= 0 || cfg->chains <= 0 || cfg->chain_offset < 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (!(0 < cfg->max_depth && cfg->max_depth <= 32)) return DHMC_ERR_INVALID_ARGUMENT;  // NUTS.jl:190
    if (!(cfg->min_delta < 0)) return DHMC_ERR_INVALID_ARGUMENT;                           // NUTS.jl:191
    if (cfg->metric != DHMC_METRIC_DIAG && cfg->metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    const int D = cfg->dim;
    switch (cfg->target) {
    case DHMC_TARGET_STD_NORMAL: case DHMC_TARGET_ALWAYS_DIVERGENT: break;
    case DHMC_TARGET_FUNNEL: if (D < 2) return DHMC_ERR_INVALID_ARGUMENT; break;
    case DHMC_TARGET_DIAG_NORMAL: case DHMC_TARGET_TRIDIAG_NORMAL:
        if (!cfg->target_params || cfg->target_params_bytes != sizeof(double) * 2 * (size_t)D) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    case DHMC_TARGET_DENSE_NORMAL:
        if (!cfg->target_params || cfg->target_params_bytes != sizeof(double) * ((size_t)D + (size_t)D * D)) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    case DHMC_TARGET_LOGISTIC: {
        if (!cfg->target_params || cfg->target_params_bytes < 8) return DHMC_ERR_INVALID_ARGUMENT;
        int64_t n;
        std::memcpy(&n, cfg->target_params, 8);
        if (n <= 0 || cfg->target_params_bytes != 8 + sizeof(double) * (uint64_t)n * (D + 1)) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    }
    case DHMC_TARGET_EXTERNAL:
        break;
    default:
        if (cfg->target >= DHMC_TARGET_USER_BASE) {
            std::lock_guard<std::mutex> lock(g_user_mutex);
            if ((size_t)(cfg->target - DHMC_TARGET_USER_BASE) >= g_user_targets.size()) return DHMC_ERR_INVALID_ARGUMENT;
            if (D > 4096 || (cfg->metric == DHMC_METRIC_DENSE && cfg->dense_per_chain)) return DHMC_ERR_UNSUPPORTED;
            if (cfg->target_params_bytes % sizeof(double) != 0 || (cfg->target_params_bytes && !cfg->target_params)) return DHMC_ERR_INVALID_ARGUMENT;
            break;
        }
        return DHMC_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return DHMC_ERR_NO_DEVICE;
    dhmc_ctx* c = new (std::nothrow) dhmc_ctx();
    if (!c) return DHMC_ERR_HIP;
    c->cfg = *cfg;
    c->cfg.target_params = nullptr;
    // beyond 1024 coordinates the streaming round-engine kernels serve external models and the built-in normal families
    c->builtin_big = D > 1024 && (cfg->target == DHMC_TARGET_STD_NORMAL || cfg->target == DHMC_TARGET_DIAG_NORMAL ||
                                  cfg->target == DHMC_TARGET_TRIDIAG_NORMAL || cfg->target == DHMC_TARGET_FUNNEL ||
                                  cfg->target == DHMC_TARGET_DENSE_NORMAL || cfg->target == DHMC_TARGET_LOGISTIC ||
                                  cfg->target == DHMC_TARGET_ALWAYS_DIVERGENT);
    // … and the logistic regression with a shared dense metric at ANY width: its functor re-reads X twice per gradient and chain
    // (hundreds of ms per leapfrog at N = 10⁵), the batched evaluation is two GEMMs over all chains
    if (cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DENSE && !cfg->dense_per_chain) c->builtin_big = 1;
    // … and a caller's device functor beyond 1024 coordinates: evaluated for all chains by functor_eval_kernel (compiled at run time)
    const bool user_big = cfg->target >= DHMC_TARGET_USER_BASE && D > 1024;
    if (user_big) c->builtin_big = 1;
    c->NPL = npl_for_dim(D, cfg->target == DHMC_TARGET_EXTERNAL || c->builtin_big);
    if (cfg->target == DHMC_TARGET_EXTERNAL)
        if (const char* e = std::getenv("DHMC_FORCE_NPL")) {       // tests: run a narrow chain through the wide kernels
            const int f = std::atoi(e);
            if ((f == 32 || f == 64) && f >= c->NPL) c->NPL = f;
        }
    if (c->NPL == 0) { delete c; return DHMC_ERR_UNSUPPORTED; }
    c->Dpad = c->NPL * WAVE;
    // Round engines pay ≈8 launches per leapfrog round; they win once a round carries enough chains to fill the
    // chip (GEMM rows), otherwise the one-wave-per-chain kernels are faster.  DHMC_*_ROUNDS=0/1 overrides.
    // logistic regression, diagonal metric: the GEMM engine whatever the chain count — the wave-per-chain functor re-reads X twice per
    // gradient and chain (4 chains, N = 10⁵, p = 256: 306 ms per leapfrog against 0.27; N = 10³, p = 16: 140 µs against 105)
    c->logistic_rounds = cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DIAG && !c->builtin_big;
    if (const char* e = std::getenv("DHMC_LOGISTIC_ROUNDS"))
        c->logistic_rounds = cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DIAG && std::atoi(e) != 0 && !c->builtin_big;
    // dense metric: the wave-per-chain kernel (a matvec per chain from L2, several rows in flight, no launches per round) up to 128
    // coordinates at any chain count and up to 256 below 2048 chains — measured, D = 64 / 128 / 256: 346 / 110 / 29 M leapfrogs/s at
    // 4096 chains against 51 / 76 / 52 for the GEMM rounds, 58 / 33 / 23 against 3 / 3 / 10 at 256–512 chains — the rounds otherwise;
    // the same bits either way
    c->dense_rounds = D > 256 || (D > 128 && cfg->chains >= 2048);
    c->logistic_batched = cfg->target == DHMC_TARGET_LOGISTIC && (c->builtin_big || c->logistic_rounds);
    c->external = cfg->target == DHMC_TARGET_EXTERNAL || c->builtin_big;
    c->nvec = (cfg->metric == DHMC_METRIC_DENSE || c->logistic_rounds || c->external) ? wd_nvec(cfg->max_depth) : ws_nvec(cfg->max_depth);
    if (const char* e = std::getenv("DHMC_L1_LDS")) c->l1_in_lds = std::atoi(e) != 0;  // tuning knob
    if (const char* e = std::getenv("DHMC_K3_BLOCK")) c->k3_block = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_GRAPH")) c->use_graph = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_HOST_CHUNK")) c->host_chunk = std::atoll(e);
    if (const char* e = std::getenv("DHMC_LAUNCH_ORDER")) c->launch_order_on = std::atoi(e) != 0;
    // Chains of at most 64 coordinates with a diagonal metric: several chains per wavefront (packed_core.hpp) for the families that
    // have a packed evaluator; the same bits as the wave-per-chain kernel, which DHMC_PACKED=0 brings back.
    c->packed = cfg->metric == DHMC_METRIC_DIAG && pk::family_is_packed(cfg->target) && pk::dim_is_packed(D);
    if (const char* e = std::getenv("DHMC_PACKED")) { c->packed = c->packed && std::atoi(e) != 0; c->packed_force = c->packed; }
    // … and as a PIPELINE of four wavefronts per chain (integrator ‖ turn statistics ‖ visited statistic ‖ proposals, nuts_pipeline_kernel.hpp): the lowest
    // latency per leapfrog of a short chain, for launches that wait for a few deep chains
    c->pipeline = cfg->metric == DHMC_METRIC_DIAG && D <= 256 &&          // (rows of 64, 128 or 256 doubles: one, two or four slots per lane)
              (cfg->target == DHMC_TARGET_STD_NORMAL || cfg->target == DHMC_TARGET_DIAG_NORMAL || cfg->target == DHMC_TARGET_TRIDIAG_NORMAL ||
               cfg->target == DHMC_TARGET_FUNNEL || cfg->target == DHMC_TARGET_DENSE_NORMAL || cfg->target == DHMC_TARGET_ALWAYS_DIVERGENT);
    if (const char* e = std::getenv("DHMC_PIPELINE")) { c->pipeline = c->pipeline && std::atoi(e) != 0; c->pipeline_force = c->pipeline; }
    if (const char* e = std::getenv("DHMC_PK_QUEUE")) c->pk_queue = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_PK_MAX_WAVES")) c->pk_max_waves = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("DHMC_MANY_CHAINS")) c->many_chains_min = std::max(0, std::atoi(e));     // (tests: the chain count from which a launch counts as throughput-bound)
    if (const char* e = std::getenv("DHMC_PK_HANDOVER")) c->pk_handover = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("DHMC_HYBRID")) c->hybrid = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_HYBRID_SEGMENTS")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) c->hybrid_segments = v; }
    if (const char* e = std::getenv("DHMC_HYBRID_BUDGET")) c->hybrid_budget = std::atof(e);
    if (const char* e = std::getenv("DHMC_HYBRID_DEEP_CAP")) c->hybrid_deep_cap = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("DHMC_HYBRID_DEEP")) c->hybrid_deep_wave = std::string(e) == "wave";
    if (const char* e = std::getenv("DHMC_HYBRID_PROMOTE")) c->hybrid_promote = std::atof(e);
    if (const char* e = std::getenv("DHMC_HYBRID_DEEP_CUS")) c->hybrid_deep_cus = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("DHMC_HYBRID_MIN_CHAINS")) c->hybrid_min_chains = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("DHMC_PK_ALIGN")) { const int v = std::atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->pk_align = v; }
    if (const char* e = std::getenv("DHMC_PK_LDS_LEVELS")) c->pk_lds_levels = std::atoi(e);
    if (const char* e = std::getenv("DHMC_PK_CPL")) { const int v = std::atoi(e); if (v == 2 || v == 4) c->pk_cpl = v; }
    if (const char* e = std::getenv("DHMC_FUSE_K2")) c->fuse_k2 = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_DENSE_ROW_LISTS")) c->dense_row_lists = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_DENSE_PARTS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) c->dense_parts = v; }
    auto fail = [&](int rc) { dhmc_destroy(c); return rc; };
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(DHMC_ERR_NO_DEVICE);
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && n > 0) c->num_cus = n; }
    const size_t C = cfg->chains, Dp = c->Dpad;
    int rc;
    if ((rc = dev_alloc(c, &c->st.q, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.g, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.lq, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.minv, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.W, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.eps, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.da, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.transition, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.status, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.ws, C * (size_t)c->nvec * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->d_counter, 4))) return fail(rc);      // [0] leapfrog steps of a call; [1] a packed launch's queue of places; [2] the chains it gave up; [3] its lane groups that still have a chain
    if ((rc = dev_alloc(c, &c->d_chain_work, (size_t)cfg->chains))) return fail(rc);
    if ((rc = dev_alloc(c, &c->d_launch_order, (size_t)cfg->chains))) return fail(rc);
    if (hipMemset(c->st.q, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.g, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.da, 0, C * sizeof(DAState)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.status, 0, C * sizeof(uint32_t)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.transition, 0, C * sizeof(uint32_t)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (cfg->target >= DHMC_TARGET_USER_BASE) {
        // the caller's functor: its parameters to the device, its kernels from the cache or from hiprtc
        const size_t nb = (size_t)cfg->target_params_bytes;
        if (nb) {
            if (hipMalloc(&c->d_user_params, nb) != hipSuccess) return fail(DHMC_ERR_HIP);
            c->allocs.push_back(c->d_user_params);
            if (hipMemcpy(c->d_user_params, cfg->target_params, nb, hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        c->tp.a = (const double*)c->d_user_params;
        c->tp.n = (int64_t)(nb / sizeof(double));
        std::lock_guard<std::mutex> lock(g_user_mutex);
        UserTarget& U = g_user_targets[cfg->target - DHMC_TARGET_USER_BASE];
        const auto key = std::make_pair((int)cfg->device, c->NPL);
        auto it = U.built.find(key);
        if (it == U.built.end() && user_big) {
            std::vector<char> code;
            std::vector<std::string> low;
            if ((rc = rtc_compile(U.source, U.name, c->NPL, false, &code, &low))) return fail(rc);
            UserKernels K;
            auto load = [&]() { return rtc_load(code, low, &K.mod, {&K.eval}); };
            if ((rc = load())) {
                code.clear(); low.clear();
                if ((rc = rtc_compile(U.source, U.name, c->NPL, false, &code, &low, true)) || (rc = load())) return fail(rc);
            }
            it = U.built.emplace(key, K).first;
        }
        if (it == U.built.end()) {
            std::vector<char> code;
            std::vector<std::string> low;
            if ((rc = rtc_compile(U.source, U.name, c->NPL, false, &code, &low))) return fail(rc);
            UserKernels K;
            auto load = [&]() {
                return c->NPL <= 2 ? rtc_load(code, low, &K.mod, {&K.run_lds, &K.run, &K.init, &K.search, &K.probe_traj, &K.probe_ratio, &K.pipeline})
                                   : rtc_load(code, low, &K.mod, {&K.run_lds, &K.run, &K.init, &K.search, &K.probe_traj, &K.probe_ratio});
            };
            if ((rc = load())) {      // e.g. a cached object of another build of the tool chain: compile afresh once, replacing the file
                code.clear(); low.clear();
                if ((rc = rtc_compile(U.source, U.name, c->NPL, false, &code, &low, true)) || (rc = load())) return fail(rc);
            }
            {
                hipDeviceptr_t sym = nullptr;
                size_t bytes = 0;
                if (hipModuleGetGlobal(&sym, &bytes, K.mod, "dhmc_user_traits") == hipSuccess && bytes == sizeof(int))
                    (void)hipMemcpyDtoH(&K.traits, sym, sizeof(int));
                else
                    (void)hipGetLastError();
            }
            it = U.built.emplace(key, K).first;
        }
        if (cfg->metric == DHMC_METRIC_DENSE && !it->second.dense_mod && !user_big) {
            std::vector<char> code;
            std::vector<std::string> low;
            if ((rc = rtc_compile(U.source, U.name, c->NPL, true, &code, &low))) return fail(rc);
            UserKernels& K = it->second;
            auto load = [&]() { return rtc_load(code, low, &K.dense_mod, {&K.k0, &K.k2, &K.k3, &K.run_dense, &K.search_dense, &K.probe_traj_dense, &K.probe_ratio_dense}); };
            if ((rc = load())) {
                code.clear(); low.clear();
                if ((rc = rtc_compile(U.source, U.name, c->NPL, true, &code, &low, true)) || (rc = load())) return fail(rc);
            }
        }
        if (user_big) c->user_eval = it->second.eval;       // the streaming engine's kernels are the library's own (ExternalT)
        else c->user = &it->second;
        // the pipeline kernel for a functor whose gradient is recomputed from a stored position, up to two slots per lane
        if (c->user && c->user->pipeline && (c->user->traits & 1) && cfg->metric == DHMC_METRIC_DIAG && c->NPL <= 2) {
            c->pipeline = true;
            if (const char* e = std::getenv("DHMC_PIPELINE")) { c->pipeline = std::atoi(e) != 0; c->pipeline_force = c->pipeline; }
        }
    }
    if (cfg->target == DHMC_TARGET_DIAG_NORMAL || cfg->target == DHMC_TARGET_TRIDIAG_NORMAL) {
        std::vector<double> a(Dp, 0.0), b(Dp, 0.0);
        const double* src = (const double*)cfg->target_params;
        std::memcpy(a.data(), src, sizeof(double) * D);
        std::memcpy(b.data(), src + D, sizeof(double) * D);
        double *da = nullptr, *db = nullptr;
        if ((rc = dev_alloc(c, &da, Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &db, Dp))) return fail(rc);
        if (hipMemcpy(da, a.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(db, b.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = da;
        c->tp.b = db;
    }
    if (cfg->metric == DHMC_METRIC_DENSE) {   // GaussianKineticEnergy(N) as a dense identity
        c->per_chain_dense = cfg->dense_per_chain != 0;
        if (c->per_chain_dense && c->external) return fail(DHMC_ERR_UNSUPPORTED);
        const size_t nmat = c->per_chain_dense ? C : 1;
        if ((rc = dev_alloc(c, &c->d_Minv, nmat * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_WT, nmat * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_fwork, 4 * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_fflags, 2))) return fail(rc);
        c->dm = DenseMetric{c->d_Minv, c->d_WT, c->per_chain_dense ? Dp * Dp : (size_t)0};
        if (const char* e = std::getenv("DHMC_DENSE_ROUNDS")) c->dense_rounds = std::atoi(e) != 0;  // 0: wave-per-chain matvec kernel
        if (c->per_chain_dense) c->dense_rounds = 0;   // the GEMM engine shares one M⁻¹ across the rows of a product
        c->dense_products = c->per_chain_dense ? 2 : 1;
        if (const char* e = std::getenv("DHMC_DENSE_PRODUCTS")) { const int v = std::atoi(e); if (v == 1 || v == 2) c->dense_products = v; }
        std::vector<double> I((size_t)D * D, 0.0);
        for (int i = 0; i < D; ++i) I[(size_t)i * D + i] = 1.0;
        if ((rc = upload_dense_metric(c, I, I))) return fail(rc);
    }
    if (cfg->metric == DHMC_METRIC_DENSE || c->logistic_rounds || c->external) {   // buffers of the round engines
        if ((rc = dev_alloc(c, &c->rb.cp, C * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.cps, C * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.tbuf, C * Dp))) return fail(rc);
        if (cfg->metric == DHMC_METRIC_DENSE) {
            if ((rc = dev_alloc(c, &c->rb.cu, C * Dp))) return fail(rc);
            if (hipMemset(c->rb.cu, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        if ((rc = dev_alloc(c, &c->rb.ts, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.list, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.list_count, 8))) return fail(rc);
        if (cfg->metric == DHMC_METRIC_DENSE && !c->lr.act) {       // row lists of the dense round engine: [C] + a counter per part
            if ((rc = dev_alloc(c, &c->lr.act, C + 4))) return fail(rc);
            c->lr.act_count = c->lr.act + C;
        }
        c->rb.done_count = c->rb.list_count + 1;
        if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) return fail(DHMC_ERR_HIP);
        for (int i = 2; i < 4; ++i)
            if (hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        for (int i = 1; i < 4; ++i)
            if (hipEventCreateWithFlags(&c->ev_joins[i], hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.ts, 0, C * sizeof(TreeState)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.cp, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.cps, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.tbuf, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    }
    c->tp.Dpad = (int32_t)Dp;
    if (cfg->target == DHMC_TARGET_DENSE_NORMAL) {
        const double* src = (const double*)cfg->target_params;
        std::vector<double> mu(Dp, 0.0), Pm(Dp * Dp, 0.0);
        std::memcpy(mu.data(), src, sizeof(double) * D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) Pm[(size_t)i * Dp + j] = (i <= j) ? src[D + (size_t)i * D + j] : src[D + (size_t)j * D + i];
        double *dmu = nullptr, *dP = nullptr;
        if ((rc = dev_alloc(c, &dmu, Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &dP, Dp * Dp))) return fail(rc);
        if (hipMemcpy(dmu, mu.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dP, Pm.data(), Dp * Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = dmu; c->tp.b = dP;
    }
    if (cfg->target == DHMC_TARGET_LOGISTIC) {
        // X [npad][Dpad] row-major and Xᵀ [Dpad][npad], zero padded (GEMM operands of the round engine; the
        // wave-per-chain functor reads the same arrays), y [npad]
        int64_t n;
        std::memcpy(&n, cfg->target_params, 8);
        const double* X = (const double*)((const char*)cfg->target_params + 8);
        const double* y = X + (size_t)n * D;
        const size_t npad = ((size_t)n + WAVE - 1) / WAVE * WAVE;
        std::vector<double> xp(npad * Dp, 0.0), xt(Dp * npad, 0.0), yp(npad, 0.0);
        for (int64_t i = 0; i < n; ++i)
            for (int d = 0; d < D; ++d) {
                xp[(size_t)i * Dp + d] = X[(size_t)i * D + d];
                xt[(size_t)d * npad + i] = X[(size_t)i * D + d];
            }
        for (int64_t i = 0; i < n; ++i) yp[i] = y[i];
        double *dx = nullptr, *dxt = nullptr, *dy = nullptr;
        if ((rc = dev_alloc(c, &dx, xp.size()))) return fail(rc);
        if ((rc = dev_alloc(c, &dxt, xt.size()))) return fail(rc);
        if ((rc = dev_alloc(c, &dy, yp.size()))) return fail(rc);
        if (hipMemcpy(dx, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dxt, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dy, yp.data(), yp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = dx; c->tp.b = dxt; c->tp.c = dy; c->tp.n = n; c->tp.npad = (int64_t)npad; c->tp.Dpad = (int32_t)Dp;
        if (c->logistic_batched) {        // the gradient of all chains by GEMMs, in the rounds and between other kernels (external_eval)
            if ((rc = dev_alloc(c, &c->lr.H, C * npad))) return fail(rc);
            c->lr.nz = (int)((npad + DHMC_LOGISTIC_BLOCK - 1) / DHMC_LOGISTIC_BLOCK);
            if ((rc = dev_alloc(c, &c->lr.P, (size_t)c->lr.nz * C * Dp))) return fail(rc);
            if ((rc = dev_alloc(c, &c->lr.S1P, (size_t)c->lr.nz * C))) return fail(rc);
            if ((rc = dev_alloc(c, &c->d_all_rows, C + 1))) return fail(rc);
            if (c->builtin_big && !c->lr.act) {                          // row list of a round (the rounds engine below has its own)
                if ((rc = dev_alloc(c, &c->lr.act, C + 4))) return fail(rc);
                c->lr.act_count = c->lr.act + C;
            }
            hipLaunchKernelGGL(builtin_all_rows_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, nullptr, (int)C, c->d_all_rows, c->d_all_rows + C);
            if (hipDeviceSynchronize() != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        if (c->logistic_rounds) {         // (H, P, S1P: above)
            if ((rc = dev_alloc(c, &c->lr.S1, C))) return fail(rc);
            if ((rc = dev_alloc(c, &c->lr.act, C + 1))) return fail(rc);
            c->lr.act_count = c->lr.act + C;
            if ((rc = dev_alloc(c, &c->d_ss, C))) return fail(rc);      // the batched step-size search's per-chain state
        }
    }
    if (c->builtin_big && cfg->target == DHMC_TARGET_DENSE_NORMAL)
        for (int i = 0; i < 2; ++i)
            if ((rc = dev_alloc(c, &c->d_big[i], C * Dp))) return fail(rc);
    if (c->external) {
        if ((rc = dev_alloc(c, &c->lr.S1, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_ss, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_sflags, 4 * C))) return fail(rc);
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return fail(DHMC_ERR_HIP);
    // unit metric, ε unspecified
    {
        std::vector<double> ones(C * Dp, 1.0), w(C * Dp, 0.0), nanv(C, std::nan(""));
        for (size_t ch = 0; ch < C; ++ch)
            for (int e = 0; e < D; ++e) w[ch * Dp + e] = 1.0;
        if (hipMemcpy(c->st.minv, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(c->st.W, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(c->st.eps, nanv.data(), C * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
    }
    *out = c;
    return DHMC_OK;
}

int dhmc_host_alloc(void** out, uint64_t nbytes) {
    if (!out) return DHMC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (nbytes == 0) return DHMC_OK;
    return hipHostMalloc(out, (size_t)nbytes, hipHostMallocDefault) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
}
int dhmc_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? DHMC_OK : DHMC_ERR_HIP; }

int dhmc_destroy(dhmc_ctx* c) {
    if (!c) return DHMC_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    for (hipStream_t st : {c->stream_deep, c->stream_bulk}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t ev : {c->ev_round[0], c->ev_round[1], c->ev_round[2], c->ev_round[3], c->ev_join2}) if (ev) (void)hipEventDestroy(ev);
    for (int i = 2; i < 4; ++i)
        if (c->streams[i]) { (void)hipStreamSynchronize(c->streams[i]); (void)hipStreamDestroy(c->streams[i]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 1; i < 4; ++i) if (c->ev_joins[i]) (void)hipEventDestroy(c->ev_joins[i]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_done) (void)hipHostFree(c->h_done);
    for (int b = 0; b < 2; ++b) if (c->ev_done[b]) (void)hipEventDestroy(c->ev_done[b]);
    for (int b = 0; b < 2; ++b) {
        for (auto& sb : c->stage[b]) if (sb.p) (void)hipFree(sb.p);
        if (c->ev_k0[b]) (void)hipEventDestroy(c->ev_k0[b]);
        if (c->ev_k1[b]) (void)hipEventDestroy(c->ev_k1[b]);
        if (c->ev_copy[b]) (void)hipEventDestroy(c->ev_copy[b]);
    }
    delete c;
    return DHMC_OK;
}

// ---- DHMC_TARGET_EXTERNAL (external_rounds.hpp) -------------------------------------------------
}  // extern "C"
namespace capi {
#define DHMC_EXT_NPL_FWD(KERNEL, GRID, ...)                                                                    \
    switch (c->NPL) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    case 32: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    default: hipLaunchKernelGGL((KERNEL<64>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    }
// ℓ and ∇ℓ of `q` ([C][Dpad], device) for all chains through the host's callback: lq -> c->lr.S1, grad -> c->rb.tbuf
// `active` (the logistic family's GEMM evaluation only): the rows of the chains in a leaf phase, listed in c->lr.act by the caller —
// the round loops pass it, so that chains which have finished their transitions are not multiplied
int external_eval(dhmc_ctx* c, const double* q, bool active) {
    if (c->logistic_batched) {
        // the GEMM gradient of the logistic round engine over all chains (logistic_rounds.hpp), folded by builtin_logistic_fold_kernel
        const int C = c->cfg.chains, ld = c->Dpad, npad = (int)c->tp.npad;
        RunParams P{};
        P.D = c->cfg.dim; P.Dpad = ld; P.C = C; P.tp = c->tp;
        RoundBuffers R{};
        LogisticRound L = c->lr;
        if (!(active && c->lr.act)) { L.act = c->d_all_rows; L.act_count = c->d_all_rows + C; }                          // else: every chain
        launch_gemm_list(q, ld, c->tp.b, npad, L.H, npad, C, ld, npad, L.act, L.act_count, c->stream);                    // η = Q·Xᵀ
        hipLaunchKernelGGL(logistic_link_kernel, dim3((unsigned)L.nz, C), dim3(WAVE), 0, c->stream, P, R, L);            // r, the blocks' sums
        launch_gemm_splitk(L.H, npad, c->tp.a, ld, L.P, ld, (size_t)C * ld, C, npad, ld, DHMC_LOGISTIC_BLOCK, L.act, L.act_count,
                           c->stream);                                                                                   // Xᵀr, block by block
        DHMC_EXT_NPL_FWD(builtin_logistic_fold_kernel, dim3(C), C, ld, q, L, c->lr.S1, c->rb.tbuf)                         // (listed chains only)
        return DHMC_OK;
    }
    if (c->builtin_big && c->cfg.target == DHMC_TARGET_DENSE_NORMAL) {
        const dim3 g(c->cfg.chains), b(WAVE);
        const int ld = c->Dpad;
        if (c->NPL == 32) hipLaunchKernelGGL((builtin_dense_normal_pre_kernel<32>), g, b, 0, c->stream, ld, q, c->tp.a, c->d_big[0]);
        else hipLaunchKernelGGL((builtin_dense_normal_pre_kernel<64>), g, b, 0, c->stream, ld, q, c->tp.a, c->d_big[0]);
        launch_gemm_rows(c->d_big[0], c->tp.b, c->d_big[1], ld, c->cfg.chains, nullptr, nullptr, c->stream);      // P·d for every chain
        if (c->NPL == 32) hipLaunchKernelGGL((builtin_dense_normal_post_kernel<32>), g, b, 0, c->stream, ld, (const double*)c->d_big[0], (const double*)c->d_big[1], c->lr.S1, c->rb.tbuf);
        else hipLaunchKernelGGL((builtin_dense_normal_post_kernel<64>), g, b, 0, c->stream, ld, (const double*)c->d_big[0], (const double*)c->d_big[1], c->lr.S1, c->rb.tbuf);
        return DHMC_OK;
    }
    if (c->user_eval) {       // a caller's functor beyond 1024 coordinates (nuts_kernels.hpp functor_eval_kernel)
        TargetParams tp = c->tp;
        int D = c->cfg.dim, ld = c->Dpad;
        double* lq = c->lr.S1;
        double* grad = c->rb.tbuf;
        void* args[] = {&tp, &D, &ld, (void*)&q, &lq, &grad};
        if (hipModuleLaunchKernel(c->user_eval, (unsigned)c->cfg.chains, 1, 1, WAVE, 1, 1, 0, c->stream, args, nullptr) != hipSuccess) {
            c->err = "dhmc: launch of the functor's evaluation kernel failed";
            return DHMC_ERR_HIP;
        }
        return DHMC_OK;
    }
    if (c->builtin_big) {
        const int kind = c->cfg.target == DHMC_TARGET_STD_NORMAL ? 0 : c->cfg.target == DHMC_TARGET_DIAG_NORMAL ? 1 :
                         c->cfg.target == DHMC_TARGET_TRIDIAG_NORMAL ? 2 : c->cfg.target == DHMC_TARGET_FUNNEL ? 3 : 4;
        const dim3 g(c->cfg.chains), b(WAVE);
        if (c->NPL == 32)
            hipLaunchKernelGGL((builtin_normal_eval_kernel<32>), g, b, 0, c->stream, kind, c->cfg.dim, c->Dpad, q, c->tp.a, c->tp.b, c->lr.S1, c->rb.tbuf);
        else
            hipLaunchKernelGGL((builtin_normal_eval_kernel<64>), g, b, 0, c->stream, kind, c->cfg.dim, c->Dpad, q, c->tp.a, c->tp.b, c->lr.S1, c->rb.tbuf);
        return DHMC_OK;
    }
    if (!c->ext_fn) { c->err = "DHMC_TARGET_EXTERNAL: no callback set (dhmc_set_logdensity_callback)"; return DHMC_ERR_CALLBACK; }
    const int rc = c->ext_fn(c->ext_user, q, c->cfg.chains, c->Dpad, c->cfg.dim, c->lr.S1, c->rb.tbuf, (void*)c->stream);
    if (rc != 0) { c->err = "DHMC_TARGET_EXTERNAL: the callback returned " + std::to_string(rc); return DHMC_ERR_CALLBACK; }
    return DHMC_OK;
}
}  // namespace capi
extern "C" {

int dhmc_set_logdensity_callback(dhmc_ctx* c, dhmc_logdensity_fn fn, void* user) {
    if (!c || !c->external || c->builtin_big) return DHMC_ERR_INVALID_ARGUMENT;
    c->ext_fn = fn;
    c->ext_user = user;
    return DHMC_OK;
}

int dhmc_set_stream(dhmc_ctx* c, void* s) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    c->stream = (hipStream_t)s;
    return DHMC_OK;
}

int dhmc_init(dhmc_ctx* c, const double* q0, int q0_on_device) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // (the poisoned mark is lifted only once the new (q, ℓq, ∇ℓ) are complete; a failure after the positions were
    // overwritten sets it)
    Staged s;
    if (q0) {
        int rc = stage_in(c, q0, sizeof(double) * (size_t)c->cfg.chains * c->cfg.dim, q0_on_device, &s);
        if (rc) return rc;
    }
    InitParams P{c->cfg.dim, c->Dpad, c->cfg.chains, c->cfg.chain_offset, c->cfg.seed, (const double*)s.dev, c->st, c->tp};
    int rc = DHMC_OK;
    const bool batched = c->external || c->logistic_batched;     // ℓ, ∇ℓ of all chains between kernels (callback / library GEMMs)
    if (batched) { DHMC_EXT_NPL(external_init_positions_kernel, dim3(c->cfg.chains), P) }
    else rc = dispatch(c, Op::Init, &P);
    if (rc) { stage_free(c, &s); return rc; }
    c->poisoned = true;    // st.q is being overwritten: inconsistent with ℓq, ∇ℓ until the evaluation below has succeeded
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    if (batched) {         // the positions are set; ℓ and ∇ℓ come from the callback, then evaluate_ℓ(strict) (mcmc.jl:131)
        if ((rc = external_eval(c, c->st.q))) return rc;
        DHMC_EXT_NPL(external_init_finish_kernel, dim3(c->cfg.chains), c->cfg.dim, c->Dpad, c->st, c->lr.S1, c->rb.tbuf)
        HIP_TRY(c, hipGetLastError());
    }
    c->poisoned = false;
    c->win_n = -1;         // new chains: an open metric window is discarded
    c->launch_order_valid = false;
    return status_code(c);
}

int dhmc_set_position(dhmc_ctx* c, const double* q, int on_device) {
    if (!c || !q) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // dhmc_init does the evaluation (and resets κ, ϵ, counters): keep those aside and put them back
    const size_t C = c->cfg.chains, Dp = c->Dpad;
    struct Keep { void* live; size_t bytes; DevBuf copy; };
    Keep keep[7] = {{c->st.minv, C * Dp * sizeof(double), {}}, {c->st.W, C * Dp * sizeof(double), {}}, {c->st.eps, C * sizeof(double), {}},
                    {c->st.da, C * sizeof(DAState), {}}, {c->st.transition, C * sizeof(uint32_t), {}},
                    {nullptr, 0, {}}, {nullptr, 0, {}}};   // (dhmc_init leaves a dense metric alone)
    for (auto& k : keep) {
        if (!k.live) continue;
        HIP_TRY(c, hipMalloc(&k.copy.p, k.bytes));
        HIP_TRY(c, hipMemcpyAsync(k.copy.p, k.live, k.bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    const int rc = dhmc_init(c, q, on_device);    // (a failed evaluation leaves the context poisoned, as in dhmc_init)
    for (auto& k : keep)
        if (k.live) HIP_TRY(c, hipMemcpyAsync(k.live, k.copy.p, k.bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

int dhmc_get_position(dhmc_ctx* c, double* q, double* lq, double* grad, int on_device) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if (q && (rc = copy_out_padded(c, c->st.q, q, on_device))) return rc;
    if (grad && (rc = copy_out_padded(c, c->st.g, grad, on_device))) return rc;
    if (lq && (rc = copy_out_scalar(c, c->st.lq, lq, sizeof(double) * c->cfg.chains, on_device))) return rc;
    return DHMC_OK;
}

int dhmc_set_stepsize(dhmc_ctx* c, const double* eps, int per_chain, int on_device) {
    if (!c || !eps) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains;
    std::vector<double> h(C);
    if (on_device) {
        if (!per_chain) return DHMC_ERR_INVALID_ARGUMENT;
        HIP_TRY(c, hipMemcpyAsync(h.data(), eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        for (int i = 0; i < C; ++i) h[i] = eps[per_chain ? i : 0];
    }
    for (int i = 0; i < C; ++i)
        if (!(h[i] > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    HIP_TRY(c, hipMemcpyAsync(c->st.eps, h.data(), sizeof(double) * C, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}

int dhmc_get_stepsize(dhmc_ctx* c, double* eps, int on_device) {
    if (!c || !eps) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_scalar(c, c->st.eps, eps, sizeof(double) * c->cfg.chains, on_device);
}

int dhmc_get_status(dhmc_ctx* c, uint32_t* status) {
    if (!c || !status) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_scalar(c, c->st.status, status, sizeof(uint32_t) * c->cfg.chains, 0);
}

int dhmc_find_initial_stepsize(dhmc_ctx* c, const dhmc_stepsize_search* p) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    dhmc_stepsize_search d{0.1, std::log(0.8), 400, 0};
    if (p) d = *p;
    if (!(std::isfinite(d.log_threshold) && d.log_threshold < 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:31
    if (!(std::isfinite(d.initial_eps) && 0 < d.initial_eps)) return DHMC_ERR_INVALID_ARGUMENT;      // :32
    if (!(d.maxiter_crossing >= 50)) return DHMC_ERR_INVALID_ARGUMENT;                                // :33
    const int C = c->cfg.chains;
    std::vector<double> h(C);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->st.eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (double e : h)
        if (!std::isnan(e)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:137 "stepsize ϵ manually specified"
    if (c->external && c->cfg.metric == DHMC_METRIC_DENSE) {
        // dense metric: p₀ = z·Wᵀ, M⁻¹pₘ and p′♯ are GEMMs over all chains between the search kernels
        ExtSearchParams E{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold, d.maxiter_crossing,
                          c->st, c->d_ss, c->rb.cps, c->rb.cp, c->lr.S1, c->rb.tbuf, c->rb.list_count};
        const int ld = c->Dpad;
        const size_t p1_stride = (size_t)c->nvec * c->Dpad;
        int remaining = 0;
        HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
        DHMC_EXT_NPL(ext_search_dense_z_kernel, dim3(C), E, c->rb.cp)
        launch_gemm_rows(c->rb.cp, c->d_WT, c->rb.cps, ld, C, nullptr, nullptr, c->stream);                 // p₀ = z·Wᵀ
        launch_gemm_rows(c->rb.cps, c->d_Minv, c->rb.tbuf, ld, C, nullptr, nullptr, c->stream);             // p₀♯
        DHMC_EXT_NPL(ext_search_dense_begin_kernel, dim3(C), E, (const double*)c->rb.tbuf, c->rb.cp)
        HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        while (remaining > 0) {
            launch_gemm_rows(c->rb.cp, c->d_Minv, c->rb.tbuf, ld, C, nullptr, nullptr, c->stream);          // M⁻¹pₘ
            DHMC_EXT_NPL(ext_search_dense_trial_kernel, dim3(C), E, (const double*)c->rb.tbuf)
            int rc = external_eval(c, c->rb.cp);
            if (rc) return rc;
            DHMC_EXT_NPL(ext_search_dense_p1_kernel, dim3(C), E, c->d_sflags, c->st.ws, p1_stride)
            launch_gemm(c->st.ws, (int)p1_stride, c->d_Minv, ld, c->rb.tbuf, ld, C, ld, ld, c->stream);     // p′♯
            HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
            DHMC_EXT_NPL(ext_search_dense_decide_kernel, dim3(C), E, (const double*)c->st.ws, p1_stride, (const double*)c->rb.tbuf,
                         (const uint32_t*)c->d_sflags, c->rb.cp)
            HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        HIP_TRY(c, hipGetLastError());
        return status_code(c);
    }
    if (c->external || c->logistic_batched) {
        // the same bracketing search, all chains per callback: trial positions -> callback -> one decision per chain
        ExtSearchParams E{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold, d.maxiter_crossing,
                          c->st, c->d_ss, c->rb.cps, c->rb.cp, c->lr.S1, c->rb.tbuf, c->rb.list_count};
        int remaining = 0;
        HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
        DHMC_EXT_NPL(ext_search_begin_kernel, dim3(C), E)
        HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        while (remaining > 0) {
            int rc = external_eval(c, c->rb.cp);
            if (rc) return rc;
            HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
            DHMC_EXT_NPL(ext_search_step_kernel, dim3(C), E)
            HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        HIP_TRY(c, hipGetLastError());
        return status_code(c);
    }
    SearchParams P{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold,
                   d.maxiter_crossing, c->st, c->tp};
    int rc = dispatch(c, Op::Search, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    return status_code(c);
}

namespace {
int run_call(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (da) {
        if (!(0 < da->delta && da->delta < 1)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:108
        if (!(da->gamma > 0)) return DHMC_ERR_INVALID_ARGUMENT;                   // :109
        if (!(0.5 < da->kappa && da->kappa <= 1)) return DHMC_ERR_INVALID_ARGUMENT;  // :110
        if (!(da->t0 >= 0)) return DHMC_ERR_INVALID_ARGUMENT;                     // :111
    }
    const int C = c->cfg.chains, D = c->cfg.dim;
    if (!da || da->init) {
        std::vector<double> h(C);
        HIP_TRY(c, hipMemcpyAsync(h.data(), c->st.eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (double e : h)
            if (!(e > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    }
    c->last_ms = 0.0;
    c->last_leapfrogs = 0;
    if (N == 0) return status_code(c);

    RunParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset;
    P.max_depth = c->cfg.max_depth; P.nvec = c->nvec; P.min_delta = c->cfg.min_delta; P.seed = c->cfg.seed;
    P.N = N; P.st = c->st; P.tp = c->tp; P.leapfrog_counter = c->d_counter;
    P.l1_in_lds = c->l1_in_lds;
    P.k3_block = c->k3_block;
    P.one_product = c->cfg.metric == DHMC_METRIC_DENSE && c->dense_products == 1;
    const bool per_draw_kernel = !c->external && !c->logistic_rounds && !(c->cfg.metric == DHMC_METRIC_DENSE && c->dense_rounds);
    if (per_draw_kernel && c->launch_order_on && c->d_chain_work) {
        P.chain_work = c->d_chain_work;
        P.launch_order = c->launch_order_valid ? c->d_launch_order : nullptr;
    }
    // Which per-draw kernel.  The packed kernel serves several chains per instruction — 1.1 (60 coordinates) to 5 times (8) the
    // wave-per-chain kernel's throughput on chains of even work — but a trip of its loop costs more clocks than the other
    // kernel's leapfrog (4 300 against 3 700 at 30 coordinates), and a launch ends with its slowest chain, whose leapfrogs are
    // sequential: when the previous launch was held open by a few chains with many times the mean's work (Neal's funnel: chains in
    // the neck run trees of the depth limit, 13 × the mean over 1000 transitions), the launch goes to the kernel with the lowest
    // latency per leapfrog: the four-wave pipeline (nuts_pipeline_kernel.hpp, ≈ 2 000 clocks) where the family allows it, else the
    // wave-per-chain kernel.  The same bits whichever runs (all are checked against the oracle).  DHMC_PACKED / DHMC_PIPELINE = 1 / 0:
    // always / never.
    // … and when the chains are so few that each of their four waves gets a SIMD of its own (C <= the number of CUs — the reference's
    // typical handful of chains): such a launch is all latency, whatever its trees look like
    // … but not when there are so many chains that throughput counts again (the pipeline kernel gives a chain four waves): the packed
    // kernel with its queue of places then, and the pipeline kernel for its END GAME (below).  One call of 1000 transitions of the
    // funnel, leapfrog steps/s: 4096 chains pipeline 3.5e8, packed + end game 3.4e8, wave 2.0e8; 8192: 5.1e8, 5.9e8, 4.0e8; 16384:
    // 6.4e8, 9.5e8, 6.6e8; 32768: 7.0e8, 1.49e9, 8.7e8 (packed alone 9.5e8) (profiles/r05_packed_queue_rounds.txt): from 24
    // chains per CU on.  A family without a packed evaluator keeps the wave kernel from 13 chains per pipeline block slot on.
    // … and only when the trees are large: the four waves fill and drain once per transition (≈ 2.5 µs), so 4 chains of a 100-dim
    // standard normal (7 leapfrogs per transition) take 2.1 µs per leapfrog here against 1.5 in the wave kernel, the same chains on
    // a 100-dim funnel (66 per transition) 1.6 against 2.6 (profiles/r05_pipeline_kernel.txt).  From the previous call's mean.
    const bool few_chains = C <= c->num_cus && c->mean_leapfrogs_per_transition >= 24.0, many_chains = C > (c->many_chains_min > 0 ? c->many_chains_min : c->packed && c->pk_handover != 0 ? 24 * c->num_cus
                                              : 13 * (int)((size_t)160 * 1024 / pipeline_lds_bytes(c->NPL <= 4 ? c->NPL : 4)) * c->num_cus);   // (5, 2 or 1 blocks per CU)
    const bool pipeline = per_draw_kernel && c->pipeline && !c->packed_force && (c->pipeline_force || (c->tail_bound && !many_chains) || few_chains);
    const bool packed = !pipeline && per_draw_kernel && c->packed && (c->packed_force || !c->tail_bound || many_chains);
    const Op run_op = pipeline ? Op::RunPipeline : packed ? Op::RunPacked : Op::Run;
    if (per_draw_kernel && std::getenv("DHMC_DEBUG_ORDER"))
        std::fprintf(stderr, "[dhmc] engine: %s (N=%lld, chains %d)\n", pipeline ? "pipeline" : packed ? "packed" : "wave", (long long)N, C);
    // ROUNDS (DHMC_HYBRID=1; off by default): the call in rounds, the bulk packed and the deepest chains in the pipeline kernel beside
    // it on CUs of their own (below, where the rounds are launched).  Chains are independent: which kernel runs which part of a
    // chain changes none of its bits.  Measured at 32768 funnel chains: 9.7e8 against 9.5e8 for one packed launch — the funnel
    // keeps hundreds of chains at the depth limit at any time, more than the pipeline kernel can serve at its latency, and a round
    // of the packed kernel then still ends with such a chain.
    const bool hybrid = per_draw_kernel && c->hybrid && c->packed && c->pipeline && !c->packed_force && !c->pipeline_force &&
                        c->tail_bound && C > (c->hybrid_min_chains > 0 ? c->hybrid_min_chains : 32 * c->num_cus) && N >= 8LL * c->hybrid_segments &&
                        c->d_chain_work && c->launch_order_on;
    // END GAME of a tail-bound packed launch (many chains: the rule above): once few lane groups still have a chain — no more than the
    // pipeline kernel keeps resident — the packed kernel gives those chains up at their next transition boundary and the pipeline
    // kernel finishes them at a third of the latency per leapfrog: the launch's deepest chains, which would otherwise run on alone
    // at 2.5 µs per trip (RunParams::pk_live, pk_handover_below; DHMC_PK_HANDOVER = the threshold, 0: off).
    const bool endgame = packed && !hybrid && !c->packed_force && c->pipeline && c->tail_bound && many_chains && c->pk_handover != 0 &&
                         c->d_chain_work && c->launch_order_on && N >= 32;
    if (packed || hybrid) {
        // LDS: as many suspended levels as the launch's occupancy leaves room for (the kernel runs one wave per SIMD, four per CU;
        // a launch of few waves — one GPU's share of 4096 30-dim chains is 512 — has half of the CU's 160 KB to itself)
        // coordinates per lane: two while that still leaves every wave a SIMD of its own (or when the row needs no more: D <= 32 is
        // 16 lanes × 2), four beyond (D > 32 always: 16 lanes × 4)
        int cpl = D > 32 ? 4 : 2;
        if (cpl == 2) {
            const int L2 = pk::lanes_per_chain(D, 2);
            const long long waves2 = ((long long)C + 64 / L2 - 1) / (64 / L2);
            if (waves2 > 4LL * c->num_cus && pk::lanes_per_chain(D, 4) < L2) cpl = 4;
        }
        if (c->pk_cpl && D <= 32) cpl = c->pk_cpl;
        const int L = pk::lanes_per_chain(D, cpl), gpw = 64 / L;
        const long long waves = ((long long)C + gpw - 1) / gpw;
        const long long wpc = std::min<long long>(4, std::max<long long>(1, (waves + c->num_cus - 1) / c->num_cus));
        const size_t budget = std::min<size_t>(pk::kMaxLdsPerWave, (size_t)160 * 1024 / (size_t)wpc);
        const size_t fixed = pk::lds_bytes_per_wave(L, cpl, P.max_depth, 0);
        int levels = budget > fixed ? (int)((budget - fixed) / pk::lds_bytes_per_level(cpl)) : 0;
        if (c->pk_lds_levels >= 0) levels = c->pk_lds_levels;
        levels = std::max(0, std::min(levels, std::max(0, P.max_depth - 1)));
        while (levels > 0 && pk::lds_bytes_per_wave(L, cpl, P.max_depth, levels) > pk::kMaxLdsPerWave) levels -= 1;
        P.pk_cpl = cpl;
        P.pk_lds_levels = levels;
        // the gate: chains whose transitions start on trips ≡ 0 mod A run the merges below level log2 A on the same trips (a wave pays
        // for a merge level when any of its chains is at it), and wait A/2 trips per transition for it: worth 16 when the trees
        // are large (32768 funnel chains with the depth limit at 5: 2.29e9 leapfrogs/s at A = 16, 1.84e9 at 4, 1.29e9 at 1 —
        // profiles/r05_packed_queue_rounds.txt), 4 when they have a dozen leaves.  From the previous call's mean tree size.
        P.pk_align = c->pk_align > 0 ? c->pk_align : c->mean_leapfrogs_per_transition >= 48.0 ? 16 : c->mean_leapfrogs_per_transition >= 24.0 ? 8 : 4;
        // the queue of places (packed_kernels.hpp launch_run_packed): as many waves as the GPU holds at once — one per SIMD
        P.pk_queue = c->pk_queue ? reinterpret_cast<unsigned*>(c->d_counter + 1) : nullptr;
        P.pk_max_waves = c->pk_max_waves > 0 ? c->pk_max_waves : 4 * c->num_cus;
    }
    if (c->win_n >= 0) {       // an open metric window: every transition's draw joins the running moments (capi_metric.hip)
        P.win_mean = c->d_win; P.win_m2 = c->d_win + (size_t)C * c->Dpad; P.win_n0 = c->win_n;
    }
    if (da) {
        P.adapt = 1; P.da_init = da->init; P.da_finalize = da->finalize; P.t0 = da->t0;
        P.delta = da->delta; P.gamma = da->gamma; P.kappa = da->kappa;
    }
    // Outputs: device pointers pass through.  Host pointers are served from the context's persistent staging buffers
    // (grown on demand: no hipMalloc / hipFree per call).  The one-kernel engine (diagonal metric) runs a call with host
    // outputs in CHUNKS of L transitions, two staging buffers deep: chunk k leaves over the copy stream (strided 2-D
    // copies into the caller's [C][N][…] arrays; truly asynchronous when those are page-locked — dhmc_host_alloc) while
    // chunk k+1 computes.  The chunks are the same transitions of the same kernel as one launch would run: same bits.
    struct Field { void** dev; void* host; size_t elem; int idx; };   // elem: bytes of one (chain, transition) record
    std::vector<Field> staged;
    const bool one_kernel = c->cfg.metric == DHMC_METRIC_DIAG && !c->logistic_rounds && !c->external;
    const bool host_out = out && !out->on_device &&
                          (out->draws || out->logdensities || out->eps || out->pi || out->acceptance_rate || out->steps ||
                           out->term_left || out->term_right || out->depth || out->directions);
    int64_t L = N;
    if (host_out && one_kernel) {
        const int64_t per_transition = (int64_t)C * D * (int64_t)sizeof(double);
        // default: ≈ 1 GiB of draws per chunk, but at least four chunks per call so that most of the copy runs under a kernel
        L = c->host_chunk > 0 ? c->host_chunk : std::min(((int64_t)1 << 30) / (per_transition > 0 ? per_transition : 1), (N + 3) / 4);
        if (L < 1) L = 1;
        if (L > N) L = N;
    }
    const int nbuf = L < N ? 2 : 1;
    auto bind = [&](void* user, void** slot, size_t elem, int idx) -> int {
        *slot = nullptr;
        if (!user) return DHMC_OK;
        if (out->on_device) { *slot = user; return DHMC_OK; }
        const size_t need = (size_t)C * (size_t)L * elem;
        for (int b = 0; b < nbuf; ++b) {
            auto& sb = c->stage[b][idx];
            if (sb.cap < need) {
                if (sb.p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(sb.p)); sb.p = nullptr; sb.cap = 0; }
                HIP_TRY(c, hipMalloc(&sb.p, need));
                sb.cap = need;
            }
        }
        *slot = c->stage[0][idx].p;
        staged.push_back({slot, user, elem, idx});
        return DHMC_OK;
    };
    int rc = DHMC_OK;
    if (out) {
        if (!rc) rc = bind(out->draws, (void**)&P.out.draws, D * sizeof(double), 0);
        if (!rc) rc = bind(out->logdensities, (void**)&P.out.logdensities, sizeof(double), 1);
        if (!rc) rc = bind(out->eps, (void**)&P.out.eps, sizeof(double), 2);
        if (!rc) rc = bind(out->pi, (void**)&P.out.pi, sizeof(double), 3);
        if (!rc) rc = bind(out->acceptance_rate, (void**)&P.out.acceptance_rate, sizeof(double), 4);
        if (!rc) rc = bind(out->steps, (void**)&P.out.steps, sizeof(int64_t), 5);
        if (!rc) rc = bind(out->term_left, (void**)&P.out.term_left, sizeof(int64_t), 6);
        if (!rc) rc = bind(out->term_right, (void**)&P.out.term_right, sizeof(int64_t), 7);
        if (!rc) rc = bind(out->depth, (void**)&P.out.depth, sizeof(int32_t), 8);
        if (!rc) rc = bind(out->directions, (void**)&P.out.directions, sizeof(uint32_t), 9);
    }
    auto cleanup = [&]() {};   // (the staging buffers belong to the context)
    if (rc) { cleanup(); return rc; }
    if (!staged.empty() && !c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(c, hipEventCreate(&c->ev_k0[b]));
            HIP_TRY(c, hipEventCreate(&c->ev_k1[b]));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copy[b], hipEventDisableTiming));
        }
    }
    // transitions [n0, n0 + len) of every staged field: staging buffer b (record stride L) -> the caller's arrays (stride N)
    auto d2h = [&](int b, int64_t n0, int64_t len, hipStream_t s) -> hipError_t {
        for (auto& f : staged) {
            char* dst = (char*)f.host + (size_t)n0 * f.elem;
            hipError_t ce = hipMemcpy2DAsync(dst, (size_t)N * f.elem, c->stage[b][f.idx].p, (size_t)L * f.elem, (size_t)len * f.elem,
                                             (size_t)C, hipMemcpyDeviceToHost, s);
            if (ce != hipSuccess) return ce;
        }
        return hipSuccess;
    };
    double chunk_ms = 0.0;
    bool hybrid_ran = false;
    // The per-draw kernels walk all transitions of a chain in one wave (group, pipeline), so a launch ends with its slowest chain: a
    // chain whose trees are persistently deeper (a smaller adapted ϵ) and which starts in the last wave of workgroups holds the whole
    // launch open — measured on BASELINE configs[1]: one chain of 4096 at 1.48 × the mean work, 189 ms instead of 171 ms per 1000
    // transitions.  The next launch therefore starts its chains in the order of this one's work, longest first (results do not
    // depend on the order); and the shape of the work decides the next launch's engine (tail_bound, tail_count: above).
    auto refresh_order = [&](int64_t n_launch) -> hipError_t {
        c->h_chain_work.resize(C);
        hipError_t he = hipMemcpyAsync(c->h_chain_work.data(), c->d_chain_work, sizeof(unsigned) * C, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) return he;
        unsigned long long sum = 0;
        unsigned mx = 0;
        for (unsigned w : c->h_chain_work) { sum += w; mx = std::max(mx, w); }
        c->launch_order_valid = false;
        c->tail_count = 0;
        if ((double)mx * C > 1.03 * (double)sum) {
            c->h_launch_order.resize(C);
            for (int i = 0; i < C; ++i) c->h_launch_order[i] = i;
            std::stable_sort(c->h_launch_order.begin(), c->h_launch_order.end(),
                             [&](int a, int b) { return c->h_chain_work[a] > c->h_chain_work[b]; });
            he = hipMemcpyAsync(c->d_launch_order, c->h_launch_order.data(), sizeof(int) * C, hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            c->launch_order_valid = he == hipSuccess;
            if (c->launch_order_valid) {      // the head of the order whose work was more than 3 × the median's, at most what the pipeline kernel keeps resident
                std::vector<unsigned> w(c->h_chain_work);
                std::nth_element(w.begin(), w.begin() + C / 2, w.end());
                const double med = (double)w[C / 2];
                const int cap = std::min(C / 8, 5 * c->num_cus);
                int k = 0;
                while (k < cap && (double)c->h_chain_work[c->h_launch_order[k]] > 3.0 * med) ++k;
                c->tail_count = k;
            }
        }
        c->tail_bound = (double)mx * C > 3.0 * (double)sum;
        if (std::getenv("DHMC_DEBUG_ORDER"))
            std::fprintf(stderr, "[dhmc] launch order: N=%lld max=%u mean=%.1f valid=%d first=%d tail_bound=%d tail_count=%d\n", (long long)n_launch, mx,
                         (double)sum / C, (int)c->launch_order_valid, c->launch_order_valid ? c->h_launch_order[0] : -1, (int)c->tail_bound, c->tail_count);
        return he;
    };

    hipError_t e = hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
    if (e == hipSuccess && c->external && c->cfg.metric == DHMC_METRIC_DENSE) {
        // dense round engine (dense_rounds.hpp) with the host's callback as the density, one batch on one stream
        P.one_product = c->dense_products == 1;
        RoundArgs ra{P, c->rb};
        const int ld = c->Dpad;
        const RoundBuffers& R = c->rb;
        e = hipMemsetAsync(R.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_gemm_rows(R.cp, c->d_WT, R.tbuf, ld, C, R.list, R.list_count, c->stream);       // p₀ = z·Wᵀ
                launch_gemm_rows(R.tbuf, c->d_Minv, R.cps, ld, C, R.list, R.list_count, c->stream);    // p♯₀
                if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, R.list, R.list_count, c->stream);   // u₀ = ∇ℓq₀·M⁻¹
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(R.list_count, 0, sizeof(int), c->stream);
                if (!P.one_product) launch_gemm_rows(R.cp, c->d_Minv, R.tbuf, ld, C, nullptr, nullptr, c->stream);   // M⁻¹pₘ
                DHMC_EXT_NPL(rounds_k2a_dense_external_kernel, dim3(C), ra.P, ra.R)                    // q′ (one product: M⁻¹pₘ = p♯ + (ϵ/2)u)
                if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
                rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
                if (rc) { c->poisoned = true; cleanup(); return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
                DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′
                if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, nullptr, nullptr, c->stream);   // u′ = ∇ℓq′·M⁻¹
                else launch_gemm_rows(R.cp, c->d_Minv, R.cps, ld, C, nullptr, nullptr, c->stream);     // p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&done, R.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->external) {
        // round engine with the host's callback as the gradient (external_rounds.hpp)
        RoundArgs ra{P, c->rb};
        e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_logistic_op(0, c->NPL, ra, c->lr, c->stream);                                   // p = W∘z, p♯
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
                launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
                if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
                rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
                if (rc) { c->poisoned = true; cleanup(); return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
                DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′, p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&done, c->rb.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->logistic_rounds) {
        // GEMM-gradient round engine (logistic_rounds.hpp)
        RoundArgs ra{P, c->rb};
        const int ld = c->Dpad;
        const int npad = (int)c->tp.npad;
        e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        // the done-counter is read through page-locked memory one batch of four rounds behind (as in the dense engine below): the
        // host never drains the stream inside the loop; the rounds enqueued after the last chain finished find no chain in a leaf
        // phase and an empty row list
        if (!c->h_done && e == hipSuccess) {
            e = hipHostMalloc((void**)&c->h_done, 2 * 8 * sizeof(int), hipHostMallocDefault);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming);
        }
        long long batch = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_logistic_op(0, c->NPL, ra, c->lr, c->stream);                                   // p = W∘z, p♯
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
                launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
                launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);                                   // the rows of this round
                launch_gemm_list(c->st.q, ld, c->tp.b, npad, c->lr.H, npad, C, ld, npad, c->lr.act, c->lr.act_count, c->stream);   // η = Q′·Xᵀ
                launch_logistic_op(2, c->NPL, ra, c->lr, c->stream);                                   // r, S₁
                launch_gemm_splitk(c->lr.H, npad, c->tp.a, ld, c->lr.P, ld, (size_t)C * ld, C, npad, ld, DHMC_LOGISTIC_BLOCK,
                                   c->lr.act, c->lr.act_count, c->stream);                             // Xᵀr = R·X, block by block
                launch_logistic_op(3, c->NPL, ra, c->lr, c->stream);                                   // ∇ℓ, ℓ, p′, p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            int* slot = c->h_done + 8 * (batch & 1);
            if (e == hipSuccess) e = hipMemcpyAsync(slot, c->rb.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_done[batch & 1], c->stream);
            if (batch >= 1 && e == hipSuccess) {
                e = hipEventSynchronize(c->ev_done[(batch - 1) & 1]);
                done = c->h_done[8 * ((batch - 1) & 1)];
            }
            batch += 1;
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->cfg.metric == DHMC_METRIC_DENSE && c->dense_rounds) {
        // (both dense engines run either recurrence with the same bits, so which one serves a context is a matter of speed only)
        P.one_product = c->dense_products == 1;
        P.fuse_k2 = c->fuse_k2;
        // Round-based dense engine (dense_rounds.hpp): every round is one leapfrog for every chain.  The chains
        // run as two half-batches on two streams so that one half's HBM-bound tree kernel overlaps the other
        // half's MFMA-bound contractions.
        const int ld = c->Dpad;
        const int nh = (C >= 256 && C % c->dense_parts == 0) ? c->dense_parts : 1;
        struct Half { RoundArgs ra; hipStream_t s; int base, count; } H[4];
        c->streams[0] = c->stream; c->streams[1] = c->stream2;
        for (int h = 0; h < nh; ++h) {
            H[h].base = h * (C / nh);
            H[h].count = (h == nh - 1) ? C - H[h].base : C / nh;
            c->rbp[h] = c->rb;
            c->rbp[h].list = c->rb.list + H[h].base;
            c->rbp[h].list_count = c->rb.list_count + 2 * h;
            c->rbp[h].done_count = c->rb.list_count + 2 * h + 1;
            H[h].ra = RoundArgs{P, c->rbp[h]};
            H[h].ra.P.chain_base = H[h].base;
            H[h].ra.P.C = H[h].count;
            H[h].s = c->streams[h];
        }
        e = hipMemsetAsync(c->rb.list_count, 0, 8 * sizeof(int), c->stream);
        if (nh >= 2 && e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
        for (int h = 1; h < nh && e == hipSuccess; ++h) e = hipStreamWaitEvent(c->streams[h], c->ev_fork, 0);
        for (int h = 0; h < nh && e == hipSuccess; ++h)
            if ((rc = dispatch(c, Op::RoundStart, &H[h].ra, H[h].s, true))) { cleanup(); return rc; }
        unsigned long long rounds = 0;
        int done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // Four rounds of both half-batches = 64 launches on two streams, the same for the whole call.  DHMC_GRAPH=1
        // captures them ONCE into a graph (fork to stream2, join back) and launches the graph until every chain is done:
        // one host call per four rounds instead of 64.  Measured on config 3: 8.05e6 leapfrog-steps/s against 8.84e6 with
        // plain launches — the host was never the limit, and the graph's branches overlap less than two free-running
        // streams do — so it is off by default.
        constexpr int REPS = 4;
        // Once the first chains have finished their transitions (the host sees the done-counters every REPS rounds), the
        // two products of a round are taken over the rows of the chains still running only (a row list per part, rebuilt
        // every round): a call ends when its slowest chain does, and until then every round multiplied all rows.  While
        // every chain is running — BASELINE config 3's equal trees from start to end — nothing changes.
        bool row_lists = false;
        auto enqueue_reps = [&]() -> int {
            for (int rep = 0; rep < REPS && e == hipSuccess; ++rep) {
                for (int h = 0; h < nh && e == hipSuccess; ++h) {
                    const RoundBuffers& R = H[h].ra.R;
                    const size_t off = (size_t)H[h].base * ld;
                    hipStream_t s = H[h].s;
                    launch_gemm_rows(R.cp, c->d_WT, R.tbuf, ld, H[h].count, R.list, R.list_count, s);               // p₀ = z·Wᵀ
                    launch_gemm_rows(R.tbuf, c->d_Minv, R.cps, ld, H[h].count, R.list, R.list_count, s);            // p♯₀
                    if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, H[h].count, R.list, R.list_count, s);   // u₀ = ∇ℓq₀·M⁻¹
                    if (int r = dispatch(c, Op::RoundK0, &H[h].ra, s, true)) return r;
                    e = hipMemsetAsync(R.list_count, 0, sizeof(int), s);
                    if (P.one_product && !row_lists) {
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;                          // M⁻¹pₘ = p♯ + (ϵ/2)u
                        launch_gemm_rows(c->st.g + off, c->d_Minv, R.cu + off, ld, H[h].count, nullptr, nullptr, s); // u′ = ∇ℓq′·M⁻¹
                    } else if (P.one_product) {
                        LogisticRound L = c->lr;
                        L.act = c->lr.act + H[h].base;
                        L.act_count = c->lr.act + C + h;
                        if (e == hipSuccess) e = hipMemsetAsync(L.act_count, 0, sizeof(int), s);
                        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((H[h].count + 255) / 256), dim3(256), 0, s, H[h].ra.P, R, L);
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, H[h].count, L.act, L.act_count, s);          // u′
                    } else if (!row_lists) {
                        launch_gemm_rows(R.cp + off, c->d_Minv, R.tbuf + off, ld, H[h].count, nullptr, nullptr, s); // M⁻¹pₘ
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(R.cp + off, c->d_Minv, R.cps + off, ld, H[h].count, nullptr, nullptr, s);  // p♯
                    } else {
                        LogisticRound L = c->lr;
                        L.act = c->lr.act + H[h].base;
                        L.act_count = c->lr.act + C + h;
                        if (e == hipSuccess) e = hipMemsetAsync(L.act_count, 0, sizeof(int), s);
                        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((H[h].count + 255) / 256), dim3(256), 0, s, H[h].ra.P, R, L);
                        launch_gemm_rows(R.cp, c->d_Minv, R.tbuf, ld, H[h].count, L.act, L.act_count, s);           // M⁻¹pₘ
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(R.cp, c->d_Minv, R.cps, ld, H[h].count, L.act, L.act_count, s);            // p♯
                    }
                    if (int r = dispatch(c, Op::RoundK3, &H[h].ra, s, true)) return r;
                }
            }
            return DHMC_OK;
        };
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        // (not on the legacy default stream, which cannot be captured: hosts that hand over stream 0 get plain launches)
        if (c->use_graph && c->stream != nullptr && e == hipSuccess &&
            hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            if (nh >= 2 && e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
            for (int h = 1; h < nh && e == hipSuccess; ++h) e = hipStreamWaitEvent(c->streams[h], c->ev_fork, 0);
            if (e == hipSuccess) rc = enqueue_reps();
            for (int h = 1; h < nh && e == hipSuccess; ++h) {
                e = hipEventRecord(c->ev_joins[h], c->streams[h]);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_joins[h], 0);
            }
            hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess && !rc) e = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
            if (rc || e != hipSuccess) {
                if (gexec) (void)hipGraphExecDestroy(gexec);
                if (graph) (void)hipGraphDestroy(graph);
                cleanup();
                if (rc) return rc;
                c->err = std::string("dhmc_run (graph capture): ") + hipGetErrorString(e);
                return DHMC_ERR_HIP;
            }
        }
        // The host never drains the streams to look at the done-counters: after every batch of REPS rounds they are copied
        // into page-locked memory behind an event, and the host reads the PREVIOUS batch's copy once the next batch is
        // enqueued.  So it runs one batch ahead; the (at most REPS) rounds enqueued after the last chain finished find no
        // chain in a leaf phase and, with the row lists, no rows to multiply.
        if (!c->h_done && e == hipSuccess) {
            e = hipHostMalloc((void**)&c->h_done, 2 * 8 * sizeof(int), hipHostMallocDefault);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming);
        }
        long long batch = 0;
        while (e == hipSuccess && done[1] + done[3] + done[5] + done[7] < C) {
            if (gexec) {
                e = hipGraphLaunch(gexec, c->stream);
            } else {
                if ((rc = enqueue_reps())) { cleanup(); return rc; }
                if (e == hipSuccess) e = hipGetLastError();
                for (int h = 1; h < nh && e == hipSuccess; ++h) {
                    e = hipEventRecord(c->ev_joins[h], c->streams[h]);
                    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_joins[h], 0);
                }
            }
            rounds += REPS;
            int* slot = c->h_done + 8 * (batch & 1);
            if (e == hipSuccess) e = hipMemcpyAsync(slot, c->rb.list_count, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_done[batch & 1], c->stream);
            if (batch >= 1 && e == hipSuccess) {
                e = hipEventSynchronize(c->ev_done[(batch - 1) & 1]);
                std::memcpy(done, c->h_done + 8 * ((batch - 1) & 1), 8 * sizeof(int));
            }
            batch += 1;
            row_lists = c->dense_row_lists && !gexec && done[1] + done[3] + done[5] + done[7] > 0;
        }
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        c->last_rounds = rounds;
    } else if (e == hipSuccess && nbuf == 2) {
        // the one-kernel engine, host outputs, in chunks: kernel of chunk k ‖ copy of chunk k-1
        const int64_t nchunks = (N + L - 1) / L;
        bool used[2] = {false, false};
        for (int64_t k = 0; k < nchunks && e == hipSuccess; ++k) {
            const int b = (int)(k & 1);
            const int64_t n0 = k * L, len = (n0 + L <= N) ? L : N - n0;
            RunParams Q = P;
            Q.N = len;
            Q.out_stride = L;
            Q.win_n0 = P.win_n0 + n0;
            if (da) { Q.da_init = (k == 0) ? da->init : 0; Q.da_finalize = (k == nchunks - 1) ? da->finalize : 0; }
            for (auto& f : staged) *f.dev = c->stage[b][f.idx].p;          // (the slots are fields of P.out: copy them again)
            Q.out = P.out;
            if (used[b]) {                                                   // buffer b: its previous copy has left, and its kernel time is known
                e = hipStreamWaitEvent(c->stream, c->ev_copy[b], 0);
                if (e == hipSuccess) e = hipEventSynchronize(c->ev_k1[b]);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_k0[b], c->ev_k1[b]);
                chunk_ms += ms;
            }
            if (e == hipSuccess) e = hipEventRecord(c->ev_k0[b], c->stream);
            if (e == hipSuccess && (rc = dispatch(c, run_op, &Q))) { (void)hipDeviceSynchronize(); cleanup(); return rc; }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(c->ev_k1[b], c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_k1[b], 0);
            if (e == hipSuccess) e = d2h(b, n0, len, c->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_copy[b], c->copy_stream);
            used[b] = true;
        }
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            if (used[b]) {
                e = hipEventSynchronize(c->ev_k1[b]);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_k0[b], c->ev_k1[b]);
                chunk_ms += ms;
            }
        if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
    } else if (e == hipSuccess && hybrid && nbuf == 1) {
        // The call in ROUNDS of N / hybrid_segments transitions (RunParams::prog).  In every round the chains run packed, in the
        // order of their work in the round before, through the queue of places — and a chain that has taken more than the round's
        // budget of leapfrog steps is given up at its next transition boundary (pk_budget) and, with the chains that were deep in
        // the round before, continues through the pipeline kernel, whose blocks are launched first on a second stream: the bulk at
        // the packed kernel's throughput, the deep chains at the pipeline's latency, and no chain has to be predicted deep before
        // it is.  Budget: the work after which a chain alone would hold the packed launch open — (chains per lane group) × the
        // mean work of a chain in a round.
        hybrid_ran = true;
        if (!c->stream2) {
            e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
            if (e == hipSuccess && !c->ev_fork) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
            if (e == hipSuccess && !c->ev_join) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
        }
        // The two kernels on disjoint sets of CUs (DHMC_HYBRID_DEEP_CUS of them for the pipeline blocks): a block of four waves that
        // has to find room between the packed kernel's waves starts when those drain, and one that shares its SIMDs with them is
        // no longer the kernel with the lowest latency.
        if (e == hipSuccess && c->hybrid_deep_cus > 0 && !c->hybrid_deep_wave && !c->stream_bulk) {      // (the wave kernel as the deep engine sits beside the packed waves)
            const int words = (c->num_cus + 31) / 32;
            std::vector<uint32_t> deep_mask(words, 0u), bulk_mask(words, 0u);
            const int deep_cus = std::min(c->hybrid_deep_cus, c->num_cus / 2);
            int taken = 0;
            for (int i = 0; i < c->num_cus; ++i) {     // every (num_cus / deep_cus)-th CU: spread over the XCDs and shader engines
                const bool deep = (long long)(i + 1) * deep_cus / c->num_cus > (long long)i * deep_cus / c->num_cus;
                (deep ? deep_mask : bulk_mask)[i / 32] |= 1u << (i % 32);
                taken += deep;
            }
            hipStream_t sd = nullptr, sb = nullptr;
            if (hipExtStreamCreateWithCUMask(&sd, (uint32_t)words, deep_mask.data()) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&sb, (uint32_t)words, bulk_mask.data()) == hipSuccess) {
                c->stream_deep = sd; c->stream_bulk = sb; c->deep_cus = taken;
            } else {
                (void)hipGetLastError();
                if (sd) (void)hipStreamDestroy(sd);
                c->hybrid_deep_cus = 0;            // no masks on this runtime: the kernels share the GPU
            }
        }
        if (e == hipSuccess && !c->ev_round[0]) for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev_round[i]);
        if (e == hipSuccess && !c->ev_join2) e = hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming);
        hipStream_t const s_deep = c->stream_deep ? c->stream_deep : c->stream2, s_bulk = c->stream_bulk ? c->stream_bulk : c->stream;
        const int bulk_cus = c->stream_bulk ? c->num_cus - c->deep_cus : c->num_cus;
        if (e == hipSuccess && !c->d_prog) {
            if ((rc = dev_alloc(c, &c->d_prog, (size_t)C)) || (rc = dev_alloc(c, &c->d_list_packed, (size_t)C)) ||
                (rc = dev_alloc(c, &c->d_list_deep, (size_t)C)) || (rc = dev_alloc(c, &c->d_evicted, (size_t)C))) { cleanup(); return rc; }
        }
        const int S = c->hybrid_segments;
        const int64_t seg = (N + S - 1) / S;
        const int Lp = pk::lanes_per_chain(D, P.pk_cpl);
        const int bulk_waves = c->pk_max_waves > 0 ? c->pk_max_waves : 4 * bulk_cus;
        const double groups = (double)bulk_waves * (64 / Lp);
        const double factor = c->hybrid_budget > 0.0 ? c->hybrid_budget : std::max(2.0, (double)C / groups);
        const size_t deep_cap = (size_t)std::max(1, c->hybrid_deep_cap) * (size_t)(c->stream_deep ? c->deep_cus : c->num_cus);
        unsigned* const d_evict_count = reinterpret_cast<unsigned*>(c->d_counter + 2);
        std::vector<int> list_packed(C), list_deep, evicted;
        if (c->launch_order_valid) list_packed = c->h_launch_order;
        else for (int i = 0; i < C; ++i) list_packed[i] = i;
        std::vector<unsigned> work_before(C, 0u), work_round(C, 0u);
        std::vector<int> prog_now(C, 0), prog_before(C, 0);
        std::vector<float> rate(C, 0.f);
        double mean_tr = c->mean_leapfrogs_per_transition;        // of the previous call (0: unknown — the first round has no budget)
        if (e == hipSuccess) e = hipMemsetAsync(c->d_prog, 0, sizeof(int) * C, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->d_chain_work, 0, sizeof(unsigned) * C, c->stream);
        RunParams Q = P;
        Q.out_stride = P.out_stride ? P.out_stride : N;
        Q.prog = c->d_prog;
        Q.chain_work = c->d_chain_work;
        // one pair of launches: the deep list through the pipeline kernel (first: its blocks take their CUs before the packed grid
        // fills the GPU), the packed list beside it
        bool ran_deep = false, ran_bulk = false;
        auto launch_pair = [&](int64_t target, unsigned long long budget) -> int {
            Q.N = target;
            if (da) { Q.da_init = da->init; Q.da_finalize = target >= N ? da->finalize : 0; }
            ran_deep = !list_deep.empty(); ran_bulk = !list_packed.empty();
            if (ran_deep) e = hipMemcpyAsync(c->d_list_deep, list_deep.data(), sizeof(int) * list_deep.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && ran_bulk) e = hipMemcpyAsync(c->d_list_packed, list_packed.data(), sizeof(int) * list_packed.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && ran_bulk) e = hipMemsetAsync(d_evict_count, 0, sizeof(unsigned), c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
            if (e != hipSuccess) return 0;
            if (ran_deep) {
                e = hipStreamWaitEvent(s_deep, c->ev_fork, 0);
                if (e == hipSuccess) e = hipEventRecord(c->ev_round[0], s_deep);
                if (e != hipSuccess) return 0;
                RunParams T = Q;
                T.C = (int)list_deep.size();
                T.launch_order = c->d_list_deep;
                if (int r = dispatch(c, c->hybrid_deep_wave ? Op::Run : Op::RunPipeline, &T, s_deep, true)) return r;      // (the wave kernel: a wave per chain, resident beside the packed waves)
                e = hipEventRecord(c->ev_round[1], s_deep);
                if (e == hipSuccess) e = hipEventRecord(c->ev_join, s_deep);
                if (e != hipSuccess) return 0;
            }
            if (ran_bulk) {
                if (s_bulk != c->stream) e = hipStreamWaitEvent(s_bulk, c->ev_fork, 0);
                if (e == hipSuccess) e = hipEventRecord(c->ev_round[2], s_bulk);
                if (e != hipSuccess) return 0;
                RunParams B = Q;
                B.C = (int)list_packed.size();
                B.launch_order = c->d_list_packed;
                B.pk_order_base = 0;
                B.pk_max_waves = bulk_waves;
                B.pk_budget = budget;
                B.pk_evicted = c->d_evicted;
                B.pk_evict_count = d_evict_count;
                if (int r = dispatch(c, Op::RunPacked, &B, s_bulk, true)) return r;
                e = hipEventRecord(c->ev_round[3], s_bulk);
                if (e == hipSuccess && s_bulk != c->stream) {
                    e = hipEventRecord(c->ev_join2, s_bulk);
                    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join2, 0);
                }
                if (e != hipSuccess) return 0;
            }
            if (ran_deep) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
            if (e == hipSuccess) e = hipGetLastError();
            return 0;
        };
        int64_t done_to = 0;
        for (int round = 0; done_to < N && e == hipSuccess; ++round) {
            const int64_t target = std::min(N, done_to + seg), len = target - done_to;
            const unsigned long long budget = mean_tr > 0.0 ? (unsigned long long)std::max(64.0, factor * mean_tr * (double)len) : 0ull;
            if ((rc = launch_pair(target, budget))) { (void)hipDeviceSynchronize(); cleanup(); return rc; }
            // the round's work per chain and the chains the packed launch gave up: the next round's two lists
            unsigned n_evicted = 0;
            c->h_chain_work.resize(C);
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_chain_work.data(), c->d_chain_work, sizeof(unsigned) * C, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(prog_now.data(), c->d_prog, sizeof(int) * C, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && !list_packed.empty()) e = hipMemcpyAsync(&n_evicted, d_evict_count, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) break;
            evicted.resize(n_evicted);
            if (n_evicted) e = hipMemcpy(evicted.data(), c->d_evicted, sizeof(int) * n_evicted, hipMemcpyDeviceToHost);
            if (e != hipSuccess) break;
            unsigned long long sum = 0;
            for (int i = 0; i < C; ++i) { work_round[i] = c->h_chain_work[i] - work_before[i]; work_before[i] = c->h_chain_work[i]; sum += work_round[i]; }
            mean_tr = (double)sum / ((double)C * (double)len);
            // the next round's lists: by the chains' leapfrog steps per transition in this round (a chain that was given up stopped
            // early, one that was behind did more than the round's transitions), the deepest go through the pipeline kernel — as many
            // as its CUs hold at once (× hybrid_deep_cap), if they are well above the mean — and the others packed, deepest first
            for (int i = 0; i < C; ++i) {
                const int did = prog_now[i] - prog_before[i];
                rate[i] = (float)((double)work_round[i] / (double)(did > 0 ? did : 1));
                prog_before[i] = prog_now[i];
            }
            list_deep.clear();
            list_packed.resize(C);
            for (int i = 0; i < C; ++i) list_packed[i] = i;
            std::stable_sort(list_packed.begin(), list_packed.end(), [&](int a, int b) { return rate[a] > rate[b]; });
            if (target < N) {
                size_t k = 0;
                while (k < deep_cap && k < (size_t)C && (double)rate[list_packed[k]] > c->hybrid_promote * mean_tr) ++k;
                list_deep.assign(list_packed.begin(), list_packed.begin() + k);
                list_packed.erase(list_packed.begin(), list_packed.begin() + k);
            } else {
                list_packed.clear();
                list_deep = evicted;        // the clean-up: only the chains that are behind are left
                std::stable_sort(list_deep.begin(), list_deep.end(), [&](int a, int b) { return rate[a] > rate[b]; });
            }
            if (std::getenv("DHMC_DEBUG_ORDER")) {
                float ms_deep = 0.f, ms_bulk = 0.f;
                if (ran_deep) (void)hipEventElapsedTime(&ms_deep, c->ev_round[0], c->ev_round[1]);
                if (ran_bulk) (void)hipEventElapsedTime(&ms_bulk, c->ev_round[2], c->ev_round[3]);
                std::fprintf(stderr, "[dhmc] round %d to %lld: budget %llu, given up %u, next deep %zu, mean per transition %.1f; pipeline %.1f ms, packed %.1f ms\n",
                             round, (long long)target, budget, n_evicted, list_deep.size(), mean_tr, ms_deep, ms_bulk);
            }
            done_to = target;
        }
        // the chains the last round gave up: to the end through the pipeline kernel
        if (e == hipSuccess && !list_deep.empty()) {
            if ((rc = launch_pair(N, 0ull))) { (void)hipDeviceSynchronize(); cleanup(); return rc; }
        }
    } else if (e == hipSuccess && endgame && nbuf == 1) {
        if (!c->d_prog) {
            if ((rc = dev_alloc(c, &c->d_prog, (size_t)C)) || (rc = dev_alloc(c, &c->d_list_packed, (size_t)C)) ||
                (rc = dev_alloc(c, &c->d_list_deep, (size_t)C)) || (rc = dev_alloc(c, &c->d_evicted, (size_t)C))) { cleanup(); return rc; }
        }
        unsigned* const d_evict_count = reinterpret_cast<unsigned*>(c->d_counter + 2);
        e = hipMemsetAsync(c->d_prog, 0, sizeof(int) * C, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->d_chain_work, 0, sizeof(unsigned) * C, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_evict_count, 0, sizeof(unsigned), c->stream);
        RunParams B = P;
        B.prog = c->d_prog;
        B.pk_evicted = c->d_evicted;
        B.pk_evict_count = d_evict_count;
        B.pk_live = reinterpret_cast<unsigned*>(c->d_counter + 3);
        // (what the pipeline kernel keeps resident, twice that below 64 chains per CU: 8192 chains 5.9e8 against 5.6e8, 32768 1.43e9 against 1.49e9)
        B.pk_handover_below = c->pk_handover > 0 ? c->pk_handover : (C >= 64 * c->num_cus ? 5 : 10) * c->num_cus;
        if (e == hipSuccess && (rc = dispatch(c, Op::RunPacked, &B))) { cleanup(); return rc; }
        unsigned n_given_up = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&n_given_up, d_evict_count, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && n_given_up > 0) {          // the chains the packed launch gave up, in the order it gave them up
            RunParams T = B;
            T.C = (int)n_given_up;
            T.launch_order = c->d_evicted;
            T.pk_live = nullptr; T.pk_handover_below = 0;
            if ((rc = dispatch(c, Op::RunPipeline, &T))) { cleanup(); return rc; }
        }
        if (std::getenv("DHMC_DEBUG_ORDER")) std::fprintf(stderr, "[dhmc] end game: %u chains handed to the pipeline kernel\n", n_given_up);
        if (e == hipSuccess) e = hipGetLastError();
    } else if (e == hipSuccess) {
        rc = dispatch(c, run_op, &P);
        if (rc) { cleanup(); return rc; }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess && nbuf == 1 && !staged.empty()) e = d2h(0, 0, N, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&c->last_leapfrogs, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    (void)hybrid_ran;                                                // (a call in rounds leaves the whole call's work per chain in chain_work)
    const bool reorder = P.chain_work && N >= 32;                    // (a short call's counts say little about the chains, and sorting is not free)
    if (e == hipSuccess && reorder) e = refresh_order(N);
    else if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && N > 0) c->mean_leapfrogs_per_transition = (double)c->last_leapfrogs / ((double)C * (double)N);
    if (e == hipSuccess) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->last_ms = nbuf == 2 ? chunk_ms : ms;     // kernel time only: not the waits for copies between the chunks
    }
    cleanup();
    if (e != hipSuccess) {
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);     // no copy into the caller's arrays may outlive the call
        c->err = std::string("dhmc_run: ") + hipGetErrorString(e);
        return DHMC_ERR_HIP;
    }
    // staging of a call that could not be chunked (round engines: [C][N] records at once) is given back when it is large;
    // the chunked engine's two buffers (≤ ≈1 GiB of draws each) stay with the context
    if (nbuf == 1 && !staged.empty()) {
        size_t held = 0;
        for (auto& sb : c->stage[0]) held += sb.cap;
        if (held > ((size_t)1 << 30))
            for (auto& sb : c->stage[0])
                if (sb.p) { (void)hipFree(sb.p); sb.p = nullptr; sb.cap = 0; }
    }
    if (c->win_n >= 0) c->win_n += N;
    return status_code(c);
}
}  // namespace

// The round engines (dense metric, GEMM-shaped gradients, external models) advance their chains round by round, so a call cannot
// hand out its first transitions while it computes the last ones.  With HOST outputs of more than ≈ 2 GiB of draws the call is
// therefore run as several calls of L transitions (the chains resume where they stand: the same transitions, the same bits) into
// two device staging buffers of ≈ 1 GiB, and chunk k leaves over the copy stream while chunk k + 1 computes — what the diagonal
// engine does inside one call (run_call).  Dual averaging: initialised by the first chunk, finalised by the last.
int dhmc_run(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    const bool one_kernel = c->cfg.metric == DHMC_METRIC_DIAG && !c->logistic_rounds && !c->external;
    const bool host_draws = out && !out->on_device && out->draws;
    const int64_t per_transition = (int64_t)c->cfg.chains * c->cfg.dim * (int64_t)sizeof(double);
    int64_t L = N;
    if (host_draws && !one_kernel && N > 1) {
        if (c->host_chunk > 0) L = std::min<int64_t>(c->host_chunk, N);
        else if (per_transition * N > ((int64_t)2 << 30)) L = std::max<int64_t>(1, ((int64_t)1 << 30) / per_transition);
    }
    if (L >= N) return run_call(c, N, da, out);

    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    struct F { void* host; size_t elem; int idx; };
    const F fields[10] = {{out->draws, (size_t)c->cfg.dim * sizeof(double), 0}, {out->logdensities, sizeof(double), 1}, {out->eps, sizeof(double), 2},
                          {out->pi, sizeof(double), 3}, {out->acceptance_rate, sizeof(double), 4}, {out->steps, sizeof(int64_t), 5},
                          {out->term_left, sizeof(int64_t), 6}, {out->term_right, sizeof(int64_t), 7}, {out->depth, sizeof(int32_t), 8},
                          {out->directions, sizeof(uint32_t), 9}};
    const size_t C = (size_t)c->cfg.chains;
    for (const F& f : fields) {
        if (!f.host) continue;
        for (int b = 0; b < 2; ++b) {
            auto& sb = c->stage[b][f.idx];
            const size_t need = C * (size_t)L * f.elem;
            if (sb.cap < need) {
                if (sb.p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(sb.p)); sb.p = nullptr; sb.cap = 0; }
                HIP_TRY(c, hipMalloc(&sb.p, need));
                sb.cap = need;
            }
        }
    }
    if (!c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(c, hipEventCreate(&c->ev_k0[b]));
            HIP_TRY(c, hipEventCreate(&c->ev_k1[b]));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copy[b], hipEventDisableTiming));
        }
    }
    const int64_t nchunks = (N + L - 1) / L;
    double ms = 0.0;
    unsigned long long leapfrogs = 0, rounds = 0;
    int rc = DHMC_OK;
    bool used[2] = {false, false};
    // Every exit drains the copy stream: asynchronous copies into the caller's arrays must not outlive the call (the caller —
    // numpy, Julia's GC — may free them the moment it sees an error code).
    auto chunks = [&]() -> int {
        for (int64_t k = 0; k < nchunks; ++k) {
            const int b = (int)(k & 1);
            const int64_t n0 = k * L, len = std::min(L, N - n0);
            if (used[b]) HIP_TRY(c, hipEventSynchronize(c->ev_copy[b]));            // staging buffer b is free again
            dhmc_outputs dev{};
            dev.on_device = 1;
            void** slots[10] = {(void**)&dev.draws, (void**)&dev.logdensities, (void**)&dev.eps, (void**)&dev.pi, (void**)&dev.acceptance_rate,
                                (void**)&dev.steps, (void**)&dev.term_left, (void**)&dev.term_right, (void**)&dev.depth, (void**)&dev.directions};
            for (const F& f : fields)
                if (f.host) *slots[f.idx] = c->stage[b][f.idx].p;
            dhmc_dual_averaging dk{};
            if (da) { dk = *da; dk.init = (k == 0) ? da->init : 0; dk.finalize = (k == nchunks - 1) ? da->finalize : 0; }
            const int r = run_call(c, len, da ? &dk : nullptr, &dev);              // returns with the stream drained
            ms += c->last_ms; leapfrogs += c->last_leapfrogs; rounds += c->last_rounds;
            if (r != DHMC_OK && r != DHMC_ERR_CHAIN_FAILURE) return r;
            if (r != DHMC_OK) rc = r;                                              // (a failed chain: the call goes on, as one call would)
            for (const F& f : fields) {
                if (!f.host) continue;
                HIP_TRY(c, hipMemcpy2DAsync((char*)f.host + (size_t)n0 * f.elem, (size_t)N * f.elem, c->stage[b][f.idx].p, (size_t)len * f.elem,
                                            (size_t)len * f.elem, C, hipMemcpyDeviceToHost, c->copy_stream));
            }
            HIP_TRY(c, hipEventRecord(c->ev_copy[b], c->copy_stream));
            used[b] = true;
        }
        return DHMC_OK;
    };
    const int lr = chunks();
    const hipError_t se = hipStreamSynchronize(c->copy_stream);
    if (lr != DHMC_OK) return lr;
    HIP_TRY(c, se);
    c->last_ms = ms; c->last_leapfrogs = leapfrogs; c->last_rounds = rounds;
    return rc;
}

double dhmc_last_run_kernel_ms(const dhmc_ctx* c) { return c ? c->last_ms : 0.0; }
uint64_t dhmc_last_run_rounds(const dhmc_ctx* c) { return c ? c->last_rounds : 0; }
uint64_t dhmc_last_run_leapfrogs(const dhmc_ctx* c) { return c ? c->last_leapfrogs : 0; }
uint64_t dhmc_workspace_bytes(const dhmc_ctx* c) { return c ? c->ws_bytes : 0; }

}  // extern "C"
