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
        case Op::RunPacked: {
            if (M || !c->user_packed) return DHMC_ERR_UNSUPPORTED;
            RunParams Q;
            int Lp = 0;
            size_t lds = 0;
            if (int r = packed_launch_prepare(*(const RunParams*)P, cs, &Q, &Lp, &grid, &lds)) return r;
            void* args[] = {&Q};
            return launch(c->user_packed, WAVE, (unsigned)lds, args);
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
    long long ev = 0;
    if (env_list_value("DHMC_DENSE", "k3_block", &ev)) c->k3_block = ev != 0;
    if (const char* e = std::getenv("DHMC_HOST_CHUNK")) c->host_chunk = std::atoll(e);
    if (const char* e = std::getenv("DHMC_LAUNCH_ORDER")) c->launch_order_on = std::atoi(e) != 0;
    // Chains of at most 64 coordinates with a diagonal metric: several chains per wavefront (packed_core.hpp) for the families that
    // have a packed evaluator; the same bits as the wave-per-chain kernel, which DHMC_PACKED=0 brings back.
    c->packed = cfg->metric == DHMC_METRIC_DIAG && pk::family_is_packed(cfg->target) && pk::dim_is_packed(D);
    // the dense-precision normal's gradient is a D x D matvec walked through the lane group: ahead of the wave kernel to 32 coordinates
    // when the chains fill the GPU's SIMDs several times over (D = 8 … 32, 16384 chains: 3.5 … 1.7 x; D = 64: 0.75 x; 1024 chains: 0.5 x —
    // profiles/r06_packed_dense_normal.txt); the chain count is checked per call (choose_engine)
    if (cfg->target == DHMC_TARGET_DENSE_NORMAL) c->packed = c->packed && D <= 32;
    if (const char* e = std::getenv("DHMC_PACKED")) { c->packed = c->packed && std::atoi(e) != 0; c->packed_force = c->packed; }
    // … and as a PIPELINE of four wavefronts per chain (integrator ‖ turn statistics ‖ visited statistic ‖ proposals, nuts_pipeline_kernel.hpp): the lowest
    // latency per leapfrog of a short chain, for launches that wait for a few deep chains
    c->pipeline = cfg->metric == DHMC_METRIC_DIAG && D <= 256 &&          // (rows of 64, 128 or 256 doubles: one, two or four slots per lane)
              (cfg->target == DHMC_TARGET_STD_NORMAL || cfg->target == DHMC_TARGET_DIAG_NORMAL || cfg->target == DHMC_TARGET_TRIDIAG_NORMAL ||
               cfg->target == DHMC_TARGET_FUNNEL || cfg->target == DHMC_TARGET_DENSE_NORMAL || cfg->target == DHMC_TARGET_ALWAYS_DIVERGENT);
    if (const char* e = std::getenv("DHMC_PIPELINE")) { c->pipeline = c->pipeline && std::atoi(e) != 0; c->pipeline_force = c->pipeline; }
    if (env_list_value("DHMC_PK", "queue", &ev)) c->pk_queue = ev != 0;
    if (env_list_value("DHMC_PK", "max_waves", &ev)) c->pk_max_waves = std::max(0, (int)ev);
    if (env_list_value("DHMC_PK", "many_chains", &ev)) c->many_chains_min = std::max(0, (int)ev);     // (tests: the chain count from which a launch counts as throughput-bound)
    if (env_list_value("DHMC_PK", "handover", &ev)) c->pk_handover = std::max(0, (int)ev);
    if (env_list_value("DHMC_PK", "align", &ev)) { const int v = (int)ev; if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->pk_align = v; }
    if (env_list_value("DHMC_PK", "lds_levels", &ev)) c->pk_lds_levels = (int)ev;
    if (env_list_value("DHMC_PK", "cpl", &ev)) { if (ev == 2 || ev == 4) c->pk_cpl = (int)ev; }
    if (env_list_value("DHMC_DENSE", "fuse_k2", &ev)) c->fuse_k2 = ev != 0;
    if (env_list_value("DHMC_DENSE", "row_lists", &ev)) c->dense_row_lists = ev != 0;
    if (env_list_value("DHMC_DENSE", "parts", &ev)) { if (ev >= 1 && ev <= 4) c->dense_parts = (int)ev; }
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
        // … and packed (several chains per wavefront, packed_kernels.hpp PackedFunctor) where ℓ is a sum of per-coordinate terms: one
        // more module per lane-group shape (its only kernel), compiled when the first such context of the functor is created
        if (c->user && (c->user->traits & 4) && cfg->metric == DHMC_METRIC_DIAG && pk::dim_is_packed(D)) {
            const int cpl = D > 32 ? 4 : 2, Lp = pk::lanes_per_chain(D, cpl);
            const auto pkey = std::make_pair((int)cfg->device, -(100 * Lp + cpl));
            auto pit = U.built.find(pkey);
            if (pit == U.built.end()) {
                std::vector<char> code;
                std::vector<std::string> low;
                UserKernels K;
                auto load = [&]() { return rtc_load(code, low, &K.mod, {&K.packed}); };
                if ((rc = rtc_compile(U.source, U.name, pkey.second, false, &code, &low))) return fail(rc);
                if ((rc = load())) {
                    code.clear(); low.clear();
                    if ((rc = rtc_compile(U.source, U.name, pkey.second, false, &code, &low, true)) || (rc = load())) return fail(rc);
                }
                pit = U.built.emplace(pkey, K).first;
            }
            c->user_packed = pit->second.packed;
            c->user_packed_cpl = cpl;
            c->packed = 1;
            if (const char* e = std::getenv("DHMC_PACKED")) { c->packed = std::atoi(e) != 0; c->packed_force = c->packed; }
        }
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
        if (env_list_value("DHMC_DENSE", "rounds", &ev)) c->dense_rounds = ev != 0;  // 0: wave-per-chain matvec kernel
        if (c->per_chain_dense) c->dense_rounds = 0;   // the GEMM engine shares one M⁻¹ across the rows of a product
        c->dense_products = c->per_chain_dense ? 2 : 1;
        if (env_list_value("DHMC_DENSE", "products", &ev)) { if (ev == 1 || ev == 2) c->dense_products = (int)ev; }
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
            if (Dp <= 256 && (rc = dev_alloc(c, &c->lr.S1L, (size_t)c->lr.nz * C * WAVE))) return fail(rc);   // (the fused η + link kernel)
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
        launch_logistic_eta_link(P, R, L, q, C, c->stream);                                                              // η = Q·Xᵀ, r, the blocks' sums
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

uint64_t dhmc_last_run_rounds(const dhmc_ctx* c) { return c ? c->last_rounds : 0; }
uint64_t dhmc_last_run_leapfrogs(const dhmc_ctx* c) { return c ? c->last_leapfrogs : 0; }
uint64_t dhmc_workspace_bytes(const dhmc_ctx* c) { return c ? c->ws_bytes : 0; }

}  // extern "C"
