// dhmc_run (include/dhmc.h): one call = N transitions of every chain of the context = what one of the reference's per-draw loops
// does (src/mcmc.jl:271-280 warmup, :374-379 inference).  The call is put together here from four parts:
//
//   choose_engine        which of the context's engines runs the call (a small chooser over a few numbers of the PREVIOUS call:
//                        mean tree size, whether its slowest chain held it open, the chain count — the thresholds are one struct,
//                        EnginePolicy, below; the same bits whichever engine runs)
//   bind_outputs         the caller's output arrays: device pointers pass through, host arrays get staging buffers and a copy stream
//   run_<engine>         one launcher per engine: the round engines (external model with a dense / diagonal metric, logistic
//                        regression, dense metric), the per-draw kernels in one launch, in chunks (host outputs), with the end game
//   finish_call          the call's leapfrog count and kernel time, the next call's launch order and engine inputs
#include "capi_internal.hpp"

using namespace capi;

namespace {

// Everything the parts of one dhmc_run call share.
struct RunCall {
    dhmc_ctx* c;
    int64_t N;
    const dhmc_dual_averaging* da;
    const dhmc_outputs* out;
    int C, D;
    RunParams P{};
    // engine (choose_engine)
    bool per_draw_kernel = false, packed = false, pipeline = false, endgame = false;
    Op run_op = Op::Run;
    // outputs (bind_outputs)
    struct Field { void** dev; void* host; size_t elem; int idx; };   // elem: bytes of one (chain, transition) record
    std::vector<Field> staged;
    int64_t L = 0;                 // transitions per chunk of a call with host outputs (N: one chunk)
    int nbuf = 1;
    hipError_t e = hipSuccess;
    double chunk_ms = 0.0;
};

// The thresholds of the engine choice, in one place (what DESIGN.md §5 calls "engine choice per launch").  Inputs: the chain count C, the
// GPU's CU count, and three numbers the PREVIOUS call left in the context — mean_leapfrogs_per_transition, tail_bound (its slowest
// chain did more than tail_ratio × the mean's work) and the launch order.
struct EnginePolicy {
    double tail_ratio = 3.0;                 // a launch was "tail-bound": max chain work > tail_ratio × mean chain work
    double reorder_ratio = 1.03;             // the next launch starts its chains longest-first from max > reorder_ratio × mean
    double few_chains_min_tree = 24.0;       // C <= CUs and trees of at least this many leapfrogs: the pipeline kernel
    int dense_normal_chains_per_cu = 16;     // the dense-precision normal runs packed from this many chains per CU on (below: the wave kernel)
    int many_chains_per_cu = 24;             // from this many chains per CU on a tail-bound launch stays packed and gets an end game
    int many_chains_per_pipeline_slot = 13;  // … a family without a packed evaluator: the wave kernel from this many chains per resident pipeline block
    int endgame_min_transitions = 32;        // a call shorter than this is one packed launch
    int handover_groups_per_cu = 5;          // the end game starts when this many lane groups per CU still have a chain …
    int handover_groups_per_cu_small = 10;   // … twice that below 64 chains per CU
    double gate16_min_tree = 48.0, gate8_min_tree = 24.0;   // the packed kernel's gate width from the previous call's mean tree size (else 4)
};
constexpr EnginePolicy kPolicy{};

int validate_call(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (da) {
        if (!(0 < da->delta && da->delta < 1)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:108
        if (!(da->gamma > 0)) return DHMC_ERR_INVALID_ARGUMENT;                   // :109
        if (!(0.5 < da->kappa && da->kappa <= 1)) return DHMC_ERR_INVALID_ARGUMENT;  // :110
        if (!(da->t0 >= 0)) return DHMC_ERR_INVALID_ARGUMENT;                     // :111
    }
    const int C = c->cfg.chains;
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
    return -1;   // go on
}

// Which per-draw kernel.
void choose_engine(RunCall& r) {
    const dhmc_ctx* c = r.c;
    const int C = r.C;
    const int64_t N = r.N;
    RunParams& P = r.P;
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
    const bool few_chains = C <= c->num_cus && c->mean_leapfrogs_per_transition >= kPolicy.few_chains_min_tree;
    const int many_min = c->many_chains_min > 0 ? c->many_chains_min
                         : c->packed && c->pk_handover != 0 ? kPolicy.many_chains_per_cu * c->num_cus
                         : kPolicy.many_chains_per_pipeline_slot * (int)((size_t)160 * 1024 / pipeline_lds_bytes(c->NPL <= 4 ? c->NPL : 4)) * c->num_cus;   // (5, 2 or 1 blocks per CU)
    const bool many_chains = C > many_min;
    const bool pipeline = per_draw_kernel && c->pipeline && !c->packed_force && (c->pipeline_force || (c->tail_bound && !many_chains) || few_chains);
    const bool enough_chains = c->cfg.target != DHMC_TARGET_DENSE_NORMAL || C >= kPolicy.dense_normal_chains_per_cu * c->num_cus;
    const bool packed = !pipeline && per_draw_kernel && c->packed && (c->packed_force || ((!c->tail_bound || many_chains) && enough_chains));
    const Op run_op = pipeline ? Op::RunPipeline : packed ? Op::RunPacked : Op::Run;
    if (per_draw_kernel && std::getenv("DHMC_DEBUG_ORDER"))
        std::fprintf(stderr, "[dhmc] engine: %s (N=%lld, chains %d)\n", pipeline ? "pipeline" : packed ? "packed" : "wave", (long long)N, C);
    // END GAME of a tail-bound packed launch (many chains: the rule above): once few lane groups still have a chain — no more than the
    // pipeline kernel keeps resident — the packed kernel gives those chains up at their next transition boundary and the pipeline
    // kernel finishes them at a third of the latency per leapfrog: the launch's deepest chains, which would otherwise run on alone
    // at 2.5 µs per trip (RunParams::pk_live, pk_handover_below; DHMC_PK=handover= the threshold, 0: off).
    const bool endgame = packed && !c->packed_force && c->pipeline && c->tail_bound && many_chains && c->pk_handover != 0 &&
                         c->d_chain_work && c->launch_order_on && N >= kPolicy.endgame_min_transitions;
    r.per_draw_kernel = per_draw_kernel; r.packed = packed; r.pipeline = pipeline; r.endgame = endgame; r.run_op = run_op;
}

// The packed launch's layout: coordinates per lane, LDS levels, the gate, the queue of places.
void plan_packed_launch(RunCall& r) {
    const dhmc_ctx* c = r.c;
    const int C = r.C, D = r.D;
    RunParams& P = r.P;
    // LDS: as many suspended levels as the launch's occupancy leaves room for (the kernel runs one wave per SIMD, four per CU;
    // a launch of few waves — one GPU's share of 4096 30-dim chains is 512 — has half of the CU's 160 KB to itself)
    // coordinates per lane: two while that still leaves every wave a SIMD of its own (or when the row needs no more: D <= 32 is
    // 16 lanes × 2), four beyond (D > 32 always: 16 lanes × 4)
    // (round 6: two coordinates per lane also when that is more waves than SIMDs — the 2-per-lane kernels need 256 registers, so
    // TWO of their waves share a SIMD and each runs under the other's latencies: 32768 funnel chains 1.51e9 leapfrog steps/s
    // against 1.44e9 with four per lane at one wave per SIMD, 6.8e8 against 6.0e8 in calls of 20; profiles/r06_packed_occupancy2.txt)
    int cpl = D > 32 ? 4 : 2;
    if (c->pk_cpl && D <= 32) cpl = c->pk_cpl;
    if (c->user_packed) cpl = c->user_packed_cpl;                  // (a caller's functor: the one lane-group shape its module was compiled for)
    const int L = pk::lanes_per_chain(D, cpl), gpw = 64 / L;
    const long long waves = ((long long)C + gpw - 1) / gpw;
    const int simd_waves = cpl == 2 ? 2 : 1;                       // resident waves per SIMD
    const long long wpc = std::min<long long>(4 * simd_waves, std::max<long long>(1, (waves + c->num_cus - 1) / c->num_cus));
    // (a run-time compiled kernel is launched through the module API, which has no opt-in beyond 64 KB of dynamic LDS: 48 KB there)
    const size_t budget = std::min<size_t>(c->user_packed ? (size_t)48 * 1024 : pk::kMaxLdsPerWave, (size_t)160 * 1024 / (size_t)wpc);
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
    P.pk_align = c->pk_align > 0 ? c->pk_align : c->mean_leapfrogs_per_transition >= kPolicy.gate16_min_tree ? 16 : c->mean_leapfrogs_per_transition >= kPolicy.gate8_min_tree ? 8 : 4;
    // the queue of places (packed_kernels.hpp launch_run_packed): as many waves as the GPU holds at once — one per SIMD
    P.pk_queue = c->pk_queue ? reinterpret_cast<unsigned*>(c->d_counter + 1) : nullptr;
    P.pk_max_waves = c->pk_max_waves > 0 ? c->pk_max_waves : 4 * simd_waves * c->num_cus;
}

// Outputs: device pointers pass through.  Host pointers are served from the context's persistent staging buffers (grown on demand: no
// hipMalloc / hipFree per call).  The one-kernel engine (diagonal metric) runs a call with host outputs in CHUNKS of L transitions,
// two staging buffers deep: chunk k leaves over the copy stream (strided 2-D copies into the caller's [C][N][…] arrays; truly
// asynchronous when those are page-locked — dhmc_host_alloc) while chunk k+1 computes.  The chunks are the same transitions of the
// same kernel as one launch would run: same bits.
int bind_outputs(RunCall& r) {
    dhmc_ctx* c = r.c;
    const dhmc_outputs* out = r.out;
    const int C = r.C, D = r.D;
    const int64_t N = r.N;
    RunParams& P = r.P;
    auto& staged = r.staged;
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
        staged.push_back(RunCall::Field{slot, user, elem, idx});
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
    if (rc) return rc;
    if (!staged.empty() && !c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(c, hipEventCreate(&c->ev_k0[b]));
            HIP_TRY(c, hipEventCreate(&c->ev_k1[b]));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copy[b], hipEventDisableTiming));
        }
    }
    r.L = L; r.nbuf = nbuf;
    return DHMC_OK;
}

// transitions [n0, n0 + len) of every staged field: staging buffer b (record stride L) -> the caller's arrays (stride N)
hipError_t copy_out_chunk(const RunCall& r, int b, int64_t n0, int64_t len, hipStream_t s) {
    const dhmc_ctx* c = r.c;
    const int64_t N = r.N, L = r.L;
    const int C = r.C;
    for (const auto& f : r.staged) {
        char* dst = (char*)f.host + (size_t)n0 * f.elem;
        hipError_t ce = hipMemcpy2DAsync(dst, (size_t)N * f.elem, c->stage[b][f.idx].p, (size_t)L * f.elem, (size_t)len * f.elem,
                                         (size_t)C, hipMemcpyDeviceToHost, s);
        if (ce != hipSuccess) return ce;
    }
    return hipSuccess;
}

// The per-draw kernels walk all transitions of a chain in one wave (group, pipeline), so a launch ends with its slowest chain: a
// chain whose trees are persistently deeper (a smaller adapted ϵ) and which starts in the last wave of workgroups holds the whole
// launch open — measured on BASELINE configs[1]: one chain of 4096 at 1.48 × the mean work, 189 ms instead of 171 ms per 1000
// transitions.  The next launch therefore starts its chains in the order of this one's work, longest first (results do not
// depend on the order); and the shape of the work decides the next launch's engine (tail_bound, tail_count: above).
hipError_t refresh_order(dhmc_ctx* c, int C, int64_t n_launch) {
    c->h_chain_work.resize(C);
    hipError_t he = hipMemcpyAsync(c->h_chain_work.data(), c->d_chain_work, sizeof(unsigned) * C, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) return he;
    unsigned long long sum = 0;
    unsigned mx = 0;
    for (unsigned w : c->h_chain_work) { sum += w; mx = std::max(mx, w); }
    c->launch_order_valid = false;
    c->tail_count = 0;
    if ((double)mx * C > kPolicy.reorder_ratio * (double)sum) {
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
    c->tail_bound = (double)mx * C > kPolicy.tail_ratio * (double)sum;
    if (std::getenv("DHMC_DEBUG_ORDER"))
        std::fprintf(stderr, "[dhmc] launch order: N=%lld max=%u mean=%.1f valid=%d first=%d tail_bound=%d tail_count=%d\n", (long long)n_launch, mx,
                     (double)sum / C, (int)c->launch_order_valid, c->launch_order_valid ? c->h_launch_order[0] : -1, (int)c->tail_bound, c->tail_count);
    return he;
}

// dense round engine (dense_rounds.hpp) with the host's callback as the density, one batch on one stream
int run_external_dense_rounds(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
    P.one_product = c->dense_products == 1;
    RoundArgs ra{P, c->rb};
    const int ld = c->Dpad;
    const RoundBuffers& R = c->rb;
    e = hipMemsetAsync(R.list_count, 0, 2 * sizeof(int), c->stream);
    if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) return rc; }
    unsigned long long rounds = 0;
    int done = 0;
    while (e == hipSuccess && done < C) {
        for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
            launch_gemm_rows(R.cp, c->d_WT, R.tbuf, ld, C, R.list, R.list_count, c->stream);       // p₀ = z·Wᵀ
            launch_gemm_rows(R.tbuf, c->d_Minv, R.cps, ld, C, R.list, R.list_count, c->stream);    // p♯₀
            if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, R.list, R.list_count, c->stream);   // u₀ = ∇ℓq₀·M⁻¹
            if ((rc = dispatch(c, Op::RoundK0, &ra))) return rc;
            e = hipMemsetAsync(R.list_count, 0, sizeof(int), c->stream);
            if (!P.one_product) launch_gemm_rows(R.cp, c->d_Minv, R.tbuf, ld, C, nullptr, nullptr, c->stream);   // M⁻¹pₘ
            DHMC_EXT_NPL(rounds_k2a_dense_external_kernel, dim3(C), ra.P, ra.R)                    // q′ (one product: M⁻¹pₘ = p♯ + (ϵ/2)u)
            if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
            rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
            if (rc) { c->poisoned = true; return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
            DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′
            if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, nullptr, nullptr, c->stream);   // u′ = ∇ℓq′·M⁻¹
            else launch_gemm_rows(R.cp, c->d_Minv, R.cps, ld, C, nullptr, nullptr, c->stream);     // p♯
            if ((rc = dispatch(c, Op::RoundK3, &ra))) return rc;
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&done, R.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    c->last_rounds = rounds;
    (void)C; (void)rc;
    return DHMC_OK;
}

// round engine with the host's callback as the gradient (external_rounds.hpp)
int run_external_rounds(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
    RoundArgs ra{P, c->rb};
    e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
    if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) return rc; }
    unsigned long long rounds = 0;
    int done = 0;
    while (e == hipSuccess && done < C) {
        for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
            launch_logistic_op(0, c->NPL, ra, c->lr, c->stream);                                   // p = W∘z, p♯
            if ((rc = dispatch(c, Op::RoundK0, &ra))) return rc;
            e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
            launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
            if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
            rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
            if (rc) { c->poisoned = true; return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
            DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′, p♯
            if ((rc = dispatch(c, Op::RoundK3, &ra))) return rc;
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&done, c->rb.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    c->last_rounds = rounds;
    (void)C; (void)rc;
    return DHMC_OK;
}

// GEMM-gradient round engine (logistic_rounds.hpp)
int run_logistic_rounds(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
    RoundArgs ra{P, c->rb};
    const int ld = c->Dpad;
    const int npad = (int)c->tp.npad;
    e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
    if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) return rc; }
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
            if ((rc = dispatch(c, Op::RoundK0, &ra))) return rc;
            e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
            launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
            launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);                                   // the rows of this round
            launch_logistic_eta_link(ra.P, ra.R, c->lr, c->st.q, C, c->stream, /*sums_kernel=*/false);                    // η = Q′·Xᵀ, r, S₁ (one kernel)
            launch_gemm_splitk(c->lr.H, npad, c->tp.a, ld, c->lr.P, ld, (size_t)C * ld, C, npad, ld, DHMC_LOGISTIC_BLOCK,
                               c->lr.act, c->lr.act_count, c->stream);                             // Xᵀr = R·X, block by block
            launch_logistic_op(3, c->NPL, ra, c->lr, c->stream);                                   // ∇ℓ, ℓ, p′, p♯
            if ((rc = dispatch(c, Op::RoundK3, &ra))) return rc;
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
    (void)C; (void)rc;
    return DHMC_OK;
}

// round-based dense engine (dense_rounds.hpp), two half-batches on two streams
int run_dense_rounds(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
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
        if ((rc = dispatch(c, Op::RoundStart, &H[h].ra, H[h].s, true))) return rc;
    unsigned long long rounds = 0;
    int done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Four rounds of both half-batches = 64 launches on two streams per host pass.  (A hipGraph of those four rounds was measured in
    // round 3: 8.05e6 leapfrog-steps/s against 8.84e6 with plain launches — the host was never the limit — and removed in round 6.)
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
        if ((rc = enqueue_reps())) return rc;
        if (e == hipSuccess) e = hipGetLastError();
        for (int h = 1; h < nh && e == hipSuccess; ++h) {
            e = hipEventRecord(c->ev_joins[h], c->streams[h]);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_joins[h], 0);
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
        row_lists = c->dense_row_lists && done[1] + done[3] + done[5] + done[7] > 0;
    }
    c->last_rounds = rounds;
    (void)C; (void)rc;
    return DHMC_OK;
}

// the one-kernel engine, host outputs, in chunks: kernel of chunk k ‖ copy of chunk k-1
int run_chunked(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
    const int64_t N = r.N, L = r.L;
    const dhmc_dual_averaging* da = r.da;
    auto& staged = r.staged;
    double& chunk_ms = r.chunk_ms;
    const Op run_op = r.run_op;
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
        if (e == hipSuccess && (rc = dispatch(c, run_op, &Q))) { (void)hipDeviceSynchronize(); return rc; }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(c->ev_k1[b], c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_k1[b], 0);
        if (e == hipSuccess) e = copy_out_chunk(r, b, n0, len, c->copy_stream);
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
    (void)C; (void)rc;
    return DHMC_OK;
}

// a tail-bound packed launch of many chains with its END GAME in the pipeline kernel (choose_engine)
int run_endgame(RunCall& r) {
    dhmc_ctx* c = r.c;
    RunParams& P = r.P;
    hipError_t& e = r.e;
    const int C = r.C;
    int rc = DHMC_OK;
    if (!c->d_prog) {
        if ((rc = dev_alloc(c, &c->d_prog, (size_t)C)) || (rc = dev_alloc(c, &c->d_evicted, (size_t)C))) return rc;
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
    B.pk_handover_below = c->pk_handover > 0 ? c->pk_handover : (C >= 64 * c->num_cus ? kPolicy.handover_groups_per_cu : kPolicy.handover_groups_per_cu_small) * c->num_cus;
    if (e == hipSuccess && (rc = dispatch(c, Op::RunPacked, &B))) { (void)hipDeviceSynchronize(); return rc; }
    unsigned n_given_up = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_given_up, d_evict_count, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && n_given_up > 0) {          // the chains the packed launch gave up, in the order it gave them up
        RunParams T = B;
        T.C = (int)n_given_up;
        T.launch_order = c->d_evicted;
        T.pk_live = nullptr; T.pk_handover_below = 0;
        if ((rc = dispatch(c, Op::RunPipeline, &T))) { (void)hipDeviceSynchronize(); return rc; }
    }
    if (std::getenv("DHMC_DEBUG_ORDER")) std::fprintf(stderr, "[dhmc] end game: %u chains handed to the pipeline kernel\n", n_given_up);
    if (e == hipSuccess) e = hipGetLastError();
    (void)C; (void)rc;
    return DHMC_OK;
}

// The call's counts and times, the next call's launch order and engine inputs, the last copies.
int finish_call(RunCall& r) {
    dhmc_ctx* c = r.c;
    const RunParams& P = r.P;
    const int C = r.C, nbuf = r.nbuf;
    const int64_t N = r.N;
    const auto& staged = r.staged;
    hipError_t e = r.e;
    if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess && nbuf == 1 && !staged.empty()) e = copy_out_chunk(r, 0, 0, N, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&c->last_leapfrogs, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    // chain_work holds the counts of the call's LAST launch: the whole call, or a chunked call's last chunk (run_chunked) — that many
    // transitions decide (a short launch's counts say little about the chains, and sorting is not free)
    const int64_t counted = nbuf == 2 && r.L > 0 ? N - ((N + r.L - 1) / r.L - 1) * r.L : N;
    const bool reorder = P.chain_work && counted >= kPolicy.endgame_min_transitions;
    if (e == hipSuccess && reorder) e = refresh_order(c, C, counted);
    else if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && N > 0) c->mean_leapfrogs_per_transition = (double)c->last_leapfrogs / ((double)C * (double)N);
    if (e == hipSuccess) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->last_ms = nbuf == 2 ? r.chunk_ms : ms;     // kernel time only: not the waits for copies between the chunks
    }
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

int run_call(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (const int v = validate_call(c, N, da); v >= 0) return v;
    RunCall r{c, N, da, out, c->cfg.chains, c->cfg.dim};
    RunParams& P = r.P;
    const int C = r.C, D = r.D;
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset;
    P.max_depth = c->cfg.max_depth; P.nvec = c->nvec; P.min_delta = c->cfg.min_delta; P.seed = c->cfg.seed;
    P.N = N; P.st = c->st; P.tp = c->tp; P.leapfrog_counter = c->d_counter;
    P.l1_in_lds = c->l1_in_lds;
    P.k3_block = c->k3_block;
    P.one_product = c->cfg.metric == DHMC_METRIC_DENSE && c->dense_products == 1;
    choose_engine(r);
    if (r.packed) plan_packed_launch(r);
    if (c->win_n >= 0) {       // an open metric window: every transition's draw joins the running moments (capi_metric.hip)
        P.win_mean = c->d_win; P.win_m2 = c->d_win + (size_t)C * c->Dpad; P.win_n0 = c->win_n;
    }
    if (da) {
        P.adapt = 1; P.da_init = da->init; P.da_finalize = da->finalize; P.t0 = da->t0;
        P.delta = da->delta; P.gamma = da->gamma; P.kappa = da->kappa;
    }
    if (int rc = bind_outputs(r)) return rc;

    hipError_t& e = r.e;
    e = hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
    int rc = DHMC_OK;
    if (e != hipSuccess) {}
    else if (c->external && c->cfg.metric == DHMC_METRIC_DENSE) rc = run_external_dense_rounds(r);
    else if (c->external) rc = run_external_rounds(r);
    else if (c->logistic_rounds) rc = run_logistic_rounds(r);
    else if (c->cfg.metric == DHMC_METRIC_DENSE && c->dense_rounds) rc = run_dense_rounds(r);
    else if (r.nbuf == 2) rc = run_chunked(r);
    else if (r.endgame) rc = run_endgame(r);           // (needs the whole call in one launch: device outputs, or host outputs in one chunk)
    else { rc = dispatch(c, r.run_op, &P); if (!rc) e = hipGetLastError(); }
    if (rc) return rc;
    return finish_call(r);
}
}  // namespace

extern "C" {
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

}  // extern "C"
