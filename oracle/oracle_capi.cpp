// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// C ABI over the oracle for the ctypes harness in tests/oracle_lib.py.  The multi-chain
// `oracle_*` context mirrors include/dhmc.h entry for entry (host pointers only) so that the
// parity tests call both sides the same way; the `oracle_unit_*` hooks expose the individual
// reference functions for the known-answer tests of SURVEY.md §8(c).
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/dhmc.h"
#include "dummy.hpp"
#include "diagnostics.hpp"
#include "mcmc.hpp"
#include "metric.hpp"

using namespace oracle;

struct oracle_ctx {
    dhmc_config cfg;
    MathOps M;
    std::unique_ptr<Target> target;
    std::vector<Chain> chains;
    NUTS alg;
    int threads = 1;
};

static std::unique_ptr<Target> make_target(const dhmc_config& cfg) {
    int D = cfg.dim;
    const double* pd = (const double*)cfg.target_params;
    switch (cfg.target) {
    case DHMC_TARGET_STD_NORMAL: return std::make_unique<StdNormal>(D);
    case DHMC_TARGET_DIAG_NORMAL:
        if (cfg.target_params_bytes != sizeof(double) * 2 * (size_t)D) return nullptr;
        return std::make_unique<DiagNormal>(D, pd, pd + D);
    case DHMC_TARGET_TRIDIAG_NORMAL:
        if (cfg.target_params_bytes != sizeof(double) * 2 * (size_t)D) return nullptr;
        return std::make_unique<TridiagNormal>(D, pd, pd + D);
    case DHMC_TARGET_FUNNEL: return D >= 2 ? std::make_unique<Funnel>(D) : nullptr;
    case DHMC_TARGET_DENSE_NORMAL:
        if (cfg.target_params_bytes != sizeof(double) * ((size_t)D + (size_t)D * D)) return nullptr;
        return std::make_unique<DenseNormal>(D, pd, pd + D);
    case DHMC_TARGET_LOGISTIC: {
        if (!cfg.target_params || cfg.target_params_bytes < 8) return nullptr;
        int64_t n;
        std::memcpy(&n, cfg.target_params, 8);
        if (n <= 0 || cfg.target_params_bytes != 8 + sizeof(double) * (uint64_t)n * (D + 1)) return nullptr;
        const double* x = (const double*)((const char*)cfg.target_params + 8);
        return std::make_unique<Logistic>(D, n, x, x + (size_t)n * D);
    }
    case DHMC_TARGET_ALWAYS_DIVERGENT: return std::make_unique<AlwaysDivergent>(D);
    default: return nullptr;
    }
}

extern "C" {

int oracle_create(const dhmc_config* cfg, int det_math, oracle_ctx** out) {
    if (!cfg || !out) return DHMC_ERR_INVALID_ARGUMENT;
    if (cfg->dim <= 0 || cfg->chains <= 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (!(0 < cfg->max_depth && cfg->max_depth <= MAX_DIRECTIONS_DEPTH)) return DHMC_ERR_INVALID_ARGUMENT;  // NUTS.jl:190
    if (!(cfg->min_delta < 0)) return DHMC_ERR_INVALID_ARGUMENT;                                            // NUTS.jl:191
    if (cfg->metric != DHMC_METRIC_DIAG && cfg->metric != DHMC_METRIC_DENSE) return DHMC_ERR_UNSUPPORTED;
    auto c = std::make_unique<oracle_ctx>();
    c->cfg = *cfg;
    c->M.det = det_math != 0;
    c->target = make_target(*cfg);
    if (!c->target) return DHMC_ERR_INVALID_ARGUMENT;
    c->cfg.target_params = nullptr;
    c->alg.max_depth = cfg->max_depth;
    c->alg.min_delta = cfg->min_delta;
    c->chains.resize(cfg->chains);
    for (int i = 0; i < cfg->chains; ++i) {
        c->chains[i].stream = ChainStream{cfg->seed, (uint32_t)(cfg->chain_offset + i)};
        c->chains[i].kappa = GaussianKineticEnergy::unit(cfg->dim);
    }
    if (cfg->metric == DHMC_METRIC_DENSE) {
        std::vector<double> I((size_t)cfg->dim * cfg->dim, 0.0);
        for (int i = 0; i < cfg->dim; ++i) I[(size_t)i * cfg->dim + i] = 1.0;
        GaussianKineticEnergy k = GaussianKineticEnergy::dense_from(I.data(), cfg->dim);
        for (auto& ch : c->chains) ch.kappa = k;
    }
    *out = c.release();
    return DHMC_OK;
}
int oracle_destroy(oracle_ctx* c) { delete c; return DHMC_OK; }
int oracle_set_threads(oracle_ctx* c, int n) { c->threads = n < 1 ? 1 : n; return DHMC_OK; }
// every sum of the oracle left to right instead of in the ABI's order (mathops.hpp sequential_sums): process-wide
int oracle_set_sequential_sums(int on) { sequential_sums() = on != 0; return DHMC_OK; }
// dhmc_set_dense_products: 2 = the reference's recurrence (two M⁻¹ products per leapfrog), 1 = one product (hamiltonian.hpp)
int oracle_set_dense_products(oracle_ctx* c, int n) {
    if (!c || (n != 1 && n != 2) || c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    for (auto& ch : c->chains) ch.dense_one_product = (n == 1);
    return DHMC_OK;
}

static int any_failure(oracle_ctx* c) {
    for (auto& ch : c->chains) if (ch.status) return DHMC_ERR_CHAIN_FAILURE;
    return DHMC_OK;
}

int oracle_init(oracle_ctx* c, const double* q0) {
    int D = c->cfg.dim, C = c->cfg.chains;
#pragma omp parallel for num_threads(c->threads) schedule(dynamic)
    for (int i = 0; i < C; ++i)
        initialize_warmup_state(c->chains[i], *c->target, c->M, q0 ? q0 + (size_t)i * D : nullptr);
    return any_failure(c);
}
int oracle_get_position(oracle_ctx* c, double* q, double* lq, double* grad) {
    int D = c->cfg.dim, C = c->cfg.chains;
    for (int i = 0; i < C; ++i) {
        const auto& Q = c->chains[i].Q;
        if (q) std::memcpy(q + (size_t)i * D, Q.q->data(), sizeof(double) * D);
        if (lq) lq[i] = Q.lq;
        if (grad) std::memcpy(grad + (size_t)i * D, Q.g->data(), sizeof(double) * D);
    }
    return DHMC_OK;
}
int oracle_set_metric_diag(oracle_ctx* c, const double* minv, int per_chain) {
    int D = c->cfg.dim, C = c->cfg.chains;
    for (int i = 0; i < C * D; ++i)
        if (!(minv[per_chain ? i : i % D] > 0)) return DHMC_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < C; ++i)
        c->chains[i].kappa = GaussianKineticEnergy::diagonal(per_chain ? minv + (size_t)i * D : minv, D);
    return DHMC_OK;
}
int oracle_set_metric_dense(oracle_ctx* c, const double* minv) {
    if (c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    GaussianKineticEnergy k = GaussianKineticEnergy::dense_from(minv, c->cfg.dim);
    if (k.D != c->cfg.dim) return DHMC_ERR_INVALID_ARGUMENT;  // not positive definite
    for (auto& ch : c->chains) ch.kappa = k;
    return DHMC_OK;
}
int oracle_update_metric_dense(oracle_ctx* c, const double* draws, int64_t N, double lambda) {
    if (c->cfg.metric != DHMC_METRIC_DENSE || N < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;
    if (c->cfg.dense_per_chain) {   // the reference's own semantics: every chain from its own draws (src/mcmc.jl:281-285)
        for (int i = 0; i < c->cfg.chains; ++i) {
            std::vector<double> Si = pooled_regularized_cov(draws + (size_t)i * N * c->cfg.dim, N, c->cfg.dim, lambda);
            GaussianKineticEnergy ki = GaussianKineticEnergy::dense_from(Si.data(), c->cfg.dim);
            if (ki.D != c->cfg.dim) return DHMC_ERR_INVALID_ARGUMENT;
            c->chains[i].kappa = ki;
        }
        return DHMC_OK;
    }
    std::vector<double> S = pooled_regularized_cov(draws, (int64_t)c->cfg.chains * N, c->cfg.dim, lambda);
    GaussianKineticEnergy k = GaussianKineticEnergy::dense_from(S.data(), c->cfg.dim);
    if (k.D != c->cfg.dim) return DHMC_ERR_INVALID_ARGUMENT;
    for (auto& ch : c->chains) ch.kappa = k;
    return DHMC_OK;
}
int oracle_get_metric_dense_chain(oracle_ctx* c, int32_t chain, double* M, double* W) {
    if (chain < 0 || chain >= c->cfg.chains) return DHMC_ERR_INVALID_ARGUMENT;
    const auto& k = c->chains[chain].kappa;
    if (!k.dense) return DHMC_ERR_INVALID_ARGUMENT;
    if (M) std::memcpy(M, k.Minv.data(), sizeof(double) * k.Minv.size());
    if (W) std::memcpy(W, k.W.data(), sizeof(double) * k.W.size());
    return DHMC_OK;
}
int oracle_get_metric_dense_Minv(oracle_ctx* c, double* M) {
    const auto& k = c->chains[0].kappa;
    if (!k.dense) return DHMC_ERR_INVALID_ARGUMENT;
    std::memcpy(M, k.Minv.data(), sizeof(double) * k.Minv.size());
    return DHMC_OK;
}
// W (lower triangular, row-major D×D) of chain 0's dense metric
int oracle_get_metric_dense_W(oracle_ctx* c, double* W) {
    const auto& k = c->chains[0].kappa;
    if (!k.dense) return DHMC_ERR_INVALID_ARGUMENT;
    std::memcpy(W, k.W.data(), sizeof(double) * k.W.size());
    return DHMC_OK;
}
int oracle_get_metric_diag(oracle_ctx* c, double* minv) {
    int D = c->cfg.dim, C = c->cfg.chains;
    for (int i = 0; i < C; ++i) std::memcpy(minv + (size_t)i * D, c->chains[i].kappa.Minv.data(), sizeof(double) * D);
    return DHMC_OK;
}
int oracle_set_stepsize(oracle_ctx* c, const double* eps, int per_chain) {
    int C = c->cfg.chains;
    for (int i = 0; i < C; ++i)
        if (!(eps[per_chain ? i : 0] > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    for (int i = 0; i < C; ++i) c->chains[i].eps = eps[per_chain ? i : 0];
    return DHMC_OK;
}
int oracle_get_stepsize(oracle_ctx* c, double* eps) {
    for (int i = 0; i < c->cfg.chains; ++i) eps[i] = c->chains[i].eps;
    return DHMC_OK;
}
int oracle_get_status(oracle_ctx* c, uint32_t* st) {
    for (int i = 0; i < c->cfg.chains; ++i) st[i] = c->chains[i].status;
    return DHMC_OK;
}
int oracle_get_da_state(oracle_ctx* c, double* mu, int64_t* m, double* Hbar, double* logeps, double* logeps_bar) {
    for (int i = 0; i < c->cfg.chains; ++i) {
        const auto& d = c->chains[i].da;
        mu[i] = d.mu; m[i] = d.m; Hbar[i] = d.Hbar; logeps[i] = d.logeps; logeps_bar[i] = d.logeps_bar;
    }
    return DHMC_OK;
}

int oracle_find_initial_stepsize(oracle_ctx* c, const dhmc_stepsize_search* p) {
    InitialStepsizeSearch P;
    if (p) { P.initial_eps = p->initial_eps; P.log_threshold = p->log_threshold; P.maxiter_crossing = p->maxiter_crossing; }
    if (!P.valid()) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:31-33
    for (auto& ch : c->chains)
        if (!std::isnan(ch.eps)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:137
    int C = c->cfg.chains;
#pragma omp parallel for num_threads(c->threads) schedule(dynamic)
    for (int i = 0; i < C; ++i) warmup_stepsize_search(c->chains[i], *c->target, c->M, P);
    return any_failure(c);
}

// Diagnostics (include/dhmc.h: dhmc_leapfrog_trajectory / dhmc_explore_log_acceptance_ratios), same layouts
int oracle_leapfrog_trajectory(oracle_ctx* c, double eps, int32_t first, int32_t last, uint32_t momentum_index,
                               const double* p, double* delta, double* logdensity_out, double* q_out, double* p_out,
                               int32_t* range, uint32_t* status) {
    if (!(first <= 0 && 0 <= last)) return DHMC_ERR_INVALID_ARGUMENT;  // diagnostics.jl:218
    int D = c->cfg.dim, C = c->cfg.chains, npos = last - first + 1;
    int rc = DHMC_OK;
    for (int i = 0; i < C; ++i) {
        uint32_t st = 0;
        auto tr = leapfrog_trajectory(c->chains[i], *c->target, c->M, eps, first, last, p ? p + (size_t)i * D : nullptr,
                                      momentum_index, &st);
        for (int k = 0; k < npos; ++k) {
            delta[(size_t)i * npos + k] = NAN;
            logdensity_out[(size_t)i * npos + k] = NAN;
            for (int d = 0; d < D; ++d) {
                if (q_out) q_out[((size_t)i * npos + k) * D + d] = NAN;
                if (p_out) p_out[((size_t)i * npos + k) * D + d] = NAN;
            }
        }
        int lo = 0, hi = 0;
        for (auto& pi : tr) {
            int k = pi.position - first;
            delta[(size_t)i * npos + k] = pi.delta;
            logdensity_out[(size_t)i * npos + k] = pi.z->Q.lq;
            for (int d = 0; d < D; ++d) {
                if (q_out) q_out[((size_t)i * npos + k) * D + d] = (*pi.z->Q.q)[d];
                if (p_out) p_out[((size_t)i * npos + k) * D + d] = (*pi.z->p)[d];
            }
            lo = pi.position < lo ? pi.position : lo;
            hi = pi.position > hi ? pi.position : hi;
        }
        if (range) { range[2 * i] = lo; range[2 * i + 1] = hi; }
        if (status) status[i] = st;
        if (st) rc = DHMC_ERR_CHAIN_FAILURE;
    }
    return rc;
}
int oracle_explore_log_acceptance_ratios(oracle_ctx* c, const double* eps, int32_t n_eps, int32_t n_momenta,
                                         uint32_t momentum_index, const double* ps, double* out, uint32_t* status) {
    if (n_eps < 0 || n_momenta < 0) return DHMC_ERR_INVALID_ARGUMENT;
    int D = c->cfg.dim, C = c->cfg.chains;
    int rc = DHMC_OK;
    for (int i = 0; i < C; ++i) {
        uint32_t st = 0;
        double* o = out + (size_t)i * n_momenta * n_eps;
        for (int k = 0; k < n_momenta * n_eps; ++k) o[k] = NAN;
        explore_log_acceptance_ratios(c->chains[i], *c->target, c->M, eps, n_eps, n_momenta,
                                      ps ? ps + (size_t)i * n_momenta * D : nullptr, momentum_index, o, &st);
        if (status) status[i] = st;
        if (st) rc = DHMC_ERR_CHAIN_FAILURE;
    }
    return rc;
}

int oracle_run(oracle_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    DualAveraging P;
    if (da) {
        P = DualAveraging{da->delta, da->gamma, da->kappa, da->t0};
        if (!P.valid()) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:108-111
    }
    int D = c->cfg.dim, C = c->cfg.chains;
    for (auto& ch : c->chains) {
        bool need_eps = !da || da->init;
        if (need_eps && !(ch.eps > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    }
#pragma omp parallel for num_threads(c->threads) schedule(dynamic)
    for (int i = 0; i < C; ++i) {
        DrawSink s;
        if (out) {
            size_t o = (size_t)i * N;
            if (out->draws) s.draws = out->draws + o * D;
            if (out->logdensities) s.logdensities = out->logdensities + o;
            if (out->eps) s.eps = out->eps + o;
            if (out->pi) s.pi = out->pi + o;
            if (out->acceptance_rate) s.acceptance_rate = out->acceptance_rate + o;
            if (out->steps) s.steps = out->steps + o;
            if (out->term_left) s.term_left = out->term_left + o;
            if (out->term_right) s.term_right = out->term_right + o;
            if (out->depth) s.depth = out->depth + o;
            if (out->directions) s.directions = out->directions + o;
        }
        run_transitions(c->chains[i], *c->target, c->M, c->alg, N, da ? &P : nullptr,
                        da && da->init, da && da->finalize, s);
    }
    return any_failure(c);
}

int oracle_update_metric_diag(oracle_ctx* c, const double* draws, int64_t N, double lambda) {
    if (N < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:191-192 (N>=20 checked by the host wrapper)
    int D = c->cfg.dim, C = c->cfg.chains;
#pragma omp parallel for num_threads(c->threads) schedule(dynamic)
    for (int i = 0; i < C; ++i) update_metric_diag(c->chains[i], draws + (size_t)i * N * D, N, D);
    return DHMC_OK;
}

int oracle_metric_window_begin(oracle_ctx* c) {
    for (auto& ch : c->chains) metric_window_begin(ch, c->cfg.dim);
    return DHMC_OK;
}
int oracle_update_metric_diag_window(oracle_ctx* c, double lambda) {
    for (auto& ch : c->chains)
        if (ch.win_n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;
    for (auto& ch : c->chains) update_metric_diag_window(ch, c->cfg.dim);
    return DHMC_OK;
}

// ----------------------------------------------------------------------------------------
// unit hooks for the reference's known-answer tests
// ----------------------------------------------------------------------------------------

void oracle_unit_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    auto r = philox4x32_10({ctr[0], ctr[1], ctr[2], ctr[3]}, {key[0], key[1]});
    for (int i = 0; i < 4; ++i) out[i] = r[i];
}
// kind: 0 exp, 1 log, 2 log1p_nonneg, 3 sin2pi, 4 cos2pi, 5 randexp(bits), 6 randn z0 (x bits as r1, y bits as r2), 7 randn z1,
// 8 logaddexp, 9 pow, 10 / 11 the logistic link's σ(x) and log(1 + e^x) (targets.hpp LogisticTarget computes exactly these)
void oracle_unit_detmath(int kind, int64_t n, const double* x, const double* y, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        double s, c;
        uint64_t r1 = dhmc::dm_bits(x[i]), r2 = y ? dhmc::dm_bits(y[i]) : 0;
        switch (kind) {
        case 0: out[i] = dhmc::det_exp(x[i]); break;
        case 1: out[i] = dhmc::det_log(x[i]); break;
        case 2: out[i] = dhmc::det_log1p_nonneg(x[i]); break;
        case 3: dhmc::det_sincos2pi(x[i], &s, &c); out[i] = s; break;
        case 4: dhmc::det_sincos2pi(x[i], &s, &c); out[i] = c; break;
        case 5: out[i] = dhmc::det_randexp(r1); break;
        case 6: dhmc::det_randn2(r1, r2, &s, &c); out[i] = s; break;
        case 7: dhmc::det_randn2(r1, r2, &s, &c); out[i] = c; break;
        case 8: out[i] = dhmc::det_logaddexp(x[i], y[i]); break;
        case 9: out[i] = dhmc::det_pow_pos(x[i], y[i]); break;
        case 10: out[i] = dhmc::det_logistic_sigma(x[i]); break;
        case 11: out[i] = dhmc::det_log1pexp(x[i]); break;
        default: out[i] = NAN;
        }
    }
}
double oracle_unit_wave_dot(const double* a, const double* b, int n) { return wave_dot(a, b, n); }

void oracle_unit_directions(uint32_t flags, int n, int* out) {  // test_trees.jl:8-15
    Directions d{flags};
    for (int i = 0; i < n; ++i) out[i] = next_direction(d) ? 1 : 0;
}

struct DummyOut {
    int32_t valid, depth, tau_flag, assertion_failures;
    int64_t inv_left, inv_right, zeta_first, zeta_last, tau_first, tau_last, zlast, ilast, v_s;
    double omega, v_a;
    int64_t n_logp, n_visited;
};
static DummyTrajectory make_dummy(double ell_c, double ell_a, const int64_t* turning, int nt,
                                  const int64_t* divergent, int nd) {
    DummyTrajectory t;
    t.ell_c = ell_c; t.ell_a = ell_a;
    for (int i = 0; i < nt; ++i) t.turning.insert(turning[i]);
    for (int i = 0; i < nd; ++i) t.divergent.insert(divergent[i]);
    return t;
}
// adjacent_tree(nothing, trajectory, z, i, depth, is_forward) on a DummyTrajectory
void oracle_unit_dummy_adjacent_tree(double ell_c, double ell_a, const int64_t* turning, int nt,
                                     const int64_t* divergent, int nd, int64_t z, int64_t i,
                                     int depth, int fwd, DummyOut* o, double* logp, int64_t cap_logp,
                                     int64_t* visited, int64_t cap_visited) {
    DummyTrajectory t = make_dummy(ell_c, ell_a, turning, nt, divergent, nd);
    int rng = 0;
    DummyTrajectory::Visited v;
    auto r = adjacent_tree(rng, t, z, i, depth, fwd != 0, v);
    std::memset(o, 0, sizeof(*o));
    o->valid = r.valid; o->inv_left = r.invalid.left; o->inv_right = r.invalid.right;
    o->v_a = v.a; o->v_s = v.s; o->assertion_failures = t.assertion_failures;
    o->n_visited = (int64_t)t.visited.size();
    for (int64_t k = 0; k < o->n_visited && k < cap_visited; ++k) visited[k] = t.visited[k];
    if (r.valid) {
        o->zeta_first = r.zeta.first; o->zeta_last = r.zeta.last; o->omega = r.omega;
        o->tau_flag = r.tau.flag; o->tau_first = r.tau.first; o->tau_last = r.tau.last;
        o->zlast = r.zlast; o->ilast = r.ilast;
        o->n_logp = (int64_t)r.zeta.logp.size();
        for (int64_t k = 0; k < o->n_logp && k < cap_logp; ++k) logp[k] = r.zeta.logp[k];
    }
}
// sample_trajectory(nothing, trajectory, z, max_depth, Directions(flags))
void oracle_unit_dummy_sample_trajectory(double ell_c, double ell_a, const int64_t* turning, int nt,
                                         const int64_t* divergent, int nd, int64_t z,
                                         int max_depth, uint32_t flags, DummyOut* o, double* logp,
                                         int64_t cap_logp, int64_t* visited, int64_t cap_visited) {
    DummyTrajectory t = make_dummy(ell_c, ell_a, turning, nt, divergent, nd);
    int rng = 0;
    auto r = sample_trajectory(rng, t, z, max_depth, Directions{flags});
    std::memset(o, 0, sizeof(*o));
    o->valid = 1; o->depth = r.depth;
    o->inv_left = r.termination.left; o->inv_right = r.termination.right;
    o->v_a = r.v.a; o->v_s = r.v.s; o->assertion_failures = t.assertion_failures;
    o->zeta_first = r.zeta.first; o->zeta_last = r.zeta.last;
    o->n_logp = (int64_t)r.zeta.logp.size();
    for (int64_t k = 0; k < o->n_logp && k < cap_logp; ++k) logp[k] = r.zeta.logp[k];
    o->n_visited = (int64_t)t.visited.size();
    for (int64_t k = 0; k < o->n_visited && k < cap_visited; ++k) visited[k] = t.visited[k];
}

// combine_turn_statistics on explicit vectors (test_NUTS.jl:27-42); each τ is 5 vectors
// p₋, p♯₋, p₊, p♯₊, ρ of length D.  Returns 1 if turning; rho_out = ρ of the merge otherwise.
int oracle_unit_combine_turn(int D, const double* x, const double* y, double* rho_out) {
    auto mk = [&](const double* base, int k) { return std::make_shared<const Vec>(base + (size_t)k * D, base + (size_t)(k + 1) * D); };
    GeneralizedTurnStatistic X{mk(x, 0), mk(x, 1), mk(x, 2), mk(x, 3), mk(x, 4), false};
    GeneralizedTurnStatistic Y{mk(y, 0), mk(y, 1), mk(y, 2), mk(y, 3), mk(y, 4), false};
    TrajectoryNUTS traj{Hamiltonian{nullptr, nullptr, MathOps{}, nullptr}, 0.0, 1.0, -1000.0};
    auto t = traj.combine_turn_statistics(X, Y);
    if (t.turning) return 1;
    for (int i = 0; i < D; ++i) rho_out[i] = (*t.rho)[i];
    return 0;
}
// reduce leaf_acceptance_statistic(Δ_i, is_initial_i) with combine_visited_statistics, then
// acceptance_rate (test_NUTS.jl:44-55)
double oracle_unit_acceptance(int det, int n, const double* delta, const int* is_initial) {
    MathOps M; M.det = det != 0;
    TrajectoryNUTS traj{Hamiltonian{nullptr, nullptr, M, nullptr}, 0.0, 1.0, -1000.0};
    AcceptanceStatistic acc{};
    for (int i = 0; i < n; ++i) {
        AcceptanceStatistic leafv = is_initial[i] ? AcceptanceStatistic{-INFINITY, 0}
                                                  : AcceptanceStatistic{std::fmin(delta[i], 0.0), 1};
        acc = i == 0 ? leafv : traj.combine_visited_statistics(acc, leafv);
    }
    return acceptance_rate(M, acc);
}
// rand_bool_logprob n times on one stream; returns #true, *consumed = #Exp(1) draws used
// (test_NUTS.jl:10-21)
int64_t oracle_unit_rand_bool_logprob(int det, double logprob, uint64_t seed, int64_t n, int64_t* consumed) {
    MathOps M; M.det = det != 0;
    int64_t cnt = 0, used = 0;
    for (int64_t t = 0; t < n; ++t) {
        TransitionRng rng{ChainStream{seed, 0}, (uint32_t)t, 0, M};
        cnt += rand_bool_logprob(rng, logprob) ? 1 : 0;
        used += rng.ndraws;
    }
    *consumed = used;
    return cnt;
}
// logdensity(H, PhasePoint(EvaluatedLogDensity(q, lq, g), p)) with a diagonal metric
// (test_hamiltonian.jl:197-200)
double oracle_unit_logdensity(int D, double lq, const double* p, const double* minv) {
    GaussianKineticEnergy k = GaussianKineticEnergy::diagonal(minv, D);
    StdNormal t(D);
    uint32_t st = 0;
    Hamiltonian H{&k, &t, MathOps{}, &st};
    EvaluatedLogDensity Q;
    Q.q = std::make_shared<const Vec>(D, 0.0);
    Q.g = Q.q;
    Q.lq = lq;
    Z z = make_phasepoint(H, Q, std::make_shared<const Vec>(p, p + D));
    return logdensity(H, *z);
}
// n leapfrog steps of size eps from (q, p) under cfg's target with diagonal metric minv;
// writes the trajectory q,p [n][D] and joint log densities [n]; returns status bits.
uint32_t oracle_unit_leapfrog(const dhmc_config* cfg, int det, const double* minv, const double* q0,
                              const double* p0, double eps, int n, double* qs, double* ps,
                              double* pis, double* lqs) {
    auto target = make_target(*cfg);
    int D = cfg->dim;
    GaussianKineticEnergy k = GaussianKineticEnergy::diagonal(minv, D);
    uint32_t st = 0;
    MathOps M; M.det = det != 0;
    Hamiltonian H{&k, target.get(), M, &st};
    EvaluatedLogDensity Q = evaluate_l(H, std::make_shared<const Vec>(q0, q0 + D), true);
    Z z = make_phasepoint(H, Q, std::make_shared<const Vec>(p0, p0 + D));
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(z->Q.lq)) break;  // hamiltonian.jl:276
        z = leapfrog(H, *z, eps);
        std::memcpy(qs + (size_t)i * D, z->Q.q->data(), sizeof(double) * D);
        std::memcpy(ps + (size_t)i * D, z->p->data(), sizeof(double) * D);
        pis[i] = logdensity(H, *z);
        lqs[i] = z->Q.lq;
    }
    return st;
}
// n momentum draws p = W z from a dense kinetic energy built from minv [D][D] (rand_p,
// hamiltonian.jl:124; test_hamiltonian.jl:29-30): transitions 0..n-1 of chain 0; also returns W.
int oracle_unit_rand_p_dense(int D, const double* minv, uint64_t seed, int n, double* out, double* W) {
    GaussianKineticEnergy k = GaussianKineticEnergy::dense_from(minv, D);
    if (k.D != D) return 1;
    ChainStream s{seed, 0};
    for (int t = 0; t < n; ++t) {
        VecP p = rand_p(MathOps{}, k, s, PURPOSE_MOMENTUM, (uint32_t)t);
        std::memcpy(out + (size_t)t * D, p->data(), sizeof(double) * D);
    }
    std::memcpy(W, k.W.data(), sizeof(double) * (size_t)D * D);
    return 0;
}
// find_initial_stepsize with A(ϵ) = slope*ϵ + intercept (test_stepsize.jl:9-25); returns 0 on
// success, 1 where the reference throws
int oracle_unit_find_initial_stepsize_linear(double slope, double intercept, double initial_eps,
                                             double log_threshold, int maxiter, double* eps) {
    InitialStepsizeSearch P{initial_eps, log_threshold, maxiter};
    if (!P.valid()) return 2;
    bool ok = find_initial_stepsize(P, [&](double e) { return slope * e + intercept; }, *eps);
    return ok ? 0 : 1;
}
// dual averaging: state arrays of 5 doubles (mu, m, Hbar, logeps, logeps_bar)
void oracle_unit_da_init(int det, double eps, double* st) {
    MathOps M; M.det = det != 0;
    auto A = initial_adaptation_state(M, eps);
    st[0] = A.mu; st[1] = (double)A.m; st[2] = A.Hbar; st[3] = A.logeps; st[4] = A.logeps_bar;
}
void oracle_unit_da_adapt(int det, double delta, double gamma, double kappa, int t0, double* st, double a) {
    MathOps M; M.det = det != 0;
    DualAveragingState A{st[0], (int64_t)st[1], st[2], st[3], st[4]};
    A = adapt_stepsize(M, DualAveraging{delta, gamma, kappa, t0}, A, a);
    st[0] = A.mu; st[1] = (double)A.m; st[2] = A.Hbar; st[3] = A.logeps; st[4] = A.logeps_bar;
}

// Post-hoc tree-statistics diagnostics (src/diagnostics.jl:29-106) in the ABI's summation orders: out = {N, a_mean,
// a_quantiles[5], max_depth, divergence, turning, depth_counts[33]} as doubles / int64 in the dhmc_tree_statistics_summary layout
int oracle_summarize_tree_statistics(const double* pi, const double* acc, const int64_t* tl, const int64_t* tr, const int32_t* depth,
                                     int64_t chains, int64_t n, void* summary, double* ebfmi_out) {
    if (!pi || !acc || !tl || !tr || !depth || !summary || chains < 1 || n < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const TreeStatisticsSummary S = summarize_tree_statistics(acc, tl, tr, depth, chains, n);
    std::memcpy(summary, &S, sizeof(S));
    if (ebfmi_out)
        for (int64_t c = 0; c < chains; ++c) ebfmi_out[c] = ebfmi(pi + c * n, n);
    return 0;
}

}  // extern "C"
