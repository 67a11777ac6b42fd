// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of src/hamiltonian.jl: Gaussian kinetic energy (diag / dense), evaluated log
// density, phase point, joint log density, leapfrog.  Elementwise arithmetic keeps the
// reference's operation order and rounding (separate multiply and add, as Julia's array
// expressions evaluate); reductions use the ABI's wave order (mathops.hpp).
#pragma once
#include <cmath>
#include <memory>
#include <vector>
#include "mathops.hpp"
#include "targets.hpp"

namespace oracle {

using Vec = std::vector<double>;
using VecP = std::shared_ptr<const Vec>;

// status bits as include/dhmc.h
constexpr uint32_t ST_NONFINITE_POSITION = 1u;
constexpr uint32_t ST_INVALID_INITIAL = 2u;
constexpr uint32_t ST_STEPSIZE_SEARCH_FAILED = 4u;
constexpr uint32_t ST_NONFINITE_START_DENSITY = 8u;

// src/hamiltonian.jl:56-87.  diag: Minv has D entries, W = sqrt.(1 ./ Minv) (:80).
// dense: Minv is D×D row-major symmetric, W = cholesky(inv(Minv)).L (:73), lower triangular.
struct GaussianKineticEnergy {
    int D = 0;
    bool dense = false;
    Vec Minv, W;
    static GaussianKineticEnergy unit(int D) {  // :87
        GaussianKineticEnergy k;
        k.D = D;
        k.Minv.assign(D, 1.0);
        k.W.assign(D, 1.0);
        return k;
    }
    static GaussianKineticEnergy diagonal(const double* minv, int D) {  // :80
        GaussianKineticEnergy k;
        k.D = D;
        k.Minv.assign(minv, minv + D);
        k.W.resize(D);
        for (int i = 0; i < D; ++i) k.W[i] = std::sqrt(1.0 / minv[i]);
        return k;
    }
    static GaussianKineticEnergy dense_from(const double* minv, int D);  // :73, metric.hpp
    // calculate_p♯ = ∇kinetic_energy = M⁻¹ p  (:110,:117)
    void p_sharp(const double* p, double* out) const {
        if (!dense) {
            for (int i = 0; i < D; ++i) out[i] = Minv[i] * p[i];
        } else {
            // k-ordered fma chain per row (what one fp64 MFMA accumulation chain computes)
            for (int i = 0; i < D; ++i) {
                double acc = 0.0;
                const double* row = &Minv[(size_t)i * D];
                if (sequential_sums()) for (int k = 0; k < D; ++k) acc = acc + row[k] * p[k];
                else for (int k = 0; k < D; ++k) acc = __builtin_fma(row[k], p[k], acc);
                out[i] = acc;
            }
        }
    }
    // kinetic_energy = dot(p, M⁻¹ p) / 2  (:103), given p♯ already
    double kinetic_energy(const double* p, const double* ps) const {
        return wave_dot(p, ps, D) / 2.0;
    }
};

// src/hamiltonian.jl:165-186
struct EvaluatedLogDensity {
    VecP q;
    double lq = 0;
    VecP g;
};

// src/hamiltonian.jl:225-234 (+ p♯ cached next to p; the reference recomputes it, :103,:121)
struct PhasePoint {
    EvaluatedLogDensity Q;
    VecP p;
    VecP ps;   // M⁻¹ p
    double K = 0;  // kinetic_energy(κ, p)
    VecP u;    // M⁻¹ ∇ℓq — carried only by the one-product recurrence of the dense metric (leapfrog below)
};
using Z = std::shared_ptr<const PhasePoint>;

struct Hamiltonian {
    const GaussianKineticEnergy* kappa;
    const Target* target;
    MathOps M;
    uint32_t* status;  // where reference `throw`s are recorded for this chain
    // Dense metric only.  false: the reference's recurrence, two products M⁻¹·v per leapfrog (M⁻¹pₘ :278, M⁻¹p′ :103/
    // NUTS.jl:121).  true: ONE product per leapfrog, u′ = M⁻¹∇ℓ(q′), with M⁻¹pₘ = p♯ + (ϵ/2)u and p♯′ = M⁻¹pₘ + (ϵ/2)u′
    // propagated by linearity from the p♯ = M⁻¹p and u = M⁻¹∇ℓ computed afresh at the start of every transition
    // (include/dhmc.h dhmc_set_dense_products; the device round engine's default).  Mathematically the same map; the
    // rounding differs (tests/test_gpu_tolerance.py bounds it).
    bool one_product = false;
};

inline bool all_finite(const double* x, int n) {
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(x[i])) return false;
    return true;
}

// src/hamiltonian.jl:202-217.  The reference THROWS on a non-finite position (:203) and, when
// strict, on invalid ℓ/∇ℓ (:212-216); here the throw is recorded in *status and the point is
// given ℓq = -Inf so the caller can unwind the chain as a divergence.
inline EvaluatedLogDensity evaluate_l(const Hamiltonian& H, VecP q, bool strict) {
    int D = H.target->D;
    auto g = std::make_shared<Vec>(D);
    EvaluatedLogDensity Q;
    Q.q = q;
    if (!all_finite(q->data(), D)) {
        *H.status |= ST_NONFINITE_POSITION;
        Q.lq = -INFINITY;
        for (auto& x : *g) x = 0.0;
        Q.g = g;
        return Q;
    }
    double lq;
    H.target->eval(H.M, q->data(), lq, g->data());
    bool ok = (std::isfinite(lq) && all_finite(g->data(), D)) || lq == -INFINITY;
    if (!ok) {
        if (strict) *H.status |= ST_INVALID_INITIAL;
        lq = -INFINITY;
    }
    Q.lq = lq;
    Q.g = g;
    return Q;
}

inline Z make_phasepoint(const Hamiltonian& H, const EvaluatedLogDensity& Q, VecP p) {
    auto z = std::make_shared<PhasePoint>();
    z->Q = Q;
    z->p = p;
    auto ps = std::make_shared<Vec>(H.kappa->D);
    H.kappa->p_sharp(p->data(), ps->data());
    z->K = H.kappa->kinetic_energy(p->data(), ps->data());
    z->ps = ps;
    if (H.one_product && H.kappa->dense) {   // the anchor of the one-product recurrence: u = M⁻¹∇ℓq, a fresh product
        auto u = std::make_shared<Vec>(H.kappa->D);
        H.kappa->p_sharp(Q.g->data(), u->data());
        z->u = u;
    }
    return z;
}

// src/hamiltonian.jl:251-256
inline double logdensity(const Hamiltonian&, const PhasePoint& z) {
    double lq = z.Q.lq;
    if (!std::isfinite(lq)) return -INFINITY;
    double K = z.K;
    return lq - (std::isfinite(K) ? K : INFINITY);
}

// src/hamiltonian.jl:273-282.  Backward motion is a negative ϵ (src/NUTS.jl:30).
inline Z leapfrog(const Hamiltonian& H, const PhasePoint& z, double eps) {
    int D = H.kappa->D;
    const Vec& p = *z.p;
    const Vec& q = *z.Q.q;
    const Vec& g = *z.Q.g;
    double h = eps / 2;                                  // ϵ/2 formed first (:277)
    Vec pm(D), t(D);
    for (int i = 0; i < D; ++i) pm[i] = p[i] + h * g[i];            // :277
    const bool one = H.one_product && H.kappa->dense && z.u;
    if (one) {
        const Vec& ps = *z.ps;
        const Vec& u = *z.u;
        for (int i = 0; i < D; ++i) t[i] = ps[i] + h * u[i];        // M⁻¹pₘ = M⁻¹p + (ϵ/2) M⁻¹∇ℓq, no product
    } else {
        H.kappa->p_sharp(pm.data(), t.data());                       // ∇kinetic_energy(κ, pₘ)
    }
    auto q1 = std::make_shared<Vec>(D);
    for (int i = 0; i < D; ++i) (*q1)[i] = q[i] + eps * t[i];       // :278
    EvaluatedLogDensity Q1 = evaluate_l(H, q1, false);               // :279
    auto p1 = std::make_shared<Vec>(D);
    const Vec& g1 = *Q1.g;
    for (int i = 0; i < D; ++i) (*p1)[i] = pm[i] + h * g1[i];       // :280
    if (!one) return make_phasepoint(H, Q1, p1);                     // :281
    // the leapfrog's one product, then p♯′ = M⁻¹pₘ + (ϵ/2) u′
    auto z1 = std::make_shared<PhasePoint>();
    auto u1 = std::make_shared<Vec>(D);
    H.kappa->p_sharp(g1.data(), u1->data());
    auto ps1 = std::make_shared<Vec>(D);
    for (int i = 0; i < D; ++i) (*ps1)[i] = t[i] + h * (*u1)[i];
    z1->Q = Q1;
    z1->p = p1;
    z1->ps = ps1;
    z1->u = u1;
    z1->K = H.kappa->kinetic_energy(p1->data(), ps1->data());
    return z1;
}

}  // namespace oracle
