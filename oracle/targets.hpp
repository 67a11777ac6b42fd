// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU definitions of the shipped log-density family, i.e. what the user would hand the
// reference through LogDensityProblems.logdensity_and_gradient (src/hamiltonian.jl:204).
// The reference's own test targets come from LogDensityTestSuite 0.7, which is not vendored
// under /root/reference ("parity unpinned" for their constants); these are defined here and
// documented in DESIGN.md.  Arithmetic order is part of the definition (the device functors
// in dynamichmc.jl_amd/csrc/targets.hpp restate it).
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>
#include "../include/dhmc.h"
#include "mathops.hpp"

namespace oracle {

struct Target {
    int D = 0;
    virtual ~Target() {}
    // (ℓq, ∇ℓq) = logdensity_and_gradient(ℓ, q)
    virtual void eval(const MathOps& M, const double* q, double& lq, double* g) const = 0;
    // true when a finite q implies a finite gradient and ℓq is finite or -Inf, so the
    // gradient scan of evaluate_ℓ (src/hamiltonian.jl:205) cannot trigger
    virtual bool grad_finite_if_q_finite() const { return false; }
};

// DHMC_TARGET_STD_NORMAL: ℓ = -1/2 Σ q², ∇ℓ = -q
struct StdNormal : Target {
    explicit StdNormal(int d) { D = d; }
    void eval(const MathOps&, const double* q, double& lq, double* g) const override {
        lq = -0.5 * wave_dot(q, q, D);
        for (int i = 0; i < D; ++i) g[i] = -q[i];
    }
};

// DHMC_TARGET_DIAG_NORMAL: ℓ = -1/2 Σ prec_i (q_i - mu_i)²
struct DiagNormal : Target {
    std::vector<double> mu, prec;
    DiagNormal(int d, const double* m, const double* p) : mu(m, m + d), prec(p, p + d) { D = d; }
    void eval(const MathOps&, const double* q, double& lq, double* g) const override {
        std::vector<double> dv(D), w(D);
        for (int i = 0; i < D; ++i) {
            dv[i] = q[i] - mu[i];
            w[i] = prec[i] * dv[i];
            g[i] = -w[i];
        }
        lq = -0.5 * wave_dot(dv.data(), w.data(), D);
    }
};

// DHMC_TARGET_TRIDIAG_NORMAL: ℓ = -1/2 q'Pq with P symmetric tridiagonal (diag, off)
struct TridiagNormal : Target {
    std::vector<double> diag, off;
    TridiagNormal(int d, const double* a, const double* b) : diag(a, a + d), off(b, b + d) { D = d; }
    void eval(const MathOps&, const double* q, double& lq, double* g) const override {
        std::vector<double> Pq(D);
        for (int i = 0; i < D; ++i) {
            double t = diag[i] * q[i];
            if (i > 0) t = t + off[i - 1] * q[i - 1];
            if (i < D - 1) t = t + off[i] * q[i + 1];
            Pq[i] = t;
            g[i] = -t;
        }
        lq = -0.5 * wave_dot(q, Pq.data(), D);
    }
};

// DHMC_TARGET_DENSE_NORMAL: ℓ = -1/2 (q-μ)'P(q-μ), P symmetric (upper triangle); (Pd)_i is one fma chain over k
struct DenseNormal : Target {
    std::vector<double> mu, P;
    DenseNormal(int d, const double* m, const double* p) : mu(m, m + d), P((size_t)d * d) {
        D = d;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) P[(size_t)i * d + j] = (i <= j) ? p[(size_t)i * d + j] : p[(size_t)j * d + i];
    }
    void eval(const MathOps&, const double* q, double& lq, double* g) const override {
        std::vector<double> dv(D), Pd(D);
        for (int i = 0; i < D; ++i) dv[i] = q[i] - mu[i];
        for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc = __builtin_fma(P[(size_t)k * D + i], dv[k], acc);
            Pd[i] = acc;
            g[i] = -acc;
        }
        lq = -0.5 * wave_dot(dv.data(), Pd.data(), D);
    }
};

// DHMC_TARGET_FUNNEL (Neal): v = q_0 ~ N(0, 3²), q_i | v ~ N(0, e^v), i >= 1
//   ℓ = -v²/18 - 1/2 e^{-v} Σ q_i² - (D-1)/2 v
struct Funnel : Target {
    explicit Funnel(int d) { D = d; }
    void eval(const MathOps& M, const double* q, double& lq, double* g) const override {
        double v = q[0];
        double ev = M.exp(-v);
        std::vector<double> x(q, q + D);
        x[0] = 0.0;
        double S = wave_dot(x.data(), x.data(), D);
        double hd = 0.5 * (double)(D - 1);
        double hes = (0.5 * ev) * S;
        lq = (((v * v) * (-1.0 / 18.0)) - hes) - hd * v;
        g[0] = ((v * (-1.0 / 9.0)) + hes) - hd;
        for (int i = 1; i < D; ++i) g[i] = -(ev * q[i]);
    }
};

// DHMC_TARGET_LOGISTIC: Bernoulli-logit regression with a N(0, I) prior on β = q
//   η_n = x_n·β,  ℓ = Σ_n [y_n η_n - log(1 + e^{η_n})] - 1/2 β·β,  ∇ℓ = Xᵀ(y - σ(η)) - β
// Order: η_n is one fma chain over d ascending; with t = exp(-|η|): σ = η >= 0 ? 1/(1+t) : t/(1+t),
// log1pexp(η) = max(η, 0) + log1p(t); the sum over observations is, per block of DHMC_LOGISTIC_BLOCK observations, in
// wave order (64 interleaved partial sums of plain adds + butterfly), the blocks' sums added in ascending order; (Xᵀr)_d is, per block of DHMC_LOGISTIC_BLOCK observations, one fma chain
// over n ascending, the blocks' partial sums added in ascending order (include/dhmc.h).
struct Logistic : Target {
    int64_t N;
    std::vector<double> X, y;   // X row-major [N][D]
    Logistic(int d, int64_t n, const double* x, const double* yy) : N(n), X(x, x + n * d), y(yy, yy + n) { D = d; }
    void eval(const MathOps& M, const double* q, double& lq, double* g) const override {
        std::vector<double> r(N);
        if (sequential_sums()) {    // tolerance-test flavour (mathops.hpp): one left-to-right pass, no blocks, no fma
            double S1 = 0.0;
            for (int64_t n = 0; n < N; ++n) {
                const double* xn = &X[(size_t)n * D];
                double eta = 0.0;
                for (int d = 0; d < D; ++d) eta = eta + xn[d] * q[d];
                double t = M.exp(-std::fabs(eta));
                double sig = eta >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
                double l1pe = (eta > 0 ? eta : 0.0) + (M.det ? dhmc::det_log1p_nonneg(t) : std::log1p(t));
                r[n] = y[n] - sig;
                S1 = S1 + (y[n] * eta - l1pe);
            }
            lq = S1 - 0.5 * wave_dot(q, q, D);
            for (int d = 0; d < D; ++d) {
                double acc = 0.0;
                for (int64_t n = 0; n < N; ++n) acc = acc + X[(size_t)n * D + d] * r[n];
                g[d] = acc - q[d];
            }
            return;
        }
        double partial[64];
        double S1 = 0.0;
        for (int l = 0; l < 64; ++l) partial[l] = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            if (n != 0 && n % DHMC_LOGISTIC_BLOCK == 0) {       // a block of observations is complete (include/dhmc.h)
                const double bs = wave_tree(partial);
                S1 = n == DHMC_LOGISTIC_BLOCK ? bs : S1 + bs;
                for (int l = 0; l < 64; ++l) partial[l] = 0.0;
            }
            const double* xn = &X[(size_t)n * D];
            double eta = 0.0;
            for (int d = 0; d < D; ++d) eta = __builtin_fma(xn[d], q[d], eta);
            double t = M.exp(-std::fabs(eta));
            double sig = eta >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t);
            double l1pe = (eta > 0 ? eta : 0.0) + (M.det ? dhmc::det_log1p_nonneg(t) : std::log1p(t));
            r[n] = y[n] - sig;
            partial[n % 64] = partial[n % 64] + (y[n] * eta - l1pe);
        }
        {
            const double bs = wave_tree(partial);
            S1 = N <= DHMC_LOGISTIC_BLOCK ? bs : S1 + bs;
        }
        double S2 = wave_dot(q, q, D);
        lq = S1 - 0.5 * S2;
        for (int d = 0; d < D; ++d) {
            double tot = 0.0;
            for (int64_t n0 = 0; n0 < N; n0 += DHMC_LOGISTIC_BLOCK) {
                const int64_t n1 = n0 + DHMC_LOGISTIC_BLOCK < N ? n0 + DHMC_LOGISTIC_BLOCK : N;
                double acc = 0.0;
                for (int64_t n = n0; n < n1; ++n) acc = __builtin_fma(X[(size_t)n * D + d], r[n], acc);
                tot = n0 == 0 ? acc : tot + acc;
            }
            g[d] = tot - q[d];
        }
    }
};

// DHMC_TARGET_ALWAYS_DIVERGENT: the reference's AlwaysDivergentTest (test/test_NUTS.jl:58-73)
struct AlwaysDivergent : Target {
    explicit AlwaysDivergent(int d) { D = d; }
    void eval(const MathOps&, const double* q, double& lq, double* g) const override {
        bool allzero = true;
        for (int i = 0; i < D; ++i) {
            g[i] = 1.0;
            if (q[i] != 0.0) allzero = false;
        }
        lq = allzero ? 0.0 : -INFINITY;
    }
};

}  // namespace oracle
