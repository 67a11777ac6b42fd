"""HIP sources of device functors written the way a CALLER of the library writes them (dhmc_register_target_source): test
inputs for tests/test_user_functor.py and tests/test_gpu_user_functor.py."""

# a diagonal normal, written independently of csrc/targets.hpp DiagNormalT but to the same arithmetic: bit-equal results
DIAG_NORMAL = r"""
namespace dhmc {
struct MyDiagNormal {
    static constexpr bool kDeferred = true;                  // eval returns the lane's partial sum; finish() makes ℓ of the total
    static constexpr bool kElementwise = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kRecomputeGrad = true;             // ∇ℓ is cheap: proposals keep q only
    static constexpr bool kBigDims = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    const double* mu;
    const double* prec;
    int Dpad;
    __device__ explicit MyDiagNormal(const TargetParams& p) : mu(p.a), prec(p.a + p.n / 2), Dpad(p.Dpad) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        LaneAcc<1, NPL> acc;                                 // the ABI's summation order (wave.hpp)
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;                   // lane l holds coordinates l, l + 64, ...
            const bool in = e < D;                           // the caller's parameter arrays are NOT padded
            const double d = q[k] - (in ? mu[e] : 0.0);
            const double w = (in ? prec[e] : 0.0) * d;
            acc.add(0, k, d, w);
            g[k] = -w;
        }
        return acc.fold(0);
    }
    __device__ __forceinline__ double finish(double s) const { return -0.5 * s; }
};
}
"""

# a model no built-in family covers: independent Student-t(ν) coordinates with scales s_i,
#   ℓ(q) = -(ν+1)/2 Σ log(1 + (q_i/s_i)²/ν),   ∇ℓ_i = -(ν+1) q_i / (ν s_i² + q_i²)
STUDENT_T = r"""
namespace dhmc {
struct StudentT {
    static constexpr bool kDeferred = true;
    static constexpr bool kElementwise = true;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kBigDims = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    double nu;
    const double* scale;
    __device__ explicit StudentT(const TargetParams& p) : nu(p.a[0]), scale(p.a + 1) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = lane + WAVE * k;
            if (e < D) {
                const double s = scale[e], x = q[k];
                const double den = nu * (s * s) + x * x;
                part = part + det_log(den / (nu * (s * s)));         // the ABI's deterministic log (dhmc_detmath.h)
                g[k] = -((nu + 1.0) * x) / den;
            } else {
                g[k] = 0.0;
            }
        }
        return part;
    }
    __device__ __forceinline__ double finish(double s) const { return -0.5 * (nu + 1.0) * s; }
};
}
"""

BROKEN = r"""
namespace dhmc {
struct Broken {
    static constexpr bool kDeferred = true;
    __device__ explicit Broken(const TargetParams& p) {}
    template <int NPL>
    __device__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const { return undeclared_thing; }
};
}
"""


# The reference's "mixture of two normals" (test/sample-correctness_tests.jl:89-98): ℓ = log(α N(q; 0, I) + (1-α) N(q; μ₂, L₂L₂ᵀ)) in
# three dimensions, as LogDensityTestSuite's mix(α, ℓ₁, ℓ₂) of two NORMALISED densities.  Parameters: [α, c₂ = -log|det L₂|,
# μ₂ (3), P₂ = (L₂L₂ᵀ)⁻¹ row-major (9)].  Every lane computes the two quadratic forms from the three coordinates (lanes 0..2
# of slot 0); the scalar math is the ABI's (logaddexp, exp, log of dhmc_detmath.h).
MIXTURE3 = r"""
namespace dhmc {
struct Mixture3 {
    static constexpr bool kDeferred = false;                 // eval returns ℓ itself
    static constexpr bool kElementwise = false;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kBigDims = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = true;
    double la1, la2, c2, mu[3], P[9];
    __device__ explicit Mixture3(const TargetParams& p) {
        la1 = det_log(p.a[0]); la2 = det_log(1.0 - p.a[0]); c2 = p.a[1];
        for (int i = 0; i < 3; ++i) mu[i] = p.a[2 + i];
        for (int i = 0; i < 9; ++i) P[i] = p.a[5 + i];
    }
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        double x[3], d[3], Pd[3];
        for (int i = 0; i < 3; ++i) { x[i] = readlane_f64(q[0], i); d[i] = x[i] - mu[i]; }
        double q1 = 0.0, q2 = 0.0;
        for (int i = 0; i < 3; ++i) {
            Pd[i] = P[3 * i] * d[0] + P[3 * i + 1] * d[1] + P[3 * i + 2] * d[2];
            q1 += x[i] * x[i];
            q2 += d[i] * Pd[i];
        }
        const double l1 = la1 - 0.5 * q1, l2 = (la2 + c2) - 0.5 * q2;
        const double l = det_logaddexp(l1, l2);
        const double w1 = det_exp(l1 - l), w2 = det_exp(l2 - l);
#pragma unroll
        for (int k = 0; k < NPL; ++k) g[k] = 0.0;
        double gl = 0.0;
        for (int i = 0; i < 3; ++i) gl = (lane == i) ? -(w1 * x[i]) - w2 * Pd[i] : gl;
        g[0] = gl;
        return l;
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};
}
"""


# The reference's "heavier tails and skewness" cases (test/sample-correctness_tests.jl:100-118) transform a standard normal with
# LogDensityTestSuite's elongate / shift / funnel, whose source is not under /root/reference ("parity unpinned").  The definitions
# used HERE, written down so that the exact samplers of the tests and these densities agree:
#   shift(b)      x ↦ x + b
#   elongate(k)   x ↦ x ‖x‖^(k-1)          (‖y‖ = ‖x‖^k: k > 1 stretches the tails; Jacobian determinant k ‖x‖^((k-1) D))
#   funnel()      x ↦ (x₀, e^{x₀/2} x₁, …)  (Neal's funnel with a unit-variance neck coordinate)
#   mix(α, ℓ₁, ℓ₂) = log(α e^{ℓ₁} + (1-α) e^{ℓ₂}) of normalised densities
# ELONGATED:  y = elongate(k)(x + b), x ~ N(0, I):  with s = ‖y‖, a = 1/k - 1, u = y s^a (= x + b):
#   ℓ(y) = -½ ‖u - b‖² - ((k-1) D / k) log s - log k,   ∇ℓ = -[s^a (u-b) + a s^(a-2) y (y·(u-b))] - ((k-1) D / k) y / s²
# Parameters: [k, b₀ … b_{D-1}].  D <= 64 (one slot per lane).
ELONGATED = r"""
namespace dhmc {
struct Elongated {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kBigDims = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = false;
    double k;
    const double* b;
    __device__ explicit Elongated(const TargetParams& p) : k(p.a[0]), b(p.a + 1) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        static_assert(NPL == 1, "Elongated: at most 64 coordinates");
        const double y = q[0];
        const double be = lane < D ? b[lane] : 0.0;
        const double s2 = wave_allreduce1(y * y);
        const double a = 1.0 / k - 1.0, c1 = (k - 1.0) * (double)D / k;
        const double ls = 0.5 * det_log(s2);                       // log s
        const double sa = det_exp(a * ls);                         // s^a
        const double d = y * sa - be;                              // u - b
        double r[2] = {d * d, y * d};
        wave_allreduce<2>(r);
        g[0] = -(sa * d + a * (sa / s2) * y * r[1]) - c1 * y / s2;
        if (lane >= D) g[0] = 0.0;
        return -0.5 * r[0] - c1 * ls - det_log(k);
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};
}
"""

# FUNNEL_MIX:  mix(α, funnel()(N(0, I_D)), N(0, I_D)), parameters [α]; with S = Σ_{i>=1} y_i² (the shared -D/2 log 2π dropped):
#   ℓ_f = -½ y₀² - ½ e^{-y₀} S - ((D-1)/2) y₀,   ℓ_n = -½ (y₀² + S),   ℓ = logaddexp(log α + ℓ_f, log(1-α) + ℓ_n)
FUNNEL_MIX = r"""
namespace dhmc {
struct FunnelMix {
    static constexpr bool kDeferred = false;
    static constexpr bool kElementwise = false;
    static constexpr bool kPointwiseGrad = false;
    static constexpr bool kRecomputeGrad = true;
    static constexpr bool kBigDims = false;
    static constexpr bool kFiniteLqImpliesFiniteQ = true;
    static constexpr bool kFiniteLqImpliesFiniteGrad = false;
    double la, lb;
    __device__ explicit FunnelMix(const TargetParams& p) : la(det_log(p.a[0])), lb(det_log(1.0 - p.a[0])) {}
    template <int NPL>
    __device__ __forceinline__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const {
        static_assert(NPL == 1, "FunnelMix: at most 64 coordinates");
        const double y = q[0];
        const double v = readlane_f64(y, 0);
        const double S = wave_allreduce1(lane == 0 ? 0.0 : y * y);
        const double ev = det_exp(-v), hd = 0.5 * (double)(D - 1);
        const double lf = la + (-0.5 * v * v - 0.5 * ev * S - hd * v);
        const double ln = lb + (-0.5 * (v * v + S));
        const double l = det_logaddexp(lf, ln);
        const double wf = det_exp(lf - l), wn = det_exp(ln - l);
        const double gf = lane == 0 ? (-v + 0.5 * ev * S - hd) : -(ev * y);
        g[0] = lane < D ? wf * gf + wn * (-y) : 0.0;
        return l;
    }
    __device__ __forceinline__ double finish(double s) const { return s; }
};
}
"""
