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
