/* dhmc.h — C ABI of the MI355X many-chain NUTS hot path (libdhmc_amd.so).
 *
 * Drop-in boundary for the hot path of tpapp/DynamicHMC.jl v3.6.0 (paths below are relative
 * to the reference checkout).  The reference runs ONE chain per call of `mcmc_with_warmup`
 * and crosses no FFI; the natural cut for a device implementation is seam (iv) of SURVEY.md
 * §8(b): `warmup(sampling_logdensity, stage, warmup_state)` (src/mcmc.jl:99,134,258) and the
 * two per-draw loops (src/mcmc.jl:271-280 and :374-379).  One ABI call here does what one of
 * those loops does, for all `chains` chains of the context at once; everything below the loop
 * (sample_tree NUTS.jl:232, sample_trajectory trees.jl:283, adjacent_tree trees.jl:231,
 * leapfrog hamiltonian.jl:273, logdensity hamiltonian.jl:251, kinetic_energy :103,
 * combine_turn_statistics NUTS.jl:132, combine_proposals NUTS.jl:51, adapt_stepsize
 * stepsize.jl:147) runs inside HIP kernels.
 *
 * Conventions
 *  - All floating point is IEEE binary64, as in the reference (mcmc.jl:230,238; NUTS.jl:210).
 *  - Arrays are chain-major and unpadded: a [C][N][D] array holds chain c, draw n at
 *    offset (c*N + n)*D, i.e. Julia's column-major posterior_matrix[D,N] per chain
 *    (mcmc.jl:230,275,376) with a leading chain dimension.
 *  - Every buffer passed in is owned by the caller.  `*_on_device` says whether a pointer is a
 *    device (HIP) pointer or a host pointer; host buffers are staged by the library.
 *  - Every function returns a DHMC_* code; no C++ exception crosses the boundary.  Conditions
 *    the reference signals with `throw` for ONE chain are recorded in that chain's status word
 *    (dhmc_get_status) and the call returns DHMC_ERR_CHAIN_FAILURE; the host wrapper re-raises
 *    them as DynamicHMCError (utilities.jl:17-27).  Argument violations that the reference
 *    rejects with @argcheck return DHMC_ERR_INVALID_ARGUMENT.
 *  - A context is confined to one host thread at a time; distinct contexts (one per GPU) may be
 *    driven concurrently.  Calls are synchronous with respect to host output buffers.
 *
 * Random streams (the reference draws from a caller-supplied AbstractRNG in the order
 * p -> directions -> tree draws, NUTS.jl:232-233; no test pins a stream).  The ABI fixes a
 * counter-based stream so that results do not depend on how chains are scheduled or sharded:
 *   Philox4x32-10, key = (seed[31:0], global chain index),
 *   counter = (index, purpose, transition number of that chain, seed[63:32]).
 *   purpose 0: momentum normals.  Call `index` = lane + 64*kk yields the N(0,1) pair for
 *              coordinates e0 = (index%64) + 128*(index/64) and e1 = e0 + 64 by Box–Muller
 *              (dhmc_detmath.h det_randn2, r1 = words[1]<<32|words[0], r2 = words[3]<<32|words[2]);
 *              p = W .* z as rand_p (hamiltonian.jl:124).
 *   purpose 1: directions = word 0 of call index 0 (trees.jl:23).
 *   purpose 2: the k-th Exp(1) draw consumed by the transition (rand_bool_logprob,
 *              NUTS.jl:43-45; only drawn when logprob < 0), det_randexp(r1) of call index k,
 *              in the depth-first post-order of trees.jl:231-262.
 *   purpose 3: momentum for the initial step size search (mcmc.jl:139), as purpose 0.
 *   purpose 4: initial position U[-2,2)^D (mcmc.jl:108): call index j yields coordinates
 *              e0,e1 as for purpose 0, q = u01_closed_open(r)*4 - 2.
 *   purpose 5: momenta of the leapfrog probes (Diagnostics, diagnostics.jl:147,216), as purpose 0 with
 *              the momentum's index in the `transition` word.
 * A Julia `TapeRNG <: AbstractRNG` that replays this stream makes the real reference
 * reproduce these draws.
 *
 * Summation order (LinearAlgebra.dot, hamiltonian.jl:103 and NUTS.jl:130-137, is BLAS-dependent and unpinned; the ABI
 * fixes one so that CPU checker and device agree bit for bit).  Every dot product over the D coordinates of a chain
 * is computed as follows.  The row, zero padded to Dpad = 64*2^j >= D, is cut into BLOCKS of 256 coordinates.  In
 * block B, partial sum (B, l), l = 0..63, accumulates the products of coordinates 256 B + l, + 64, + 128, + 192 in
 * that order with fma, starting from +0.  Per l, the blocks' partial sums are combined by an adjacent-pairs binary
 * tree ((B0 + B1) + (B2 + B3)) ...; the 64 resulting values are combined by the xor butterfly 1, 2, 4, 8, 16, 32
 * (adjacent pairs first).  For D <= 256 this is one fma chain per l and the butterfly.  A block is what one 64-lane
 * wavefront holds at four coordinates per lane: chains of 512+ coordinates are spread over one wavefront per block.
 *
 * Engines and the environment.  A context has several engines for the per-draw loops (a wavefront per chain; several
 * chains per wavefront; four wavefronts per chain; round engines whose products are GEMMs), all producing the SAME bits; which one runs
 * a dhmc_run call is chosen from the chain count, the GPU's size and three numbers the previous call left behind (mean tree size, whether
 * its slowest chain held it open, the chains' order of work) with the thresholds of `EnginePolicy` in csrc/capi_run.hip.  The choice can
 * be overridden, for tests and timing only, by these environment variables, read when a context is created (none changes a result):
 *   DHMC_PACKED, DHMC_PIPELINE = 1 | 0     always | never the packed / the pipeline kernel
 *   DHMC_PK = "key=value,…"                the packed launch: cpl (coordinates per lane, 2 | 4), align (the gate, a power of two),
 *                                          lds_levels, max_waves, queue (0 | 1), handover (live lane groups at which the end game starts,
 *                                          0: never), many_chains (chain count from which a tail-bound launch stays packed)
 *   DHMC_DENSE = "key=value,…"             the dense engines: rounds (1: GEMM rounds, 0: a matvec per chain), products (1 | 2 M⁻¹ products
 *                                          per leapfrog; also dhmc_set_dense_products), parts (half-batches, 1–4), row_lists, k3_block,
 *                                          fuse_k2 (0 | 1)
 *   DHMC_LOGISTIC_ROUNDS, DHMC_L1_LDS, DHMC_LAUNCH_ORDER = 1 | 0;  DHMC_HOST_CHUNK = transitions per chunk of a call with host outputs;
 *   DHMC_FORCE_NPL = slots per lane (a narrow chain through the wide kernels);  DHMC_ESS_LONG (dhmc_ess_*: the long-chain path);
 *   DHMC_DEBUG_ORDER (prints the engine and launch order of every call);  DHMC_RTC_CACHE = directory (dhmc_register_target_source);
 *   DHMC_ACCEPT_UNVERSIONED_STATE (dhmc_import_state: blobs written before the version word existed).
 * The pipeline kernel's four wavefronts hand records over through LDS counters WITHOUT memory fences: it relies on the LDS executing one
 * wavefront's operations in issue order (true of every GCN / CDNA part; not an architected guarantee).  Every wait is bounded, and a
 * wait that runs out raises DHMC_ST_KERNEL_PROTOCOL; building csrc with -DDHMC_PIPE_FENCED puts a fence beside every publication.
 */
#ifndef DHMC_H
#define DHMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes -------------------------------------------------------------------- */
#define DHMC_OK 0
#define DHMC_ERR_INVALID_ARGUMENT 1 /* ArgumentError of @argcheck: NUTS.jl:190-191, stepsize.jl:31-33,108-111,135,148, mcmc.jl:191-192, hamiltonian.jl:63,146-147 */
#define DHMC_ERR_HIP 2              /* a HIP runtime call failed; see dhmc_last_error */
#define DHMC_ERR_UNSUPPORTED 3      /* valid request outside what this build implements */
#define DHMC_ERR_CHAIN_FAILURE 4    /* >=1 chain hit a reference `throw` site; see dhmc_get_status */
#define DHMC_ERR_NO_DEVICE 5        /* no HIP device / kernels unavailable: there is NO CPU fallback */
#define DHMC_ERR_CALLBACK 6         /* the DHMC_TARGET_EXTERNAL callback is missing or returned non-zero */

/* ---- per-chain status bits (reference throw sites) ----------------------------------- */
#define DHMC_ST_NONFINITE_POSITION 1u      /* hamiltonian.jl:203 "Position vector has non-finite elements." */
#define DHMC_ST_INVALID_INITIAL 2u         /* hamiltonian.jl:212-216 strict evaluate_ℓ at the initial point */
#define DHMC_ST_STEPSIZE_SEARCH_FAILED 4u  /* stepsize.jl:57-59 */
#define DHMC_ST_NONFINITE_START_DENSITY 8u /* stepsize.jl:77-79 */
#define DHMC_ST_KERNEL_PROTOCOL 0x40000000u /* INTERNAL error, never a model's fault: a bounded wait inside a kernel ran out (the pipeline kernel's
                                             * wavefront handshake; the packed kernel's trip limit).  The chain's results of the call are incomplete. */

/* ---- target family: the device-side stand-in for LogDensityProblems.logdensity_and_gradient
 *      (hamiltonian.jl:204; capabilities/dimension checks hamiltonian.jl:146-147) ---------- */
#define DHMC_TARGET_STD_NORMAL 0  /* l(q) = -1/2 sum q^2, grad = -q.            params: none */
#define DHMC_TARGET_DIAG_NORMAL 1 /* l = -1/2 sum prec_i (q_i-mu_i)^2.          params: double mu[D], prec[D] */
#define DHMC_TARGET_TRIDIAG_NORMAL 2 /* l = -1/2 q'Pq, P symmetric tridiagonal.  params: double diag[D], off[D] (off[D-1] ignored) */
#define DHMC_TARGET_FUNNEL 3      /* Neal's funnel: v=q_0~N(0,3^2), q_i|v~N(0,e^v). params: none */
#define DHMC_TARGET_LOGISTIC 4    /* Bernoulli-logit regression, N(0,I) prior.  params: int64 n; double X[n][D]; double y[n] */
/* Order of the logistic gradient's sum over the observations, (Xᵀr)_d = Σ_n X[n][d] r_n: the observations are cut into
 * BLOCKS of DHMC_LOGISTIC_BLOCK; inside a block one fma chain over n ascending from +0; the blocks' partial sums are added
 * in ascending order, ((B0 + B1) + B2) + ...  (For n <= DHMC_LOGISTIC_BLOCK: one chain.)  A block is the K-range one
 * workgroup of the split-K product R·X owns; without the blocks every output element is ONE dependent chain of n fma's,
 * whose length — not the number of chains still running — sets the time of a leapfrog round.
 * The log-likelihood Σ_n [y_n η_n − log(1 + e^{η_n})] is summed over the same blocks: inside a block in wave order (the
 * terms of the observations n = l mod 64 added in ascending order from +0 for l = 0..63, then the xor butterfly), the
 * blocks' sums added in ascending order. */
#ifndef DHMC_LOGISTIC_BLOCK
#define DHMC_LOGISTIC_BLOCK 2048
#endif
#define DHMC_TARGET_DENSE_NORMAL 6 /* l = -1/2 (q-mu)'P(q-mu), P full symmetric (read from its upper triangle). params: double mu[D], P[D][D] */
#define DHMC_TARGET_EXTERNAL 7     /* the caller's own model: l and grad come from a callback evaluated for all chains at once
                                    * (dhmc_set_logdensity_callback); dim <= 4096, diagonal or shared dense metric. params: none */
/* Dimension limits.  dim <= 1024: every family, every metric (register/LDS-resident kernels, GEMM round engines).
 * 1024 < dim <= 4096: DHMC_TARGET_EXTERNAL, a caller's device functor (round 5: its eval() compiled into a batched evaluation kernel)
 * and every built-in family (STD / DIAG / TRIDIAG / DENSE normal, funnel, logistic regression, the always-divergent test
 * density), diagonal or shared dense metric, through the streaming round engine: the library evaluates the density of all
 * chains itself where the callback would stand — the normal families and the funnel in one kernel, the dense normal as one
 * product (q − μ)·P over the chains, the logistic gradient as its two GEMMs over the chains — with the families' arithmetic
 * (dhmc_set_logdensity_callback is then DHMC_ERR_INVALID_ARGUMENT).  dim > 4096 and per-chain dense metrics beyond 1024:
 * DHMC_ERR_UNSUPPORTED. */
#define DHMC_TARGET_USER_BASE 1000  /* + the handle dhmc_register_target_source returned: the caller's own DEVICE FUNCTOR, compiled at run
                                    * time into the library's own per-draw, initialisation and step-size-search kernels — no host round
                                    * trip per leapfrog, the same kernels the built-in families run.  Diagonal or shared dense metric; dim <= 1024 inside the per-draw
                                    * kernels, 1024 < dim <= 4096 evaluated for all chains between the streaming round engine's kernels.
                                    * params: any number of doubles, handed to the functor's constructor (TargetParams::a, n = count). */
#define DHMC_TARGET_ALWAYS_DIVERGENT 5 /* the reference's fault-injection double (test/test_NUTS.jl:58-73): l = 0 at the origin, -Inf elsewhere, grad = ones. params: none */

/* ---- kinetic energy (GaussianKineticEnergy, hamiltonian.jl:56-87) -------------------- */
#define DHMC_METRIC_DIAG 0  /* per-chain diagonal M^-1 [C][D]; the unit metric (hamiltonian.jl:87) is diag of ones */
#define DHMC_METRIC_DENSE 1 /* one dense M^-1 [D][D] shared by all chains of the context */

typedef struct dhmc_ctx dhmc_ctx;

typedef struct dhmc_config {
    int32_t device;        /* HIP device ordinal */
    int32_t dim;           /* D = LogDensityProblems.dimension(l) */
    int32_t chains;        /* C chains owned by this context */
    int32_t chain_offset;  /* global index of chain 0 (RNG key); shards of one job use disjoint ranges */
    int32_t metric;        /* DHMC_METRIC_* */
    int32_t target;        /* DHMC_TARGET_* */
    const void* target_params; /* host pointer, copied at create */
    uint64_t target_params_bytes;
    int32_t max_depth;     /* NUTS.max_depth, 0 < max_depth <= 32 (NUTS.jl:190); default 10 (:166) */
    int32_t dense_per_chain; /* DHMC_METRIC_DENSE only.  0: one M^-1 shared by all chains, adapted from their pooled draws (the
                              * GEMM engines; a stated deviation).  1: one M^-1 [D][D] PER CHAIN, each adapted from that chain's own
                              * draws — the reference's semantics (mcmc.jl:281-285); chains stream their own matrix (wave-per-chain
                              * kernels), so this is for small D * chains (2 * chains * Dpad^2 doubles of HBM). */
    double min_delta;      /* NUTS.min_Δ < 0 (NUTS.jl:191); default -1000 */
    uint64_t seed;
} dhmc_config;

/* InitialStepsizeSearch (stepsize.jl:23-36) */
typedef struct dhmc_stepsize_search {
    double initial_eps;       /* > 0, default 0.1 */
    double log_threshold;     /* finite, < 0, default log(0.8) */
    int32_t maxiter_crossing; /* >= 50, default 400 */
    int32_t reserved;
} dhmc_stepsize_search;

/* DualAveraging (stepsize.jl:98-118) plus how one call maps onto a TuningNUTS stage
 * (mcmc.jl:266 initial_adaptation_state at stage start; :285 final_ϵ at stage end). */
typedef struct dhmc_dual_averaging {
    double delta;  /* 0 < δ < 1, default 0.8 */
    double gamma;  /* γ > 0, default 0.05 */
    double kappa;  /* 0.5 < κ <= 1, default 0.75 */
    int32_t t0;    /* t₀ >= 0, default 10 */
    int32_t init;     /* !=0: (re)initialise the adaptation state from the current ε before the first transition */
    int32_t finalize; /* !=0: set ε := final_ϵ = exp(logϵ̄) after the last transition */
    int32_t reserved;
} dhmc_dual_averaging;

/* Caller-owned result buffers of one dhmc_run call; any pointer may be NULL (not recorded).
 * Fields mirror what the per-draw loops store (mcmc.jl:275-277,376-377) and
 * TreeStatisticsNUTS (NUTS.jl:208-221). */
typedef struct dhmc_outputs {
    int32_t on_device;       /* !=0: all non-NULL pointers are device pointers */
    int32_t reserved;
    double* draws;           /* [C][N][D] posterior_matrix[:, i] = Q.q */
    double* logdensities;    /* [C][N]    Q.ℓq */
    double* eps;             /* [C][N]    ϵs[i] (warmup only in the reference, mcmc.jl:273) */
    double* pi;              /* [C][N]    TreeStatisticsNUTS.π = logdensity(H, ζ) */
    double* acceptance_rate; /* [C][N] */
    int64_t* steps;          /* [C][N] */
    int64_t* term_left;      /* [C][N]    InvalidTree.left  (trees.jl:180-202; (1,0) = REACHED_MAX_DEPTH) */
    int64_t* term_right;     /* [C][N]    InvalidTree.right */
    int32_t* depth;          /* [C][N] */
    uint32_t* directions;    /* [C][N]    Directions.flags as drawn */
} dhmc_outputs;

/* ---- page-locked host memory for result buffers ---------------------------------------- */
/* dhmc_run with host pointers copies every output field device -> host.  Into PAGE-LOCKED memory those copies run at PCIe
 * speed and under the next chunk's kernels (a call with more than ≈ 1–2 GiB of draws leaves in chunks of transitions); into
 * pageable memory the runtime stages them at a fraction of that.  These two calls hand out / take back page-locked
 * memory (hipHostMalloc) for callers that cannot allocate it themselves; any page-locked memory works the same. */
int dhmc_host_alloc(void** out, uint64_t nbytes);
int dhmc_host_free(void* p);

/* ---- the caller's log density as a device functor (LogDensityProblems.logdensity_and_gradient, hamiltonian.jl:204) ------
 * hip_source: HIP C++ that defines, in namespace dhmc, a struct `functor_name` with the interface of the built-in families
 * (csrc/targets.hpp; INTEGRATION.md §4 shows one in full):
 *     static constexpr bool kDeferred, kElementwise, kPointwiseGrad, kRecomputeGrad, kBigDims (= false),
 *                           kFiniteLqImpliesFiniteQ, kFiniteLqImpliesFiniteGrad;
 *     __device__ explicit functor_name(const TargetParams& p);           // p.a: the context's params (doubles, device), p.n: how many
 *     template <int NPL> __device__ double eval(const double (&q)[NPL], double (&g)[NPL], int lane, int D) const;
 *     __device__ double finish(double s) const;                          // ℓ from the wave-reduced sum when kDeferred
 * Lane l holds coordinates l, l+64, … (slot k <-> coordinate l + 64 k; pads are 0); eval fills g = ∇ℓ(q) and returns ℓ (or the
 * lane's partial sum of it).  A kDeferred functor's partial sum must be ±0 in every lane whose slots are all pads (lanes >= D of a
 * chain of at most 64 coordinates): such chains are reduced over their first 16 / 32 lanes only (csrc/wave.hpp wave_allreduce), so
 * a constant term of ℓ belongs in finish() or in a lane that holds a coordinate, never in a pad lane.  The source is compiled with hiprtc (-O3 -ffp-contract=off, as the library itself) against the
 * library's kernel templates when a context is created with target = DHMC_TARGET_USER_BASE + *target_handle; compile errors come
 * back from dhmc_create as DHMC_ERR_INVALID_ARGUMENT with the compiler's log in dhmc_target_source_log().  dim <= 1024 (beyond, up to 4096: only eval() is compiled, into functor_eval_kernel, and the streaming
 * round engine calls it where an external model's callback would stand); diagonal
 * metric (the wave-per-chain kernels) or the shared dense metric (a second module, compiled when the first dense context of the
 * functor is created: the GEMM round engine's kernels around the functor, and the wave-per-chain dense kernels for small batches).
 * dhmc_check_target_source compiles only (no device needed) the kernels of `metric` for `dim` coordinates and returns the log.
 * Environment DHMC_RTC_CACHE=<directory>: compiled modules are kept there (keyed by source, functor name, chain width, metric,
 * GPU architecture, library and hiprtc version) and loaded instead of compiled by later processes; an unreadable file, or one
 * whose module does not load on this device, is ignored and replaced. */
int dhmc_register_target_source(const char* hip_source, const char* functor_name, int32_t* target_handle);
int dhmc_check_target_source(const char* hip_source, const char* functor_name, int32_t dim, int32_t metric, char* log, uint64_t log_bytes);
const char* dhmc_target_source_log(void);   /* the log of the last run-time compilation in this process */

/* ---- lifecycle ----------------------------------------------------------------------- */
int dhmc_create(const dhmc_config* cfg, dhmc_ctx** out);
int dhmc_destroy(dhmc_ctx* ctx);
/* hipStream_t to launch on (NULL = the default stream). */
int dhmc_set_stream(dhmc_ctx* ctx, void* hip_stream);
const char* dhmc_last_error(const dhmc_ctx* ctx);
const char* dhmc_version(void);
/* DHMC_DETMATH_VERSION of the library (include/dhmc_detmath.h): the numerical contract — host code compiled against another version of
 * that header disagrees with the device in the last bits; dhmc_export_state stamps it into the blob and dhmc_import_state refuses a
 * blob of another version. */
int dhmc_detmath_version(void);

/* ---- warmup state: WarmupState(Q, κ, ϵ) per chain (mcmc.jl:72-79) -------------------- */
/* initialize_warmup_state (mcmc.jl:129-132): q0 [C][D], or NULL for random_position
 * (mcmc.jl:108); evaluates l strictly (hamiltonian.jl:202-217, strict=true), sets the per-chain diagonal κ to the
 * unit metric and ε to "unspecified" (NaN).  Clears status words and transition counters.  A DHMC_METRIC_DENSE
 * context KEEPS its shared M⁻¹ (the identity at dhmc_create, or what dhmc_set_metric_dense / dhmc_update_metric_dense
 * last installed): initialize_warmup_state takes κ as a keyword (mcmc.jl:129), and one matrix serves all chains, so
 * it is set once, before or after the chains are placed. */
int dhmc_init(dhmc_ctx* ctx, const double* q0, int q0_on_device);
/* Q := evaluate_ℓ(ℓ, q) for every chain at positions of the caller's choosing (strict, as dhmc_init: hamiltonian.jl:202-217)
 * WITHOUT touching κ, ϵ, the adaptation state or the random-stream counters — what `mcmc_next_step(steps, Q)` (mcmc.jl:348-351)
 * needs when it is handed a Q that is not the context's own.  q: [C][D]. */
int dhmc_set_position(dhmc_ctx* ctx, const double* q, int on_device);

/* q [C][D], lq [C], grad [C][D]; any may be NULL. */
int dhmc_get_position(dhmc_ctx* ctx, double* q, double* lq, double* grad, int on_device);
/* GaussianKineticEnergy(Diagonal(minv)) (hamiltonian.jl:80): minv [C][D] if per_chain else [D]. */
int dhmc_set_metric_diag(dhmc_ctx* ctx, const double* minv, int per_chain, int on_device);
int dhmc_get_metric_diag(dhmc_ctx* ctx, double* minv, int on_device); /* [C][D] */
/* GaussianKineticEnergy(M⁻¹) dense (hamiltonian.jl:73), contexts created with DHMC_METRIC_DENSE only:
 * minv [D][D], read as Symmetric(minv) from its upper triangle, shared by all chains;
 * W = cholesky(inv(M⁻¹)).L is built by the library on the device (element-wise operation order fixed by the ABI,
 * csrc/dense_factor.hpp); with dense_per_chain = 1 every chain receives a copy.
 * Returns DHMC_ERR_INVALID_ARGUMENT if the matrix is not positive definite. */
int dhmc_set_metric_dense(dhmc_ctx* ctx, const double* minv, int on_device);
/* host [D][D] each, either may be NULL: the symmetrised M⁻¹ and the lower-triangular W (W Wᵀ = M); with
 * dense_per_chain = 1 those of chain 0 — dhmc_get_metric_dense_chain returns any chain's. */
int dhmc_get_metric_dense(dhmc_ctx* ctx, double* minv, double* W);
int dhmc_get_metric_dense_chain(dhmc_ctx* ctx, int32_t chain, double* minv, double* W);
/* How many products M⁻¹·v a leapfrog with the SHARED dense metric takes.
 *   2: the reference's recurrence as written — ∇kinetic_energy(κ, pₘ) (hamiltonian.jl:278) and p♯ = M⁻¹p′ of the new point
 *      (hamiltonian.jl:103, NUTS.jl:121).  4·D² flops per leapfrog.
 *   1: the same map with one product: a chain carries u = M⁻¹∇ℓq next to p♯ = M⁻¹p; M⁻¹pₘ = p♯ + (ϵ/2)u (no product),
 *      u′ = M⁻¹∇ℓq′ (the product), p♯′ = M⁻¹pₘ + (ϵ/2)u′.  p♯ and u of a transition's initial point are fresh products, so
 *      rounding does not accumulate across transitions.  2·D² flops per leapfrog.  Every elementwise step is a separate
 *      multiply and add in the order written.  The DEFAULT of a shared dense metric (a stated deviation from the
 *      reference's operation order, like the pooled adaptation: per-step energies agree to ≈1e-12, trees are the same —
 *      tests/test_gpu_tolerance.py).  Both dense engines — the wave-per-chain kernel (up to 128 coordinates, and up to 256 below 2048
 *      chains) and the GEMM rounds (otherwise) — run either recurrence with the same bits.
 * dense_per_chain contexts — the reference's semantics — default to 2 and take 1 on request.  May be changed between
 * dhmc_run calls.  DHMC_DENSE="products=1|2" in the environment sets the default of every dense context. */
int dhmc_set_dense_products(dhmc_ctx* ctx, int32_t products);
int dhmc_get_dense_products(const dhmc_ctx* ctx);   /* 1 or 2; 0 for a context without a dense metric */
/* eps [C] if per_chain else a single value broadcast; must be > 0 (stepsize.jl:135). */
int dhmc_set_stepsize(dhmc_ctx* ctx, const double* eps, int per_chain, int on_device);
int dhmc_get_stepsize(dhmc_ctx* ctx, double* eps, int on_device); /* [C] */
int dhmc_get_status(dhmc_ctx* ctx, uint32_t* status); /* host [C] */

/* ---- DHMC_TARGET_EXTERNAL: the downward plugin API.  The reference calls LogDensityProblems.logdensity_and_gradient(ℓ, q)
 *      (hamiltonian.jl:204) once per leapfrog per chain; here the callback is called once per leapfrog ROUND with the
 *      positions of all chains: q [chains][ld] (device; columns >= dim are padding), and must put ℓ(q_c) into
 *      lq [chains] and ∇ℓ(q_c) into grad [chains][ld] (device; padding columns are ignored), enqueued on `stream`
 *      (a hipStream_t) or finished when it returns.  Non-finite outputs are legal: evaluate_ℓ's rules
 *      (hamiltonian.jl:202-217) are applied to them.  Return 0; anything else aborts the call with DHMC_ERR_CALLBACK.
 *      Must be set before dhmc_init.  The Python host wraps a batched PyTorch function this way (api.TorchLogDensity). */
typedef int (*dhmc_logdensity_fn)(void* user, const double* q, int64_t chains, int64_t ld, int64_t dim, double* lq,
                                  double* grad, void* stream);
int dhmc_set_logdensity_callback(dhmc_ctx* ctx, dhmc_logdensity_fn fn, void* user);

/* ---- warmup(::InitialStepsizeSearch) (mcmc.jl:134-148 -> stepsize.jl:46-85) ---------- */
int dhmc_find_initial_stepsize(dhmc_ctx* ctx, const dhmc_stepsize_search* params);

/* ---- the per-draw loops (mcmc.jl:271-280 with `da`, :374-379 with da == NULL); `out` may be NULL (nothing recorded) -------- */
int dhmc_run(dhmc_ctx* ctx, int64_t n_transitions, const dhmc_dual_averaging* da,
             const dhmc_outputs* out);

/* ---- end-of-stage metric update (mcmc.jl:209-223,281-284): κ := GaussianKineticEnergy(
 *      regularize_M⁻¹(sample_M⁻¹(Diagonal, posterior_matrix), λ)) per chain from that chain's
 *      own draws [C][N][D] (device pointer if on_device).  regularize is the identity for
 *      Diagonal (mcmc.jl:223); lambda is accepted for signature parity. */
int dhmc_update_metric_diag(dhmc_ctx* ctx, const double* draws, int64_t n, double lambda,
                            int on_device);

/* The same update WITHOUT the draws (a tuning stage's posterior matrix is [D][N] per chain in the reference, mcmc.jl:267 — for 8192
 * chains × 1000 coordinates the five default metric windows are 25 GB that would be written only to be read once):
 *   dhmc_metric_window_begin   opens a window: two [chains][Dpad] rows per chain (running mean, sum of squared deviations) are
 *                              zeroed, and from then on the kernels of every dhmc_run — whichever engine serves the context — add
 *                              each transition's draw to them when they store it (dhmc_detmath.h dm_window_update: Welford's
 *                              recurrence in a pinned order, so the oracle reproduces the bits);
 *   dhmc_metric_window_count   draws in the open window (-1: none open);
 *   dhmc_update_metric_diag_window   κ := GaussianKineticEnergy(Diagonal(m2 / (n - 1))) per chain and closes the window
 *                              (n >= 2, else DHMC_ERR_INVALID_ARGUMENT and the window stays open);
 *   dhmc_metric_window_end     closes a window without using it.
 * The estimate is Statistics.var's to rounding (≈ 1e-15 relative; the two-pass dhmc_update_metric_diag is the reference's own
 * summation order).  dhmc_outputs.draws may be NULL while a window is open — that is the point.  dhmc_init and
 * dhmc_import_state discard an open window; a state blob does not carry one.  Diagonal metric only. */
int dhmc_metric_window_begin(dhmc_ctx* ctx);
int64_t dhmc_metric_window_count(const dhmc_ctx* ctx);
int dhmc_update_metric_diag_window(dhmc_ctx* ctx, double lambda);
int dhmc_metric_window_end(dhmc_ctx* ctx);

/* Dense counterpart (contexts created with DHMC_METRIC_DENSE): κ := GaussianKineticEnergy(regularize_M⁻¹(
 * Symmetric(cov(pm; dims=2)), λ)) (mcmc.jl:210,218-222).  The reference estimates one matrix per chain; the
 * dense M⁻¹ of a context is shared, so the draws of all chains are pooled (J = C·n rows, chain-major) — with
 * chains == 1 this is the reference's estimator.  Returns DHMC_ERR_INVALID_ARGUMENT if the estimate is not
 * positive definite (e.g. J <= D with λ = 0); the metric then stays as it was.  With dense_per_chain = 1 every chain is
 * estimated and factorised from its own n draws and stands for itself, as in the reference: a chain whose estimate is refused
 * keeps its metric, all others are updated, and the call returns DHMC_ERR_INVALID_ARGUMENT (dhmc_last_error names how many). */
int dhmc_update_metric_dense(dhmc_ctx* ctx, const double* draws, int64_t n, double lambda, int on_device);

/* ---- Diagnostics that call the hot path directly (src/diagnostics.jl), for every chain from its current
 *      position; the chains are not modified.  All buffers are HOST pointers.  status [C] (may be NULL)
 *      receives the DHMC_ST_* bits of the reference's throw sites; any non-zero bit makes the call return
 *      DHMC_ERR_CHAIN_FAILURE (outputs are still written).  Every family and metric: inside one kernel for the
 *      device functors (built-in or the caller's); as lock-step leapfrogs of all chains around one batched
 *      evaluation per step for DHMC_TARGET_EXTERNAL (the callback is called last − first times for a trajectory,
 *      n_momenta · n_eps times for the ratios) and for the normal families beyond 1024 coordinates. ---------- */
/* leapfrog_trajectory(ℓ, q, ϵ, first:last; κ, p) (diagnostics.jl:214-227): positions first..last relative
 * to the chain's position (first <= 0 <= last, else DHMC_ERR_INVALID_ARGUMENT as the @argcheck of :218), step
 * eps forward and -eps backward, each direction tracked until the first non-finite ℓq (that point included,
 * :176-186).  p: [C][D] momenta, or NULL for p = rand_p (stream purpose 5, index momentum_index).
 * delta, logdensity: [C][npos] (Δ = logdensity(H,z) - π₀, :196; ℓq), npos = last-first+1; q_out, p_out:
 * [C][npos][D] or NULL; range: [C][2] = the positions actually visited (lo, hi).  Entries of positions
 * that were not visited are NaN. */
int dhmc_leapfrog_trajectory(dhmc_ctx* ctx, double eps, int32_t first, int32_t last, uint32_t momentum_index,
                             const double* p, double* delta, double* logdensity, double* q_out, double* p_out,
                             int32_t* range, uint32_t* status);
/* explore_log_acceptance_ratios(ℓ, q, log2ϵs; κ, N, ps) (diagnostics.jl:144-152): out [C][n_momenta][n_eps]
 * (= the reference's [n_eps, N] matrix, column-major, per chain) of the uncapped log acceptance ratios
 * local_log_acceptance_ratio(H, PhasePoint(Q, p))(eps[i]) (stepsize.jl:75-85); eps = 2.0 .^ log2ϵs is formed
 * by the caller.  ps: [C][n_momenta][D] or NULL for rand_p draws (purpose 5, indices momentum_index + m).
 * A non-finite starting Hamiltonian sets DHMC_ST_NONFINITE_START_DENSITY (stepsize.jl:77-79). */
int dhmc_explore_log_acceptance_ratios(dhmc_ctx* ctx, const double* eps, int32_t n_eps, int32_t n_momenta,
                                       uint32_t momentum_index, const double* ps, double* out, uint32_t* status);

/* ---- posterior diagnostics on the device: effective sample size and R-hat of `ncoords`
 *      coordinates of draws [chains][n][dim] held in HBM (e.g. dhmc_outputs.draws with on_device = 1), so ESS/s can
 *      be reported without moving the draws.  Estimator: multi-chain autocorrelation with Geyer's initial monotone
 *      positive sequence (what the reference's tests obtain from MCMCDiagnosticTools.ess_rhat,
 *      test/sample-correctness_utilities.jl:40-43; that package is not vendored, parity unpinned).  coords, ess,
 *      rhat are HOST arrays; stream may be NULL; n >= 4.  Series of up to 7680 draws are held in LDS; longer ones stay
 *      in HBM and their autocovariances are computed 1024 lags at a time until every coordinate's sequence has
 *      truncated — the same arithmetic order, so the same bits either way. ------------------------------------ */
int dhmc_ess_rhat(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat);
/* Bulk ESS and rank-normalised split-R̂ (Vehtari et al. 2021 — MCMCDiagnosticTools.ess_rhat's default kind): every chain
 * is split in two halves (an odd last draw is dropped), the 2·chains·(n/2) draws of a coordinate are replaced by the
 * normal scores of their average ranks, and the estimator of dhmc_ess_rhat runs on those.  Same arguments; n >= 8,
 * chains·n < 2³¹. */
int dhmc_ess_bulk(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat);
/* Tail ESS (Vehtari et al. 2021; MCMCDiagnosticTools ess(kind = :tail)): the smaller of the ESS of I(x <= q5%) and I(x >= q95%)
 * over the split chains, the quantiles (type 7) taken over all draws of the coordinate.  Same limits as dhmc_ess_bulk. */
int dhmc_ess_tail(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess);

/* Post-hoc NUTS diagnostics over the SoA tree statistics of dhmc_run (DynamicHMC.Diagnostics, diagnostics.jl:29-106), all
 * [chains][n] arrays as dhmc_outputs writes them (device pointers if on_device, so 4096 x 1000 statistics need not
 * cross PCIe): EBFMI per chain (:29-32; n >= 2), count_terminations (:65-82) and count_depths (:87-95) pooled over the
 * chains, and summarize_tree_statistics' mean and ACCEPTANCE_QUANTILES of the pooled acceptance rates (:100-106, Julia's
 * default quantile definition).  `summary` and `ebfmi` ([chains], may be NULL) are HOST buffers.  Summation orders:
 * csrc/treestat_kernels.hpp. */
typedef struct dhmc_tree_statistics_summary {
    int64_t n;                 /* chains * n: "Sample length" */
    double a_mean;
    double a_quantiles[5];     /* 0.05, 0.25, 0.5, 0.75, 0.95 */
    int64_t max_depth, divergence, turning;
    int64_t depth_counts[33];  /* depth 0 .. 32 (the reference trims trailing zeros) */
} dhmc_tree_statistics_summary;
int dhmc_summarize_tree_statistics(int32_t device, void* stream, const double* pi, const double* acceptance_rate,
                                   const int64_t* term_left, const int64_t* term_right, const int32_t* depth,
                                   int64_t chains, int64_t n, int on_device, dhmc_tree_statistics_summary* summary,
                                   double* ebfmi);

/* ---- a shared dense metric adapted from the draws of ALL ranks (mcmc.jl:210,218-222 pooled over the whole job) ----
 * Without this, dhmc_update_metric_dense pools the draws of the chains of ONE context — one GPU's block — so a job sharded over
 * 8 ranks adapts eight different matrices where one GPU holding all chains would adapt one.  With an all-reduce installed the
 * estimate is job-wide: (1) every rank sums its rows per coordinate (sequentially in chain-major row order, as before) and the
 * D sums, the row count and an error slot (D + 2 doubles) are added over the ranks; mean = sum / J_total; (2) every rank forms Σ_j (x_j - mean)(x_j - mean)ᵀ
 * over ITS rows (the same k-ordered fp64-MFMA chain as before) and the Dpad² sums are added over the ranks; then regularisation
 * with J_total and the factorisation as before — every rank ends with the same matrix.  The callback must add `count` doubles at
 * `device_buf` (device memory) over all ranks IN PLACE, ordered on `hip_stream` (RCCL: ncclAllReduce on that stream; from Python
 * torch.distributed.all_reduce — dynamichmc.jl_amd.sharding.TorchAllReduce), and return 0.  A collective's summation order is its
 * own: the result of N ranks agrees with one rank holding all chains to rounding (tests: rtol 1e-12), and is bit-identical to
 * dhmc_update_metric_dense without a callback when there is one rank.  Shared dense metric only (per-chain metrics need no
 * pooling); fn == NULL removes it.  Failure is job-wide: every rank makes both collective calls or none after the first — a rank
 * that cannot stage its draws or allocate its buffers still takes part in the first all-reduce (zeros and a raised error slot),
 * and all ranks then return DHMC_ERR_HIP together; too few rows in the whole job: DHMC_ERR_INVALID_ARGUMENT on every rank.  (A
 * callback that itself fails on one rank cannot be repaired here: the collective library's own error handling applies.) */
typedef int (*dhmc_allreduce_fn)(void* user, double* device_buf, int64_t count, void* hip_stream);
int dhmc_set_metric_allreduce(dhmc_ctx* ctx, dhmc_allreduce_fn fn, void* user);

/* ---- resume: flat POD image of every chain's (Q, κ, ϵ, adaptation state, counters) ---- */
int dhmc_state_bytes(dhmc_ctx* ctx, uint64_t* nbytes);
int dhmc_export_state(dhmc_ctx* ctx, void* host_blob, uint64_t nbytes);
int dhmc_import_state(dhmc_ctx* ctx, const void* host_blob, uint64_t nbytes);

/* ---- measurement: HIP-event time of the kernels of the last dhmc_run on its stream --- */
double dhmc_last_run_kernel_ms(const dhmc_ctx* ctx);
/* total leapfrog steps (= gradient evaluations, Σ TreeStatisticsNUTS.steps) of the last dhmc_run */
uint64_t dhmc_last_run_leapfrogs(const dhmc_ctx* ctx);
/* dense contexts: number of leapfrog rounds (one fp64-MFMA GEMM pair each) the last dhmc_run took */
uint64_t dhmc_last_run_rounds(const dhmc_ctx* ctx);
/* bytes of device workspace the context holds (sizing for 288 GB HBM) */
uint64_t dhmc_workspace_bytes(const dhmc_ctx* ctx);

/* ---- self-test of the numerical contract: the scalar functions of dhmc_detmath.h evaluated on the device ----
 * kind: 0 exp, 1 log, 2 log1p (x >= 0), 3 sin(2πx), 4 cos(2πx), 5 randexp (x = 64 random bits as a double's bit pattern),
 * 6 / 7 the two normals of randn2 (x, y = 2 × 64 random bits), 8 logaddexp(x, y), 9 x^y (x > 0), 10 / 11 the logistic family's
 * link σ(x) and log(1 + e^x) (det_logistic_sigma, det_log1pexp);  policy: how the kernels
 * place operands — 0 the header as compiled for the device, 1 per-lane arguments (kinds 10 / 11: the logistic round engine's own
 * batched evaluation, both quotients from one reciprocal), 2 wave-uniform arguments (scalar loads,
 * scalar-register coefficients; not for kinds 10 / 11).  Host arrays of n values (y may be null where unused).  Every policy must return the bits
 * the CPU side of the same header returns: a caller that builds its own host code against dhmc_detmath.h can check its
 * compiler / flags against the device with this. */
int dhmc_detmath_selftest(int32_t device, int32_t kind, int32_t policy, int64_t n, const double* x, const double* y, double* out);

#ifdef __cplusplus
}
#endif
#endif /* DHMC_H */
