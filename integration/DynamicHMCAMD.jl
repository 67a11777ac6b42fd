# NOT EXECUTED in this repository (no Julia runtime in the build image or on the GPU box): the ccall shim of
# INTEGRATION.md §2 as a file.  Every call it makes is exercised through the identical ctypes binding
# (dynamichmc.jl_amd/_abi.py, context.py) by the GPU tests, and through tests/cabi/cabi_client.c from plain C.
#
# Two layers:
#   1. primitives — one Julia function per C-ABI entry point (Context, init!, run!, update_metric!, ...);
#   2. the drop-in surface — methods of DynamicHMC's OWN generic functions `warmup`, `mcmc`, `mcmc_steps`,
#      `mcmc_next_step` for a `SamplingLogDensityAMD`, plus `mcmc_keep_warmup` / `mcmc_with_warmup` methods for a
#      `DeviceLogDensity`, so that the reference's stage fold `_warmup` (src/mcmc.jl:450-457), its stage tuples
#      (`default_warmup_stages`, `fixed_stepsize_warmup_stages`) and its result NamedTuples are used unchanged, with a
#      leading chain dimension on every array.
module DynamicHMCAMD
import DynamicHMC
using DynamicHMC: NUTS, DualAveraging, FixedStepsize, InitialStepsizeSearch, TuningNUTS, DynamicHMCError,
                  TreeStatisticsNUTS, InvalidTree, Directions, default_warmup_stages, default_reporter, report, make_mcmc_reporter,
                  NoProgressReport, LogProgressReport,
                  REPORT_SIGDIGITS
using LinearAlgebra: Diagonal, Symmetric
using Statistics: median
using AMDGPU: ROCArray                      # device-resident results (run!(...; on_device = true)) and external models
const libdhmc = "libdhmc_amd.so"            # dynamichmc.jl_amd/lib/

# Page-locked host arrays (dhmc_host_alloc): dhmc_run fills them at PCIe speed, chunk k's copy under chunk k+1's kernel;
# an ordinary Array works too, at the runtime's staged rate (DESIGN.md §6: 7.6e7 against 1.0e8 leapfrog-steps/s).
function pinned_array(::Type{T}, dims...) where {T}
    p = Ref{Ptr{Cvoid}}()
    rc = ccall((:dhmc_host_alloc, libdhmc), Cint, (Ref{Ptr{Cvoid}}, UInt64), p, prod(dims) * sizeof(T))
    rc == 0 || throw(OutOfMemoryError())
    a = unsafe_wrap(Array, Ptr{T}(p[]), dims; own = false)
    finalizer(_ -> ccall((:dhmc_host_free, libdhmc), Cint, (Ptr{Cvoid},), p[]), a)
end

struct Config                              # dhmc_config, include/dhmc.h
    device::Int32; dim::Int32; chains::Int32; chain_offset::Int32
    metric::Int32; target::Int32
    target_params::Ptr{Cvoid}; target_params_bytes::UInt64
    max_depth::Int32; dense_per_chain::Int32; min_delta::Float64; seed::UInt64
end
struct DualAveragingABI                    # dhmc_dual_averaging
    delta::Float64; gamma::Float64; kappa::Float64
    t0::Int32; init::Int32; finalize::Int32; reserved::Int32
end
struct StepsizeSearchABI                   # dhmc_stepsize_search
    initial_eps::Float64; log_threshold::Float64; maxiter_crossing::Int32; reserved::Int32
end
struct Outputs                             # dhmc_outputs: any pointer may be C_NULL
    on_device::Int32; reserved::Int32
    draws::Ptr{Float64}; logdensities::Ptr{Float64}; eps::Ptr{Float64}; pi::Ptr{Float64}
    acceptance_rate::Ptr{Float64}; steps::Ptr{Int64}; term_left::Ptr{Int64}; term_right::Ptr{Int64}
    depth::Ptr{Int32}; directions::Ptr{UInt32}
end

const STATUS_MESSAGES = (
    (0x1, "Position vector has non-finite elements."),                 # hamiltonian.jl:203
    (0x2, "Invalid log posterior."),                                   # hamiltonian.jl:212-216
    (0x4, "Initial stepsize search reached maximum number of iterations without crossing."),  # stepsize.jl:58
    (0x8, "Starting point has non-finite density."),                   # stepsize.jl:78
    (0x40000000, "Internal error: a bounded wait inside a kernel ran out (DHMC_ST_KERNEL_PROTOCOL); not the model's fault."))

function check(ctx, rc, what; status = nothing)                      # status: the probes' own per-chain words
    rc == 0 && return
    rc == 1 && throw(ArgumentError(what))                              # the @argcheck sites
    if rc == 4                                                         # DHMC_ERR_CHAIN_FAILURE
        st = status === nothing ? Vector{UInt32}(undef, ctx.chains) : status
        status === nothing && ccall((:dhmc_get_status, libdhmc), Cint, (Ptr{Cvoid}, Ptr{UInt32}), ctx.h, st)
        bad = findall(!iszero, st)
        msg = first(m for (b, m) in STATUS_MESSAGES if any(s -> s & b != 0, st[bad]))
        throw(DynamicHMCError(msg, (chains = bad, status = st[bad])))  # utilities.jl:17-27
    end
    error("$what: ", unsafe_string(ccall((:dhmc_last_error, libdhmc), Cstring, (Ptr{Cvoid},), ctx.h)))
end

mutable struct Context
    h::Ptr{Cvoid}; chains::Int; dim::Int; metric::Int; dense_per_chain::Bool
    callback::Any                             # keeps the @cfunction of an external model alive
end

function Context(; dim, chains, target = 0, params = Float64[], seed = 0x23ef614d, algorithm = NUTS(),
                 chain_offset = 0, device = 0, metric = 0, dense_per_chain = false)
    # metric: 0 per-chain diagonal; 1 dense — one M⁻¹ shared and adapted from the pooled draws (GEMM engines), or with
    # dense_per_chain one M⁻¹ per chain adapted from that chain's own draws, as the reference does (mcmc.jl:281-285)
    cfg = Config(device, dim, chains, chain_offset, metric, target, pointer(params), sizeof(params),
                 algorithm.max_depth, dense_per_chain, algorithm.min_Δ, seed)
    h = Ref{Ptr{Cvoid}}()
    rc = GC.@preserve params ccall((:dhmc_create, libdhmc), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || throw(ArgumentError("dhmc_create: code $rc"))
    finalizer(c -> ccall((:dhmc_destroy, libdhmc), Cint, (Ptr{Cvoid},), c.h), Context(h[], chains, dim, metric, dense_per_chain, nothing))
end

# initialize_warmup_state (mcmc.jl:129-132); q0 is D×C (each column a chain) or nothing
init!(ctx, q0 = nothing) = check(ctx, ccall((:dhmc_init, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint),
        ctx.h, q0 === nothing ? C_NULL : pointer(q0), 0), "dhmc_init")

# warmup(::InitialStepsizeSearch) (mcmc.jl:134-148)
find_initial_stepsize!(ctx, s::InitialStepsizeSearch) = check(ctx,
    ccall((:dhmc_find_initial_stepsize, libdhmc), Cint, (Ptr{Cvoid}, Ref{StepsizeSearchABI}), ctx.h,
          StepsizeSearchABI(s.initial_ϵ, s.log_threshold, s.maxiter_crossing, 0)), "dhmc_find_initial_stepsize")

# the per-draw loops (mcmc.jl:271-280 with `da`, :374-379 without)
# on_device: the draws stay in HBM (a ROCArray, for dhmc_update_metric_* / dhmc_ess_* on device pointers); the per-draw
# scalars always come to the host
function run!(ctx, N; da::Union{Nothing,DualAveraging} = nothing, on_device::Bool = false, da_init::Bool = true, da_finalize::Bool = true)
    D, C = ctx.dim, ctx.chains
    mk(T, dims...) = on_device ? ROCArray{T}(undef, dims...) : pinned_array(T, dims...)
    pm = mk(Float64, D, N, C)                          # posterior_matrix[:, i] per chain (mcmc.jl:275)
    ℓs = mk(Float64, N, C); ϵs = mk(Float64, N, C); π = mk(Float64, N, C); a = mk(Float64, N, C)
    steps = mk(Int64, N, C); tl = mk(Int64, N, C); tr = mk(Int64, N, C)
    depth = mk(Int32, N, C); dirs = mk(UInt32, N, C)
    out = Outputs(on_device, 0, pointer(pm), pointer(ℓs), pointer(ϵs), pointer(π), pointer(a), pointer(steps),
                  pointer(tl), pointer(tr), pointer(depth), pointer(dirs))
    daref = da === nothing ? C_NULL : Ref(DualAveragingABI(da.δ, da.γ, da.κ, da.t₀, da_init, da_finalize, 0))   # a stage run as several calls: init first, finalize last
    rc = GC.@preserve pm ℓs ϵs π a steps tl tr depth dirs ccall((:dhmc_run, libdhmc), Cint,
            (Ptr{Cvoid}, Int64, Ptr{DualAveragingABI}, Ref{Outputs}), ctx.h, N, daref, out)
    check(ctx, rc, "dhmc_run")
    if on_device                                       # the scalars to the host; pm stays where it is
        ℓs, ϵs, π, a, steps, tl, tr, depth, dirs = Array.((ℓs, ϵs, π, a, steps, tl, tr, depth, dirs))
    end
    stats = [TreeStatisticsNUTS(π[i, c], depth[i, c], InvalidTree(tl[i, c], tr[i, c]), a[i, c], steps[i, c],
                                Directions(dirs[i, c])) for i in 1:N, c in 1:C]       # NUTS.jl:208-221
    (posterior_matrix = pm, tree_statistics = stats, ϵs = ϵs, logdensities = ℓs)
end

# end of a TuningNUTS{Diagonal} stage (mcmc.jl:281-284), from a posterior matrix the caller holds …
update_metric!(ctx, pm, λ) = check(ctx, ccall((:dhmc_update_metric_diag, libdhmc), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Cint), ctx.h, pointer(pm), size(pm, 2), λ, 0), "dhmc_update_metric_diag")
# … or from a metric window: running moments that the kernels update with every draw between `metric_window_begin!` and
# `update_metric_window!` (include/dhmc.h), so that the stage's variance needs no posterior matrix on the device
metric_window_begin!(ctx) = check(ctx, ccall((:dhmc_metric_window_begin, libdhmc), Cint, (Ptr{Cvoid},), ctx.h), "dhmc_metric_window_begin")
update_metric_window!(ctx, λ) = check(ctx, ccall((:dhmc_update_metric_diag_window, libdhmc), Cint, (Ptr{Cvoid}, Float64), ctx.h, λ),
                                      "dhmc_update_metric_diag_window")
metric_window_end!(ctx) = check(ctx, ccall((:dhmc_metric_window_end, libdhmc), Cint, (Ptr{Cvoid},), ctx.h), "dhmc_metric_window_end")

# dense κ (contexts created with metric = 1): GaussianKineticEnergy(M⁻¹) (hamiltonian.jl:73) and the end of a
# TuningNUTS{Symmetric} stage (mcmc.jl:210,218-222; pooled over the context's chains)
set_metric_dense!(ctx, Minv::Matrix{Float64}) = check(ctx, ccall((:dhmc_set_metric_dense, libdhmc), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Cint), ctx.h, pointer(Minv), 0), "dhmc_set_metric_dense")
# 2: the reference's leapfrog with a dense metric (two M⁻¹ products, hamiltonian.jl:278 and :103); 1 (default of a shared
# dense metric): the same map with one product per step (include/dhmc.h)
set_dense_products!(ctx, n::Integer) = check(ctx, ccall((:dhmc_set_dense_products, libdhmc), Cint, (Ptr{Cvoid}, Int32), ctx.h, n),
                                             "dhmc_set_dense_products")
update_metric_dense!(ctx, pm, λ) = check(ctx, ccall((:dhmc_update_metric_dense, libdhmc), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Cint), ctx.h, pointer(pm), size(pm, 2), λ, 0), "dhmc_update_metric_dense")

# Diagnostics.leapfrog_trajectory (diagnostics.jl:214-227) from every chain's current position; p is D×C or nothing
function leapfrog_trajectory(ctx, ϵ, positions::UnitRange{<:Integer}; p = nothing, momentum_index = 0)
    A, B = first(positions), last(positions); npos = B - A + 1
    Δ = Matrix{Float64}(undef, npos, ctx.chains); ℓq = similar(Δ)
    q = Array{Float64,3}(undef, ctx.dim, npos, ctx.chains); ps = similar(q)
    range = Matrix{Int32}(undef, 2, ctx.chains); status = Vector{UInt32}(undef, ctx.chains)
    check(ctx, ccall((:dhmc_leapfrog_trajectory, libdhmc), Cint,
        (Ptr{Cvoid}, Float64, Int32, Int32, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{UInt32}),
        ctx.h, ϵ, A, B, momentum_index, p === nothing ? C_NULL : pointer(p), Δ, ℓq, q, ps, range, status), "dhmc_leapfrog_trajectory"; status)
    (; Δ, ℓq, q, p = ps, range)      # columns outside range[1,c]:range[2,c] are NaN (trajectory ended at a non-finite ℓq)
end

# Diagnostics.explore_log_acceptance_ratios (diagnostics.jl:144-152): returns length(log2ϵs) × N × C
function explore_log_acceptance_ratios(ctx, log2ϵs; N = 20, ps = nothing, momentum_index = 0)
    ϵs = collect(2.0 .^ log2ϵs); out = Array{Float64,3}(undef, length(ϵs), N, ctx.chains); status = Vector{UInt32}(undef, ctx.chains)
    check(ctx, ccall((:dhmc_explore_log_acceptance_ratios, libdhmc), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{UInt32}),
        ctx.h, ϵs, length(ϵs), N, momentum_index, ps === nothing ? C_NULL : pointer(ps), out, status), "dhmc_explore_log_acceptance_ratios"; status)
    out
end

# ESS and R̂ of selected coordinates of draws that are still in HBM (pm_dev: device pointer to a D×N×C array)
function ess_rhat(pm_dev::Ptr{Float64}, C, N, D, coords::Vector{Int32}; device = 0)
    ess = Vector{Float64}(undef, length(coords)); rhat = similar(ess)
    rc = ccall((:dhmc_ess_rhat, libdhmc), Cint, (Int32, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Int32}, Int32, Ptr{Float64}, Ptr{Float64}),
               device, C_NULL, pm_dev, C, N, D, coords .- Int32(1), length(coords), ess, rhat)
    rc == 0 || error("dhmc_ess_rhat: code $rc")
    (; ess, rhat)
end

# the caller's own model (contexts created with target = 7): f!(lq, grad, q) on GPU arrays, all chains at once
function set_logdensity!(ctx, f!)
    cb = @cfunction($((user, q, C, ld, D, lq, g, stream) -> (f!(unsafe_wrap(ROCArray, Ptr{Float64}(lq), (C,)),
            unsafe_wrap(ROCArray, Ptr{Float64}(g), (ld, C)), unsafe_wrap(ROCArray, Ptr{Float64}(q), (ld, C))); Cint(0))),
        Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
    ctx.callback = cb                         # keep it alive
    check(ctx, ccall((:dhmc_set_logdensity_callback, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, cb, C_NULL), "dhmc_set_logdensity_callback")
end

# resume (WarmupState + adaptation state + RNG counters of every chain as one POD blob)
function export_state(ctx)
    n = Ref{UInt64}(); ccall((:dhmc_state_bytes, libdhmc), Cint, (Ptr{Cvoid}, Ref{UInt64}), ctx.h, n)
    blob = Vector{UInt8}(undef, n[])
    check(ctx, ccall((:dhmc_export_state, libdhmc), Cint, (Ptr{Cvoid}, Ptr{UInt8}, UInt64), ctx.h, blob, n[]), "dhmc_export_state")
    blob
end

# ---- Q, κ, ϵ of all chains: the WarmupState (mcmc.jl:72-79) with a chain dimension ------------------------------
# Q := evaluate_ℓ(ℓ, q) at positions of the caller's (D×C), keeping κ, ϵ and the random streams (dhmc_set_position)
set_position!(ctx, q::Matrix{Float64}) = check(ctx, ccall((:dhmc_set_position, libdhmc), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Cint), ctx.h, q, 0), "dhmc_set_position")
function position(ctx)
    q = Matrix{Float64}(undef, ctx.dim, ctx.chains); ℓq = Vector{Float64}(undef, ctx.chains); ∇ℓq = similar(q)
    check(ctx, ccall((:dhmc_get_position, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint),
                     ctx.h, q, ℓq, ∇ℓq, 0), "dhmc_get_position")
    (; q, ℓq, ∇ℓq)                            # column c = EvaluatedLogDensity of chain c (hamiltonian.jl:165-186)
end
function stepsize(ctx)
    ϵ = Vector{Float64}(undef, ctx.chains)
    check(ctx, ccall((:dhmc_get_stepsize, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), ctx.h, ϵ, 0), "dhmc_get_stepsize")
    ϵ
end
set_stepsize!(ctx, ϵ::Vector{Float64}) = check(ctx, ccall((:dhmc_set_stepsize, libdhmc), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint), ctx.h, ϵ, length(ϵ) == ctx.chains, 0), "dhmc_set_stepsize")
function kinetic_energy(ctx)                  # κ: D×C diagonals of M⁻¹ (one per chain), the shared Symmetric M⁻¹, or one Symmetric per chain
    if ctx.metric == 0
        m = Matrix{Float64}(undef, ctx.dim, ctx.chains)
        check(ctx, ccall((:dhmc_get_metric_diag, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), ctx.h, m, 0), "dhmc_get_metric_diag")
        return [DynamicHMC.GaussianKineticEnergy(Diagonal(m[:, c])) for c in 1:ctx.chains]
    end
    M = Matrix{Float64}(undef, ctx.dim, ctx.dim); W = similar(M)
    if ctx.dense_per_chain
        return map(0:ctx.chains-1) do c
            check(ctx, ccall((:dhmc_get_metric_dense_chain, libdhmc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}), ctx.h, c, M, W),
                  "dhmc_get_metric_dense_chain")
            DynamicHMC.GaussianKineticEnergy(Symmetric(copy(M)), collect(W'))
        end
    end
    check(ctx, ccall((:dhmc_get_metric_dense, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.h, M, W), "dhmc_get_metric_dense")
    DynamicHMC.GaussianKineticEnergy(Symmetric(M), collect(W'))      # row-major [i][j] read column-major is the transpose
end
"The triple of mcmc.jl:72-79 for all chains; the numbers live in the context, this is their host-side view."
current_warmup_state(ctx) = (Q = position(ctx), κ = kinetic_energy(ctx), ϵ = let e = stepsize(ctx); all(isnan, e) ? nothing : e end)

# ---- the drop-in surface ------------------------------------------------------------------------------------------
"A log density evaluated on the device: a built-in family (`target`, `params`: include/dhmc.h DHMC_TARGET_*) or the caller's
own batched `logdensity_and_gradient!` (target 7).  `dimension`/`capabilities` are those of LogDensityProblems."
struct DeviceLogDensity
    dim::Int; target::Int; params::Vector{Float64}; f!::Any
end
DeviceLogDensity(dim; target = 0, params = Float64[], f! = nothing) = DeviceLogDensity(dim, f! === nothing ? target : 7, params, f!)

"SamplingLogDensity (mcmc.jl:40-53) + the context that holds the chains.  `rng` is the seed of the ABI's counter-based streams."
struct SamplingLogDensityAMD{L,O,S}
    rng::UInt64; ℓ::L; algorithm::O; reporter::S; ctx::Context
end

# The N transitions of a stage with the reference's step reports (mcmc.jl:279,378 -> reporting.jl:120-137).  With NoProgressReport: one
# dhmc_run.  Any other reporter: the stage runs as calls of `step_interval` transitions (ProgressMeterReport: N ÷ 100) — the chains resume
# where they stand, so these are the same transitions with the same bits — and report(mcmc_reporter, step; ϵ) follows each call.
function run_reported!(sl, N; da = nothing, currently_warmup, meta...)
    mcmc_reporter = make_mcmc_reporter(sl.reporter, N; currently_warmup, meta...)
    chunk = sl.reporter isa NoProgressReport ? N : sl.reporter isa LogProgressReport ? sl.reporter.step_interval : max(1, N ÷ 100)
    (chunk ≥ N || N == 0) && return run!(sl.ctx, N; da), mcmc_reporter
    parts = map(0:chunk:N-1) do n0
        L = min(chunk, N - n0)
        r = run!(sl.ctx, L; da, da_init = n0 == 0, da_finalize = n0 + L ≥ N)
        ϵ = round(median(stepsize(sl.ctx)); sigdigits = REPORT_SIGDIGITS)
        da === nothing ? report(mcmc_reporter, n0 + L) : report(mcmc_reporter, n0 + L; ϵ)
        r
    end
    ((posterior_matrix = cat((p.posterior_matrix for p in parts)...; dims = 2), tree_statistics = vcat((p.tree_statistics for p in parts)...),
      ϵs = vcat((p.ϵs for p in parts)...), logdensities = vcat((p.logdensities for p in parts)...)), mcmc_reporter)
end

# warmup(sampling_logdensity, stage, warmup_state) -> (results, warmup_state′): the reference's seam (iv), src/mcmc.jl:99,134,258
DynamicHMC.warmup(sl::SamplingLogDensityAMD, ::Nothing, warmup_state) = (nothing, warmup_state)
function DynamicHMC.warmup(sl::SamplingLogDensityAMD, s::InitialStepsizeSearch, warmup_state)
    warmup_state.ϵ ≡ nothing || throw(ArgumentError("stepsize ϵ manually specified, won't perform initial search"))  # mcmc.jl:137
    find_initial_stepsize!(sl.ctx, s)
    st = current_warmup_state(sl.ctx)
    report(sl.reporter, "found initial stepsize", ϵ = round.(st.ϵ; sigdigits = REPORT_SIGDIGITS))
    nothing, st
end
function DynamicHMC.warmup(sl::SamplingLogDensityAMD, tuning::TuningNUTS{M}, warmup_state) where {M}
    (; N, stepsize_adaptation, λ) = tuning
    ctx = sl.ctx
    M ≡ Diagonal && metric_window_begin!(ctx)
    window_open = M ≡ Diagonal
    results, mcmc_reporter = try
        r = run_reported!(sl, N; da = stepsize_adaptation isa DualAveraging ? stepsize_adaptation : nothing,
                          currently_warmup = true, tuning = M ≡ Nothing ? "stepsize" : "stepsize and $(M) metric")   # mcmc.jl:268-280
        if M ≡ Diagonal
            update_metric_window!(ctx, λ)                                                             # mcmc.jl:209,281-284 (closes the window)
            window_open = false
        elseif M ≡ Symmetric
            update_metric_dense!(ctx, r[1].posterior_matrix, λ)                                       # mcmc.jl:210,218-222 (pooled)
        end
        r
    finally
        window_open && metric_window_end!(ctx)        # whatever threw — the run or the update — no window stays open on the context
    end
    M ≢ Nothing && report(mcmc_reporter, "adaptation finished", adapted_kinetic_energy = kinetic_energy(ctx))
    results, current_warmup_state(ctx)
end

# mcmc(sampling_logdensity, N, warmup_state) (mcmc.jl:366-381)
DynamicHMC.mcmc(sl::SamplingLogDensityAMD, N, warmup_state) =
    let (r, _) = run_reported!(sl, N; currently_warmup = false); (posterior_matrix = r.posterior_matrix, tree_statistics = r.tree_statistics, logdensities = r.logdensities) end

# stepwise (mcmc.jl:335-351): κ and ϵ of `warmup_state` are the context's unless the caller changed them
struct MCMCStepsAMD; sl::SamplingLogDensityAMD; end
function DynamicHMC.mcmc_steps(sl::SamplingLogDensityAMD, warmup_state)
    warmup_state.ϵ ≡ nothing && throw(ArgumentError("ϵ ≢ nothing"))
    ctx = sl.ctx
    warmup_state.ϵ == stepsize(ctx) || set_stepsize!(ctx, warmup_state.ϵ)
    # κ of the warmup state, if it is not the context's own (mcmc.jl:337-339 builds the Hamiltonian from warmup_state.κ).
    # Compared by VALUE with what the context holds (current_warmup_state builds fresh objects every time, so identity says
    # nothing), dispatched on the context's metric layout, uploaded only when it differs.
    κ, own = warmup_state.κ, kinetic_energy(ctx)
    same(a, b) = a.M⁻¹ == b.M⁻¹
    if ctx.metric == 0                                                            # per-chain Diagonal κ: a Vector, or one κ for all chains
        κs = κ isa DynamicHMC.GaussianKineticEnergy ? fill(κ, ctx.chains) : κ
        all(k -> k.M⁻¹ isa Diagonal, κs) || throw(ArgumentError("a Diagonal-metric context takes Diagonal kinetic energies"))
        if !all(same.(κs, own))
            m = reduce(hcat, [Vector(k.M⁻¹.diag) for k in κs])                     # D×C
            check(ctx, ccall((:dhmc_set_metric_diag, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint), ctx.h, m, 1, 0), "dhmc_set_metric_diag")
        end
    elseif ctx.dense_per_chain                                                    # one Symmetric κ per chain, adapted on the device
        all(same.(κ, own)) || throw(ArgumentError("per-chain dense kinetic energies live in the context (dhmc_update_metric_dense); " *
                                                  "a warmup state with other matrices cannot be uploaded chain by chain"))
    else                                                                          # one Symmetric κ shared by the chains
        κ isa DynamicHMC.GaussianKineticEnergy || throw(ArgumentError("a shared-dense-metric context takes one GaussianKineticEnergy"))
        same(κ, own) || set_metric_dense!(ctx, Matrix(κ.M⁻¹))
    end
    warmup_state.Q.q == position(ctx).q || set_position!(ctx, warmup_state.Q.q)
    MCMCStepsAMD(sl)
end
function DynamicHMC.mcmc_next_step(steps::MCMCStepsAMD, Q)
    ctx = steps.sl.ctx
    Q.q == position(ctx).q || set_position!(ctx, Q.q)          # a Q that is not the context's own: evaluate it there
    r = run!(ctx, 1)
    position(ctx), r.tree_statistics[1, :]
end

# mcmc_keep_warmup / mcmc_with_warmup (mcmc.jl:521-532,575-584) with `chains` chains on one GPU
function DynamicHMC.mcmc_keep_warmup(rng::Integer, ℓ::DeviceLogDensity, N::Integer; chains = 1, initialization = (),
                                     warmup_stages = default_warmup_stages(), algorithm = NUTS(),
                                     reporter = default_reporter(), device = 0, chain_offset = 0, per_chain_metric = nothing)
    # per_chain_metric (Symmetric κ only; a Diagonal κ is always per chain): false — one M⁻¹ adapted from the pooled draws of all
    # chains (the leapfrog's products are one GEMM); true — every chain its own, as `chains` separate calls of the reference;
    # nothing (default) — the rule of the Python twin (api.py _per_chain_metric_default): the reference's per-chain metric where that
    # is affordable (a built-in functor family, D ≤ 256, the chains' matrices AND their workspace within 2 GiB), pooled otherwise; a
    # caller who hands over ONE matrix κ asks for the shared metric.  κ.M⁻¹ comes back [D, D, chains] with the per-chain metric.
    dense = any(s -> s isa TuningNUTS{Symmetric}, warmup_stages) || get(initialization, :κ, nothing) isa Matrix
    if per_chain_metric === nothing
        dpad = 64 * cld(ℓ.dim, 64)
        nvec = 18 + 7 * algorithm.max_depth                                       # workspace rows of the dense kernels (nuts_dense_kernel.hpp wd_nvec)
        per_chain_metric = dense && !(get(initialization, :κ, nothing) isa Matrix) && ℓ.f! === nothing && ℓ.target != 4 &&   # 4 = DHMC_TARGET_LOGISTIC
                           ℓ.dim <= 256 && 16 * chains * dpad * dpad + 8 * chains * nvec * dpad <= 2 << 30
        per_chain_metric && reporter isa LogProgressReport && @info "Symmetric metric: one M⁻¹ per chain (κ.M⁻¹ is [D, D, chains]); per_chain_metric = false pools one"
    end
    ctx = Context(; dim = ℓ.dim, chains, target = ℓ.target, params = ℓ.params, seed = UInt64(rng), algorithm,
                  chain_offset, device, metric = dense ? 1 : 0, dense_per_chain = dense && per_chain_metric)
    ℓ.f! === nothing || set_logdensity!(ctx, ℓ.f!)
    sl = SamplingLogDensityAMD(UInt64(rng), ℓ, algorithm, reporter, ctx)
    init!(ctx, get(initialization, :q, nothing))                                         # initialize_warmup_state (mcmc.jl:129-132)
    κ = get(initialization, :κ, nothing)
    κ isa Matrix && set_metric_dense!(ctx, κ)
    κ isa Vector && check(ctx, ccall((:dhmc_set_metric_diag, libdhmc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint), ctx.h, κ, 0, 0), "dhmc_set_metric_diag")
    haskey(initialization, :ϵ) && initialization.ϵ ≢ nothing && set_stepsize!(ctx, fill(Float64(initialization.ϵ), chains))
    initial_warmup_state = current_warmup_state(ctx)
    warmup, warmup_state = DynamicHMC._warmup(sl, warmup_stages, initial_warmup_state)   # the reference's own fold, unchanged
    inference = DynamicHMC.mcmc(sl, N, warmup_state)
    (; initial_warmup_state, warmup, final_warmup_state = warmup_state, inference, sampling_logdensity = sl)
end
function DynamicHMC.mcmc_with_warmup(rng::Integer, ℓ::DeviceLogDensity, N; kwargs...)
    (; final_warmup_state, inference) = DynamicHMC.mcmc_keep_warmup(rng, ℓ, N; kwargs...)
    (; κ, ϵ) = final_warmup_state
    (; inference..., κ, ϵ)
end

# post-hoc diagnostics where the statistics lie (device pointers to [N×C] arrays of a run with on_device outputs)
struct TreeStatisticsSummaryABI
    n::Int64; a_mean::Float64; a_quantiles::NTuple{5,Float64}; max_depth::Int64; divergence::Int64; turning::Int64
    depth_counts::NTuple{33,Int64}
end
function summarize_tree_statistics(π, a, tl, tr, depth, C, N; on_device = true, device = 0)
    out = Ref{TreeStatisticsSummaryABI}(); ebfmi = Vector{Float64}(undef, C)
    rc = ccall((:dhmc_summarize_tree_statistics, libdhmc), Cint,
               (Int32, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Int64, Int64, Cint, Ref{TreeStatisticsSummaryABI}, Ptr{Float64}),
               device, C_NULL, π, a, tl, tr, depth, C, N, on_device, out, ebfmi)
    rc == 0 || error("dhmc_summarize_tree_statistics: code $rc")
    (summary = out[], EBFMI = ebfmi)
end
end
