# NOT EXECUTED in this repository's environment (no Julia runtime in the build image or on the GPU box).
#
# External reproduction of the CPU reference number for BASELINE.json configs[0] (and, with CHAINS/DIM set,
# a reduced configs[1]) with the real DynamicHMC.jl: leapfrog-steps/s = Σ tree_statistics.steps / wall-clock of
# the sampling phase, plus the warmup-inclusive rate.  Compare with `python bench.py`'s `cpu_baseline`
# (the C++ restatement under oracle/) and `value` (MI355X).
#
#   julia --project -t 4 bench/reference_cpu.jl            # config 1: D = 100, 4 chains, 900 warmup + 1000 draws
#   DIM=1000 CHAINS=64 julia --project -t auto bench/reference_cpu.jl
using DynamicHMC, LogDensityProblems, Random, Statistics

struct StdNormal
    D::Int
end
LogDensityProblems.capabilities(::Type{StdNormal}) = LogDensityProblems.LogDensityOrder{1}()
LogDensityProblems.dimension(ℓ::StdNormal) = ℓ.D
LogDensityProblems.logdensity(ℓ::StdNormal, q) = -sum(abs2, q) / 2
LogDensityProblems.logdensity_and_gradient(ℓ::StdNormal, q) = (-sum(abs2, q) / 2, -q)

const D = parse(Int, get(ENV, "DIM", "100"))
const C = parse(Int, get(ENV, "CHAINS", "4"))
const N = parse(Int, get(ENV, "DRAWS", "1000"))
ℓ = StdNormal(D)

# one chain = one mcmc_keep_warmup call (docs/src/worked_example.md: chains are the caller's loop)
function chain(seed)
    rng = Random.Xoshiro(seed)
    t0 = time()
    r = mcmc_keep_warmup(rng, ℓ, 0; reporter = NoProgressReport())          # default_warmup_stages(): 900 transitions
    t1 = time()
    warm_steps = sum(s -> s.results === nothing ? 0 : sum(t -> t.steps, s.results.tree_statistics), r.warmup)
    inf = DynamicHMC.mcmc(r.sampling_logdensity, N, r.final_warmup_state)
    t2 = time()
    (warm_steps = warm_steps, warm_s = t1 - t0, steps = sum(t -> t.steps, inf.tree_statistics), s = t2 - t1,
     ϵ = r.final_warmup_state.ϵ, depth = mean(t -> t.depth, inf.tree_statistics))
end

chain(0)                                                                    # compile
t = time()
rs = Vector{Any}(undef, C)
Threads.@threads for c in 1:C
    rs[c] = chain(c - 1)
end
wall = time() - t
println("threads = ", Threads.nthreads(), ", D = $D, chains = $C, draws = $N")
println("sampling: ", sum(r -> r.steps, rs) / maximum(r -> r.s, rs), " leapfrog-steps/s (Σ steps / slowest chain's sampling time)")
println("whole job (warmup + sampling): ", sum(r -> r.steps + r.warm_steps, rs) / wall, " leapfrog-steps/s")
println("mean ϵ = ", mean(r -> r.ϵ, rs), ", mean depth = ", mean(r -> r.depth, rs))
