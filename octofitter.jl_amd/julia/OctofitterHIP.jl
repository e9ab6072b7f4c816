# OctofitterHIP.jl — the reference-side binding of include/octofitter_hip.h: pure `ccall`, no CUDA.jl/AMDGPU.jl.
#
# NOT EXECUTED IN THE BUILD CONTAINER (Julia is not installed there). Every behaviour it relies on is
# exercised through the same shared library from Python (tests/, ctypes) — see INTEGRATION.md.
#
# What it adds next to an existing `Octofitter.LogDensityModel` (src/logdensitymodel.jl) — nothing in the
# reference is modified:
#
#   g = OctofitterHIP.GPUBatchedLikelihood(model)        # walks model.system, uploads the tables once
#   ll      = ln_like_batch(g, Θ)                        # Θ :: Vector of arr2nt NamedTuples (or a D×W matrix of θ_t)
#   ll, ∇   = ln_post_and_grad_batch(g, Θ_t)             # log-posterior and gradient w.r.t. θ_t, D×W
#   g(system, θ_nt)                                      # drop-in for the closure make_ln_like returns (W = 1)
#
# Eligibility follows SURVEY.md §8(b): tables the kernels implement go to the device; epoch-free prior-like
# terms (UnitLengthPrior, UserLikelihood, PlanetOrderPrior, ...) are evaluated on the host and added; any other
# epoch-bearing observation (full HGCA line fit, GP RV, images, ...) makes the model ineligible and the constructor throws, so
# callers keep using model.ℓπcallback.
module OctofitterHIP

using Octofitter, PlanetOrbits, ForwardDiff
using Octofitter: PlanetRelAstromObs, System, Planet, normalizename, likelihoodname

const LIB = get(ENV, "OCTOFITTER_HIP_LIB", "liboctofitter_hip.so")

const OCTO_OK = Int32(0)
const ASTROM_RADEC, ASTROM_SEPPA, RV_ABS, RV_ABS_MARG, RV_REL = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const ONEIL_RADEC, ONEIL_SEPPA, HGCA = Int32(5), Int32(6), Int32(7)
const ORBIT_VISUAL_KEP, ORBIT_RADVEL, ORBIT_THIELE_INNES = Int32(0), Int32(1), Int32(2)
const N_EL, N_NUIS = 9, 3
const EL_KEYS = (:a, :e, :i, :ω, :Ω, :tp, :M, :plx, :mass)
const EL_KEYS_TI = (:A, :e, :B, :F, :G, :tp, :M, :plx, :mass)     # ThieleInnesOrbit: constants [mas] in the rows of a, i, ω, Ω

struct OctoConsts            # mirrors `octo_consts`
    kepler_year_to_julian_day::Float64; year2day_julian::Float64; au2m::Float64; sec2year_julian::Float64
    pc2au::Float64; rad2as::Float64; mjup2msol::Float64
end
struct OctoObsDesc           # mirrors `octo_obs_desc`
    kind::Int32; planet::Int32; n_epochs::Int64
    epoch::Ptr{Float64}; y1::Ptr{Float64}; y2::Ptr{Float64}; s1::Ptr{Float64}; s2::Ptr{Float64}; cor::Ptr{Float64}
    extra::Ptr{Float64}; n_extra::Int64
end
struct OctoPlanetDesc        # mirrors `octo_planet_desc`
    orbit_kind::Int32; has_mass::Int32
end

check(ctx, st, what) = st == OCTO_OK ? nothing :
    error("$what failed with status $st: " * unsafe_string(ccall((:octo_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))

mutable struct GPUBatchedLikelihood{TModel}
    model::TModel
    ctx::Ptr{Cvoid}
    ds::Ptr{Cvoid}
    n_planets::Int
    obs_entries::Vector{Any}          # (obs, i_planet or 0, θ_obs key) in evaluation order
    host_terms::Vector{Any}           # epoch-free observations evaluated in Julia
    has_mass::Vector{Bool}
    columns::Vector{Vector{Float64}}  # keeps the uploaded host columns alive during octo_dataset_create
end

_f64(x) = collect(Float64, vec(x))

function _table(obs, i_planet)
    t = obs.table
    z = Ptr{Float64}(C_NULL)
    if obs isa PlanetRelAstromObs
        if hasproperty(t, :pa) && hasproperty(t, :sep)      # relative-astrometry.jl:53
            cols = (_f64(t.epoch), _f64(t.pa), _f64(t.sep), _f64(t.σ_pa), _f64(t.σ_sep)); kind = ASTROM_SEPPA
        else
            cols = (_f64(t.epoch), _f64(t.ra), _f64(t.dec), _f64(t.σ_ra), _f64(t.σ_dec)); kind = ASTROM_RADEC
        end
        cor = hasproperty(t, :cor) ? _f64(t.cor) : Float64[]
        return kind, Int32(i_planet - 1), [cols..., cor]
    end
    T = nameof(typeof(obs))
    if T === :ObsPriorAstromONeil2019 && obs.wrapped_like isa PlanetRelAstromObs      # prior-observable.jl:56-76
        kind, planet, cols = _table(obs.wrapped_like, i_planet)
        return (kind == ASTROM_SEPPA ? ONEIL_SEPPA : ONEIL_RADEC), planet, cols
    end
    if T === :HGCAInstantaneousObs                                                     # hgca.jl:58-152
        h = obs.hgca
        meas = Float64[m === :ra ? 0 : 1 for m in t.meas]; inst = Float64[i === :hip ? 0 : 1 for i in t.inst]
        cov(d) = (s = sqrt.(Octofitter.diag(d.Σ)); (s[1], s[2], d.Σ[1, 2] / (s[1] * s[2])))   # includes `factor`
        extra = Float64[h.pmra_hip, h.pmdec_hip, cov(h.dist_hip)..., h.pmra_hg, h.pmdec_hg, cov(h.dist_hg)...,
                        h.pmra_gaia, h.pmdec_gaia, cov(h.dist_gaia)...]
        return HGCA, Int32(-1), [_f64(t.epoch), meas, inst, Float64[], Float64[], Float64[], extra]
    end
    kind = T === :StarAbsoluteRVObs ? RV_ABS : T === :MarginalizedStarAbsoluteRVObs ? RV_ABS_MARG :
           T === :PlanetRelativeRVObs ? RV_REL : nothing
    kind === nothing && return nothing
    # GP / trend branches are Julia closures: not on the device path (rv-absolute.jl:205-315)
    (hasproperty(obs, :gaussian_process) && !isnothing(obs.gaussian_process)) && return nothing
    return kind, Int32(kind == RV_REL ? i_planet - 1 : -1), [_f64(t.epoch), _f64(t.rv), Float64[], _f64(t.σ_rv), Float64[], Float64[]]
end

_has_epochs(obs) = hasproperty(obs, :table) && hasproperty(obs.table, :epoch)     # system.jl:39,48

function GPUBatchedLikelihood(model; device::Integer=0)
    system = model.system
    θ0 = model.arr2nt(model.sample_priors(Octofitter.Random.default_rng()))
    entries = Any[]; host_terms = Any[]; descs = OctoObsDesc[]; columns = Vector{Float64}[]
    add! = function (obs, ip, ctxkind)
        if !_has_epochs(obs)
            push!(host_terms, (obs, ip, ctxkind)); return
        end
        tb = _table(obs, ip)
        tb === nothing && error("observation $(likelihoodname(obs)) is not on the HIP path; keep using model.ℓπcallback")
        kind, planet, cols = tb
        append!(columns, cols)
        p(c) = isempty(c) ? Ptr{Float64}(C_NULL) : pointer(c)
        ex = length(cols) >= 7 ? cols[7] : Float64[]
        push!(descs, OctoObsDesc(kind, planet, length(cols[1]), p(cols[1]), p(cols[2]), p(cols[3]), p(cols[4]), p(cols[5]), p(cols[6]), p(ex), length(ex)))
        push!(entries, (obs, ip, normalizename(likelihoodname(obs))))
    end
    # evaluation order of the generated closure: planet observations planet by planet, then system ones (system.jl:229-235)
    for (ip, pl) in enumerate(system.planets), obs in pl.observations
        add!(obs, ip, :planet)
    end
    for obs in system.observations
        add!(obs, 0, :system)
    end
    planets = OctoPlanetDesc[]
    has_mass = Bool[]
    for (ip, pl) in enumerate(system.planets)
        OT = Octofitter.orbittype(pl)
        ok = OT <: Visual{<:KepOrbit} ? ORBIT_VISUAL_KEP : OT <: RadialVelocityOrbit ? ORBIT_RADVEL : OT <: ThieleInnesOrbit ? ORBIT_THIELE_INNES :
             error("orbit type $OT is not on the HIP path")
        hm = hasproperty(θ0.planets[ip], :mass)                 # relative-astrometry.jl:122
        push!(planets, OctoPlanetDesc(ok, hm)); push!(has_mass, hm)
    end
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:octo_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32), ctx, device)
    st == OCTO_OK || error("octo_ctx_create failed with status $st (no usable MI355X?)")
    consts = OctoConsts(PlanetOrbits.kepler_year_to_julian_day_conversion_factor, PlanetOrbits.year2day_julian,
                        PlanetOrbits.au2m, PlanetOrbits.sec2year_julian, PlanetOrbits.pc2au, PlanetOrbits.rad2as, Octofitter.mjup2msol)
    check(ctx[], ccall((:octo_consts_set, LIB), Int32, (Ptr{Cvoid}, Ref{OctoConsts}), ctx[], consts), "octo_consts_set")
    ds = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve columns begin
        check(ctx[], ccall((:octo_dataset_create, LIB), Int32,
                           (Ptr{Cvoid}, Ptr{OctoObsDesc}, Int32, Ptr{OctoPlanetDesc}, Int32, Ref{Ptr{Cvoid}}),
                           ctx[], descs, length(descs), planets, length(planets), ds), "octo_dataset_create")
    end
    g = GPUBatchedLikelihood(model, ctx[], ds[], length(planets), entries, host_terms, has_mass, columns)
    finalizer(g) do x
        ccall((:octo_dataset_destroy, LIB), Int32, (Ptr{Cvoid},), x.ds)
        ccall((:octo_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), x.ctx)
    end
    return g
end

# ---- θ (nested NamedTuple from arr2nt) -> the kernel's inputs --------------------------------------------------
"Resolved orbital elements and nuisances of ONE parameter set, in C-ABI order; generic in the number type."
function kernel_inputs(g::GPUBatchedLikelihood, θ)
    T = Octofitter._system_number_type(θ)
    x = Vector{T}(undef, g.n_planets * N_EL + length(g.obs_entries) * N_NUIS)
    for ip in 1:g.n_planets
        θp = merge(θ, θ.planets[ip])                                    # system.jl:117
        keys = Octofitter.orbittype(g.model.system.planets[ip]) <: ThieleInnesOrbit ? EL_KEYS_TI : EL_KEYS
        for (k, key) in enumerate(keys)
            x[(ip-1)*N_EL+k] = hasproperty(θp, key) ? getproperty(θp, key) : zero(T)
        end
    end
    o0 = g.n_planets * N_EL
    for (io, (obs, ip, key)) in enumerate(g.obs_entries)
        src = ip > 0 ? θ.planets[ip].observations : θ.observations
        θobs = hasproperty(src, key) ? getproperty(src, key) : (;)
        if nameof(typeof(obs)) === :HGCAInstantaneousObs                  # θ_system.pmra / .pmdec, hgca.jl:266-267
            x[o0+(io-1)*N_NUIS+1] = θ.pmra; x[o0+(io-1)*N_NUIS+2] = θ.pmdec; x[o0+(io-1)*N_NUIS+3] = zero(T)
        elseif obs isa PlanetRelAstromObs || nameof(typeof(obs)) === :ObsPriorAstromONeil2019   # relative-astrometry.jl:170-172
            x[o0+(io-1)*N_NUIS+1] = hasproperty(θobs, :jitter) ? θobs.jitter : zero(T)
            x[o0+(io-1)*N_NUIS+2] = hasproperty(θobs, :platescale) ? θobs.platescale : one(T)
            x[o0+(io-1)*N_NUIS+3] = hasproperty(θobs, :northangle) ? θobs.northangle : zero(T)
        else
            x[o0+(io-1)*N_NUIS+1] = hasproperty(θobs, :offset) ? θobs.offset : zero(T)
            x[o0+(io-1)*N_NUIS+2] = hasproperty(θobs, :jitter) ? θobs.jitter : zero(T)
            x[o0+(io-1)*N_NUIS+3] = zero(T)
        end
    end
    return x
end

"Host-side (epoch-free) likelihood terms, summed exactly as the reference does."
function host_ll(g::GPUBatchedLikelihood, θ)
    ll = zero(Octofitter._system_number_type(θ))
    for (obs, ip, kind) in g.host_terms
        key = normalizename(likelihoodname(obs))
        if kind === :planet
            src = θ.planets[ip].observations
            θobs = hasproperty(src, key) ? getproperty(src, key) : (;)
            orbits = ntuple(i -> Octofitter.orbittype(g.model.system.planets[i])(; merge(θ, θ.planets[i])...), g.n_planets)
            ll += Octofitter.ln_like(obs, Octofitter.PlanetObservationContext(θ, θ.planets[ip], θobs, orbits, ntuple(_ -> (), g.n_planets), ip, -1))
        else
            θobs = hasproperty(θ.observations, key) ? getproperty(θ.observations, key) : (;)
            orbits = ntuple(i -> Octofitter.orbittype(g.model.system.planets[i])(; merge(θ, θ.planets[i])...), g.n_planets)
            ll += Octofitter.ln_like(obs, Octofitter.SystemObservationContext(θ, θobs, orbits, ntuple(_ -> (), g.n_planets), -1))
        end
    end
    return ll
end

# ---- raw batched call ---------------------------------------------------------------------------------------------
"X :: (n_planets*9 + n_obs*3) × W, column per walker. Returns ll[W] and, if grad, ∂ll/∂X of the same shape."
function eval_inputs(g::GPUBatchedLikelihood, X::Matrix{Float64}; grad::Bool=false)
    n_el = g.n_planets * N_EL
    W = size(X, 2)
    Xt = permutedims(X)                                  # [W, inputs]: walker index fastest, as the C ABI wants
    ll = Vector{Float64}(undef, W)
    G = grad ? similar(Xt) : Xt
    n_nu = size(X, 1) - n_el
    pel, pnu = pointer(Xt), n_nu > 0 ? pointer(Xt, n_el * W + 1) : Ptr{Float64}(C_NULL)
    gel = grad ? pointer(G) : Ptr{Float64}(C_NULL)
    gnu = grad && n_nu > 0 ? pointer(G, n_el * W + 1) : Ptr{Float64}(C_NULL)
    GC.@preserve Xt ll G begin
        check(g.ctx, ccall((:octo_eval, LIB), Int32,
                           (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                           g.ctx, g.ds, pel, pnu, W, W, ll, gel, gnu), "octo_eval")
    end
    return grad ? (ll, permutedims(G)) : ll
end

# ---- the callable surface -----------------------------------------------------------------------------------------
"ln_like for a batch of structured parameter sets (what `make_ln_like(system, θ)(system, θ)` returns, W at a time)."
function ln_like_batch(g::GPUBatchedLikelihood, Θ::AbstractVector)
    X = reduce(hcat, (Float64.(kernel_inputs(g, θ)) for θ in Θ))
    ll = eval_inputs(g, X)
    isempty(g.host_terms) || (ll .+= (host_ll(g, θ) for θ in Θ))
    return ll
end

"Drop-in for the closure returned by make_ln_like (src/likelihoods/system.jl:206): one θ at a time."
(g::GPUBatchedLikelihood)(system, θ) = ln_like_batch(g, [θ])[1]

"""
log-posterior and its gradient w.r.t. the unconstrained θ_t for a D×W batch — the batched sibling of
`model.∇ℓπcallback` (src/logdensitymodel.jl:169-177). The kernel returns ḡ = ∂ll/∂(elements, nuisances); the
cheap per-walker map θ_t -> (elements, nuisances) and the prior are differentiated on the host with ForwardDiff
(no epoch loop), and ∇θ_t = Jᵀ ḡ + ∇θ_t(ln prior + host terms).
"""
function ln_post_and_grad_batch(g::GPUBatchedLikelihood, Θt::AbstractMatrix{<:Real}, ln_prior_transformed)
    m = g.model
    D, W = size(Θt)
    tonat(θt) = m.arr2nt(m.invlink(θt))
    X = Matrix{Float64}(undef, g.n_planets * N_EL + length(g.obs_entries) * N_NUIS, W)
    Js = Vector{Matrix{Float64}}(undef, W)
    lp = Vector{Float64}(undef, W); ∇lp = Matrix{Float64}(undef, D, W)
    for w in 1:W
        θt = collect(Θt[:, w])
        res = ForwardDiff.jacobian(t -> kernel_inputs(g, tonat(t)), θt)
        Js[w] = res
        X[:, w] = Float64.(kernel_inputs(g, tonat(θt)))
        host(t) = ln_prior_transformed(m.invlink(t), true) + host_ll(g, tonat(t))
        lp[w] = host(θt); ∇lp[:, w] = ForwardDiff.gradient(host, θt)
    end
    ll, G = eval_inputs(g, X; grad=true)
    ∇ = similar(∇lp)
    for w in 1:W
        ∇[:, w] = Js[w]' * G[:, w] .+ ∇lp[:, w]
        isfinite(lp[w]) || (ll[w] = 0.0; ∇[:, w] .= 0.0)           # logdensitymodel.jl:130-133
    end
    return lp .+ ll, ∇
end

export GPUBatchedLikelihood, ln_like_batch, ln_post_and_grad_batch
end # module
