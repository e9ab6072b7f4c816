# OctofitterHIP.jl — the reference-side binding of include/octofitter_hip.h: pure `ccall`, no CUDA.jl / AMDGPU.jl.
#
# NOT EXECUTED IN THE BUILD CONTAINER (Julia is not installed there). What can be checked without Julia is checked:
# tests/test_abi.py compares every struct mirrored below with the header (field order, types, sizes through a compiled C
# program) and requires a `ccall` for every symbol the header declares. Every behaviour the shim relies on is exercised through
# the same shared library from Python (tests/, ctypes) — see INTEGRATION.md.
#
# Nothing in the reference is modified. Three ways in, from least to most of the callback on the device:
#
#  (1) UNCHANGED SAMPLERS, ANY MODEL      model = Octofitter.LogDensityModel(OctofitterHIP.accelerate(system))
#      `accelerate` returns the same System with every eligible observation wrapped in a `HIPObs` (the pattern of the reference's
#      own ObsPriorAstromONeil2019 wrapper, src/likelihoods/prior-observable.jl:56-76). The reference then builds its callbacks as
#      always (priors, Derived code, arr2nt, ForwardDiff — src/logdensitymodel.jl:25-250) and `octofit`, `octofit_pigeons`,
#      `octofit_rejection`, … run unchanged (src/sampling.jl:412-423 calls LogDensityProblems.logdensity_and_gradient at :252-256).
#      Inside, `ln_like(::HIPObs, ctx)` (the plug-in interface, src/variables.jl:94-102) evaluates ALL wrapped tables in one
#      octo_eval call on the VALUES of the resolved elements and, when they are ForwardDiff.Dual numbers, returns a Dual whose
#      partials are Σ_k ḡ_k · partials(input_k) (SURVEY.md §8b). The wrapped tables expose no `table.epoch`, so the reference
#      solves no Kepler equation on the CPU (src/likelihoods/system.jl:35-54, :131-134).
#
#  (2) THE WHOLE CALLBACK ON THE DEVICE    hm = OctofitterHIP.HIPLogDensityModel(model)
#      for models made of the standard blocks (Uniform/LogUniform/Normal/truncated Normal/Sine priors, UniformCircular,
#      θ_at_epoch_to_tperi): ℓπcallback / ∇ℓπcallback become ONE kernel launch per θ_t (octo_model_logpost), and batches of θ_t
#      (guess_starting_position's 5e5 prior draws, octofit_rejection's draws, Pigeons' replicas) one launch per batch.
#      `hm` implements LogDensityProblems.{dimension, capabilities, logdensity, logdensity_and_gradient} and carries the fields
#      the reference's drivers read (D, ℓπcallback, ∇ℓπcallback, system, link, invlink, arr2nt, sample_priors, starting_points).
#
#  (3) BATCHES OF STRUCTURED θ             g = OctofitterHIP.GPUBatchedLikelihood(model); ln_like_batch(g, Θ)
module OctofitterHIP

using Octofitter, PlanetOrbits, ForwardDiff, LogDensityProblems, Random, Distributions, LinearAlgebra
using Octofitter: PlanetRelAstromObs, System, Planet, AbstractObs, Priors, Derived, normalizename, likelihoodname

include("OctofitterHIP_capi.jl")      # constants, structs and one ccall per symbol of include/octofitter_hip.h (LIB, OctoObsDesc, octo_eval!, …)

# ---------------------------------------------------------------------------------------------------- tables -> octo_obs_desc
_f64(x) = collect(Float64, vec(x))

"θ_obs of `obs` (attached to planet `ip`, or 0 = a system observation) inside a full θ; `(;)` when it declares no variables (system.jl:91-100)."
function _θobs(θ, obs, ip)
    key = Symbol(normalizename(likelihoodname(obs)))
    src = ip > 0 ? θ.planets[ip].observations : θ.observations
    return hasproperty(src, key) ? getproperty(src, key) : (;)
end

"""
    _trend_basis(obs, θobs_draws) -> (nothing, Float64[]) | (coef::Symbol, basis::Vector{Float64}) | nothing

Every RV observation type carries a `trend_function(θ_obs, epoch)` closure (rv-absolute.jl:63-69,143; rv-relative.jl:57-64,131;
rv-absolute-margin.jl:46-52,111) — arbitrary user code; the default returns zero. The kernels carry a trend as ONE θ_obs variable times
a per-row basis column (header: OCTO_NU_RV_TREND). Which case is it? Probed numerically at every table epoch for the given draws of θ_obs:
  * identically zero                                      -> `(nothing, [])`: no trend on the device;
  * `θ_obs[c] * b(epoch)` for one variable `c`             -> `(c, b)` with `b[j] = trend_function(θ_obs with c = 1, epoch_j)` — the
    documented `θ_obs.trend_slope * (epoch - 57000)` (rv-absolute.jl:26) and the reference's own test model (test/runtests.jl:196-201);
  * anything else                                          -> `nothing`: the observation is NOT eligible and stays on the reference's CPU path.
"""
function _trend_basis(obs, θobs_draws)
    hasproperty(obs, :trend_function) || return (nothing, Float64[])
    tf = obs.trend_function
    epochs = _f64(obs.table.epoch)
    f(θ) = Float64[tf(θ, t) for t in epochs]
    vals = [f(θ) for θ in θobs_draws]
    all(v -> all(iszero, v), vals) && return (nothing, Float64[])
    θ1 = first(θobs_draws)
    for c in keys(θ1)
        getproperty(θ1, c) isa Real || continue
        b = f(merge(θ1, NamedTuple{(c,)}((1.0,))))
        z = f(merge(θ1, NamedTuple{(c,)}((0.0,))))
        (all(iszero, z) && all(isfinite, b)) || continue
        scale = max(maximum(abs, b; init=0.0), floatmin(Float64))
        linear = all(zip(vals, θobs_draws)) do (v, θ)
            x = Float64(getproperty(θ, c))
            all(abs.(v .- x .* b) .<= 1e-12 * scale * max(1.0, abs(x)))
        end
        linear && return (c, b)
    end
    return nothing
end

"""
(kind, 0-based planet or -1, [epoch, y1, y2, s1, s2, cor, extra], trend coefficient or nothing) of an observation the kernels
implement, or nothing. `θobs_draws`: a few draws of the observation's θ_obs (RV kinds: to classify the trend closure).
"""
function _table(obs, i_planet, θobs_draws=[(;)])
    t = obs.table
    if obs isa PlanetRelAstromObs
        if hasproperty(t, :pa) && hasproperty(t, :sep)      # relative-astrometry.jl:53
            cols = (_f64(t.epoch), _f64(t.pa), _f64(t.sep), _f64(t.σ_pa), _f64(t.σ_sep)); kind = ASTROM_SEPPA
        else
            cols = (_f64(t.epoch), _f64(t.ra), _f64(t.dec), _f64(t.σ_ra), _f64(t.σ_dec)); kind = ASTROM_RADEC
        end
        cor = hasproperty(t, :cor) ? _f64(t.cor) : Float64[]
        return kind, Int32(i_planet - 1), [cols..., cor, Float64[]], nothing
    end
    T = nameof(typeof(obs))
    if T === :ObsPriorAstromONeil2019 && obs.wrapped_like isa PlanetRelAstromObs      # prior-observable.jl:56-76
        kind, planet, cols, _ = _table(obs.wrapped_like, i_planet)
        return (kind == ASTROM_SEPPA ? ONEIL_SEPPA : ONEIL_RADEC), planet, cols, nothing
    end
    if T === :HGCAInstantaneousObs                                                     # hgca.jl:58-152
        h = obs.hgca
        meas = Float64[m === :ra ? 0 : 1 for m in t.meas]; inst = Float64[i === :hip ? 0 : 1 for i in t.inst]
        cov(d) = (s = sqrt.(diag(d.Σ)); (s[1], s[2], d.Σ[1, 2] / (s[1] * s[2])))   # includes `factor`
        extra = Float64[h.pmra_hip, h.pmdec_hip, cov(h.dist_hip)..., h.pmra_hg, h.pmdec_hg, cov(h.dist_hg)...,
                        h.pmra_gaia, h.pmdec_gaia, cov(h.dist_gaia)...]
        return HGCA, Int32(-1), [_f64(t.epoch), meas, inst, Float64[], Float64[], Float64[], extra], nothing
    end
    kind = T === :StarAbsoluteRVObs ? RV_ABS : T === :MarginalizedStarAbsoluteRVObs ? RV_ABS_MARG :
           T === :PlanetRelativeRVObs ? RV_REL : nothing
    kind === nothing && return nothing
    # the GP branch is a Julia closure over AbstractGPs: not on the device path (rv-absolute.jl:205-315)
    (hasproperty(obs, :gaussian_process) && !isnothing(obs.gaussian_process)) && return nothing
    # the trend closure: zero, or linear in one θ_obs variable (-> basis column), or the observation is not eligible
    tb = _trend_basis(obs, θobs_draws)
    tb === nothing && return nothing
    coef, basis = tb
    return kind, Int32(kind == RV_REL ? i_planet - 1 : -1), [_f64(t.epoch), _f64(t.rv), Float64[], _f64(t.σ_rv), Float64[], Float64[], basis], coef
end

_has_epochs(obs) = hasproperty(obs, :table) && hasproperty(obs.table, :epoch)     # system.jl:39,48

function _orbit_kind(pl)
    OT = Octofitter.orbittype(pl)
    OT <: Visual{<:KepOrbit} ? ORBIT_VISUAL_KEP : OT <: RadialVelocityOrbit ? ORBIT_RADVEL : OT <: ThieleInnesOrbit ? ORBIT_THIELE_INNES :
    OT <: KepOrbit ? ORBIT_KEP : error("orbit type $OT is not on the HIP path")
end

_consts() = OctoConsts(PlanetOrbits.kepler_year_to_julian_day_conversion_factor, PlanetOrbits.year2day_julian,
                       PlanetOrbits.au2m, PlanetOrbits.sec2year_julian, PlanetOrbits.pc2au, PlanetOrbits.rad2as, Octofitter.mjup2msol)

"A few full parameter sets drawn from the priors: what `_table` classifies an RV trend closure against."
_θ_draws(system, n=3) = (arr2nt = Octofitter.make_arr2nt(system); sampler = Octofitter.make_prior_sampler(system);
                         rng = Random.Xoshiro(20260929); [arr2nt(sampler(rng)) for _ in 1:n])

"""
Context + dataset for a list of (obs, i_planet or 0) in evaluation order. Returns (ctx, ds, entries, columns); an entry is
(obs, i_planet or 0, θ_obs key, trend coefficient or nothing).
"""
function _upload(system, eligible, θs; device::Integer=0)
    θ0 = first(θs)
    entries = Any[]; descs = OctoObsDesc[]; columns = Vector{Float64}[]
    p(c) = isempty(c) ? Ptr{Float64}(C_NULL) : pointer(c)
    for (obs, ip) in eligible
        kind, planet, cols, coef = _table(obs, max(ip, 1), [_θobs(θ, obs, ip) for θ in θs])
        append!(columns, cols)
        push!(descs, OctoObsDesc(kind, planet, length(cols[1]), p(cols[1]), p(cols[2]), p(cols[3]), p(cols[4]), p(cols[5]), p(cols[6]), p(cols[7]), length(cols[7])))
        push!(entries, (obs, ip, Symbol(normalizename(likelihoodname(obs))), coef))
    end
    planets = OctoPlanetDesc[]
    for (ip, pl) in enumerate(system.planets)
        push!(planets, OctoPlanetDesc(_orbit_kind(pl), hasproperty(θ0.planets[ip], :mass)))                 # relative-astrometry.jl:122
    end
    ctx = octo_ctx_create(device)
    try
        octo_consts_set(ctx, _consts())
        ds = GC.@preserve columns octo_dataset_create(ctx, descs, planets)
        return ctx, ds, entries, columns
    catch
        octo_ctx_destroy(ctx)                                              # a refused dataset must not leak the context it was offered to
        rethrow()
    end
end

"""
Why a system cannot go to the device at all, or `nothing`: SURVEY.md §8(b) — the shim FALLS BACK to the reference closure, it does not throw.
The library compiles its epoch-loop kernels for 1 … OCTO_MAX_PLANETS planets (the reference unrolls over any number, system.jl:116-118,156-170).
"""
function _not_on_device(system)
    np = length(system.planets)
    np < 1 && return "the system has no planet"
    np > OCTO_MAX_PLANETS && return "$np planets: the device kernels are compiled for at most $OCTO_MAX_PLANETS"
    return nothing
end

"Resolved orbital elements and nuisances of ONE parameter set, in C-ABI order; generic in the number type (Float64 or Dual)."
function kernel_inputs(system, entries, θ)
    T = Octofitter._system_number_type(θ)
    nP = length(system.planets)
    x = Vector{T}(undef, nP * N_EL + length(entries) * N_NUIS)
    for ip in 1:nP
        θp = merge(θ, θ.planets[ip])                                    # system.jl:117
        keys = Octofitter.orbittype(system.planets[ip]) <: ThieleInnesOrbit ? EL_KEYS_TI : EL_KEYS
        for (k, key) in enumerate(keys)
            x[(ip-1)*N_EL+k] = hasproperty(θp, key) ? getproperty(θp, key) : zero(T)
        end
    end
    o0 = nP * N_EL
    for (io, (obs, ip, key, trendcoef)) in enumerate(entries)
        src = ip > 0 ? θ.planets[ip].observations : θ.observations
        θobs = hasproperty(src, key) ? getproperty(src, key) : (;)
        T1 = nameof(typeof(obs))
        if T1 === :HGCAInstantaneousObs                                   # θ_system.pmra / .pmdec, hgca.jl:266-267
            x[o0+(io-1)*N_NUIS+1] = θ.pmra; x[o0+(io-1)*N_NUIS+2] = θ.pmdec; x[o0+(io-1)*N_NUIS+3] = zero(T)
        elseif obs isa PlanetRelAstromObs || T1 === :ObsPriorAstromONeil2019   # relative-astrometry.jl:170-172
            x[o0+(io-1)*N_NUIS+1] = hasproperty(θobs, :jitter) ? θobs.jitter : zero(T)
            x[o0+(io-1)*N_NUIS+2] = hasproperty(θobs, :platescale) ? θobs.platescale : one(T)
            x[o0+(io-1)*N_NUIS+3] = hasproperty(θobs, :northangle) ? θobs.northangle : zero(T)
        else
            x[o0+(io-1)*N_NUIS+1] = hasproperty(θobs, :offset) ? θobs.offset : zero(T)
            x[o0+(io-1)*N_NUIS+2] = hasproperty(θobs, :jitter) ? θobs.jitter : zero(T)
            x[o0+(io-1)*N_NUIS+3] = trendcoef === nothing ? zero(T) : getproperty(θobs, trendcoef)      # OCTO_NU_RV_TREND
        end
    end
    return x
end

# ==================================================================================================== (1) accelerate(system)
"One evaluation slot: a context (its own stream and scratch) and the staging arrays of the W = 1 call."
struct HIPSlot
    ctx::Ptr{Cvoid}
    Xt::Matrix{Float64}               # 1 × inputs
    G::Matrix{Float64}
    ll::Vector{Float64}
end

"""
State shared by the HIPObs wrappers of one system: ONE dataset (immutable once uploaded, shareable between contexts — the header's
threading contract) and a pool of slots. `ℓπcallback` is called concurrently from many Julia threads (`guess_starting_position`,
src/initialization.jl:33-48; Pigeons with `multithreaded=true`): each call checks a slot out of `free`, so up to `length(slots)`
evaluations are in flight on the device at once and none of them shares scratch.
"""
mutable struct HIPShared
    ds::Ptr{Cvoid}
    slots::Vector{HIPSlot}
    free::Channel{Int}
    system::Any                       # the accelerated System (set after construction)
    entries::Vector{Any}              # (wrapped obs, i_planet or 0, θ_obs key) in evaluation order
    columns::Vector{Vector{Float64}}
    n_el::Int
    nuis_default::Vector{Float64}     # the value each nuisance input has when the model does not declare it; NaN = always pass (HGCA's pmra, pmdec)
end

"""
    HIPObs(obs, shared, leader)

Wrapper of an observation whose `ln_like` runs on the device. The `leader` (first wrapped observation in the evaluation order of
src/likelihoods/system.jl:229-235) evaluates ALL wrapped observations in one call; the others contribute zero. It keeps the wrapped
observation's `priors` / `derived` (so the model's variables are unchanged) and its name, but exposes no `table`, so the reference
gathers no epochs for it and pre-solves nothing on the CPU.
"""
struct HIPObs{TObs<:AbstractObs} <: AbstractObs
    wrapped_like::TObs
    priors::Priors
    derived::Union{Derived,Nothing}
    shared::HIPShared
    leader::Bool
end
Octofitter.likelihoodname(obs::HIPObs) = likelihoodname(obs.wrapped_like)
Octofitter._isprior(::HIPObs) = false
Octofitter.likeobj_from_epoch_subset(obs::HIPObs, inds) = Octofitter.likeobj_from_epoch_subset(obs.wrapped_like, inds)   # cross-validation falls back to the CPU

_eligible(obs, ip, θs) = _has_epochs(obs) && _table(obs, max(ip, 1), [_θobs(θ, obs, ip) for θ in θs]) !== nothing

"""
    accelerate(system::System; device=0) -> System

The same model with every observation the kernels implement wrapped in `HIPObs`. Epoch-free terms (UnitLengthPrior, UserLikelihood,
PlanetOrderPrior, …) stay as they are and run in Julia. Any OTHER epoch-bearing observation (full HGCA line fit, Hipparcos/Gaia
IAD, GP RV, images, …) also stays in Julia and the reference keeps solving the orbits for its epochs — the two mix freely because
each `ln_like` method is independent (src/variables.jl:94-102).
"""
function accelerate(system::System; device::Integer=0, n_contexts::Integer=Threads.nthreads(), verbosity::Integer=1)
    θs = _θ_draws(system)
    eligible = Any[]
    for (ip, pl) in enumerate(system.planets), obs in pl.observations
        _eligible(obs, ip, θs) && push!(eligible, (obs, ip))
    end
    for obs in system.observations
        _eligible(obs, 0, θs) && push!(eligible, (obs, 0))
    end
    isempty(eligible) && (verbosity >= 1 && @info "OctofitterHIP: no observation of this system is on the HIP path"; return system)
    # ---- fall back to the un-accelerated system (the reference's own closure) instead of throwing: more planets than the kernels are
    # compiled for, no usable device, or a dataset the library refuses (SURVEY.md §8(b)); nothing is leaked on any of these paths
    why = _not_on_device(system)
    if why !== nothing
        verbosity >= 1 && @info "OctofitterHIP: $why — the system stays on the reference's CPU path"
        return system
    end
    local ctx, ds, entries, columns
    try
        ctx, ds, entries, columns = _upload(system, eligible, θs; device)
    catch e
        # OCTO_ENODEV / OCTO_ENOTSUP: the library's refusal of a VALID system (no device, a kind set it does not take for this many planets, …);
        # NotOnHIPPath: this shim's own (an orbit type that is not on the HIP path, `_orbit_kind`). Anything else — bad input (OCTO_EINVAL: σ <= 0,
        # non-finite epochs, |cor| >= 1), OCTO_EHIP, OCTO_ENOMEM, a bug in this file — is the caller's to see (ADVICE r5)
        is_fallback(e) || rethrow()
        verbosity >= 1 && @info "OctofitterHIP: $(sprint(showerror, e)) — the system stays on the reference's CPU path"
        return system
    end
    n_in = length(system.planets) * N_EL + length(entries) * N_NUIS
    slot(c) = HIPSlot(c, Matrix{Float64}(undef, 1, n_in), Matrix{Float64}(undef, 1, n_in), Vector{Float64}(undef, 1))
    slots = [slot(ctx)]
    for _ in 2:max(1, n_contexts)                                       # more contexts on the same device share the dataset
        try
            c = octo_ctx_create(device)
            try octo_consts_set(c, _consts()) catch; octo_ctx_destroy(c); rethrow() end
            push!(slots, slot(c))
        catch e
            e isa OctoError || rethrow()
            break                                                        # fewer slots than threads: callers queue on `free`, nothing else changes
        end
    end
    free = Channel{Int}(length(slots)); foreach(i -> put!(free, i), eachindex(slots))
    nuis_default = Float64[]
    for (obs, _, _, _) in entries
        T1 = nameof(typeof(obs))
        append!(nuis_default, T1 === :HGCAInstantaneousObs ? (NaN, NaN, 0.0) :
                              (obs isa PlanetRelAstromObs || T1 === :ObsPriorAstromONeil2019) ? (0.0, 1.0, 0.0) : (0.0, 0.0, 0.0))
    end
    shared = HIPShared(ds, slots, free, nothing, entries, columns, length(system.planets) * N_EL, nuis_default)
    finalizer(shared) do s
        octo_dataset_destroy(s.ds); foreach(x -> octo_ctx_destroy(x.ctx), s.slots)
    end
    first_seen = Ref(false)
    wrapped = Set(objectid(e[1]) for e in entries)
    wrap(obs) = objectid(obs) in wrapped ? (l = !first_seen[]; first_seen[] = true; HIPObs(obs, obs.priors, obs.derived, shared, l)) : obs
    planets = map(system.planets) do pl
        obs2 = map(wrap, pl.observations)
        Planet{Octofitter.orbittype(pl),typeof(pl.priors),typeof(pl.derived),typeof(obs2)}(pl.priors, pl.derived, obs2, pl.name)
    end
    sysobs = map(wrap, system.observations)
    sys2 = System(system.priors, system.derived, sysobs, planets, system.name)
    shared.system = sys2
    verbosity >= 1 && @info "OctofitterHIP: $(length(entries)) observation table(s), $(octo_dataset_n_rows(ds)) epochs on the device, $(length(slots)) context(s)"
    return sys2
end

_value(x::Real) = Float64(x)
_value(x::ForwardDiff.Dual) = Float64(ForwardDiff.value(x))

"ll and (when the inputs carry partials) its Dual: partials(ll) = Σ_k ∂ll/∂input_k · partials(input_k)."
function _ln_like_all(sh::HIPShared, θ_system)
    x = kernel_inputs(sh.system, sh.entries, θ_system)
    T = eltype(x)
    i = take!(sh.free)                                                    # a context of our own for the duration of the call
    try
        sl = sh.slots[i]
        @inbounds for k in eachindex(x)
            sl.Xt[1, k] = _value(x[k])
        end
        # every nuisance at its default (the model declares none): the nuisance block is not passed at all
        with_nuis = any(k -> !(sl.Xt[1, sh.n_el+k] == sh.nuis_default[k]), eachindex(sh.nuis_default))
        n_used = with_nuis ? length(x) : sh.n_el
        if T <: ForwardDiff.Dual
            octo_eval!(sl.ctx, sh.ds, sl.Xt, sh.n_el, sl.ll, sl.G; with_nuis)
            ll = sl.ll[1]
            isfinite(ll) || return T(ll)                                  # -Inf: zero partials (logdensitymodel.jl:120-124)
            p = zero(ForwardDiff.partials(x[1]))
            @inbounds for k in 1:n_used
                p += sl.G[1, k] * ForwardDiff.partials(x[k])
            end
            return T(ll, p)
        else
            octo_eval!(sl.ctx, sh.ds, sl.Xt, sh.n_el, sl.ll, nothing; with_nuis)
            return T(sl.ll[1])
        end
    finally
        put!(sh.free, i)
    end
end

# The plug-in interface of the reference (src/variables.jl:94-102): one method per context type.
function Octofitter.ln_like(obs::HIPObs, ctx::Octofitter.PlanetObservationContext)
    obs.leader || return zero(Octofitter._system_number_type(ctx.θ_system))
    return _ln_like_all(obs.shared, ctx.θ_system)
end
function Octofitter.ln_like(obs::HIPObs, ctx::Octofitter.SystemObservationContext)
    obs.leader || return zero(Octofitter._system_number_type(ctx.θ_system))
    return _ln_like_all(obs.shared, ctx.θ_system)
end
# posterior-predictive simulation and plotting use the wrapped observation on the CPU
Octofitter.generate_from_params(obs::HIPObs, args...; kwargs...) = Octofitter.generate_from_params(obs.wrapped_like, args...; kwargs...)

# ==================================================================================================== (3) batches of structured θ
mutable struct GPUBatchedLikelihood{TModel}
    model::TModel
    ctx::Ptr{Cvoid}
    ds::Ptr{Cvoid}
    n_planets::Int
    obs_entries::Vector{Any}          # (obs, i_planet or 0, θ_obs key) in evaluation order
    host_terms::Vector{Any}           # epoch-free observations evaluated in Julia
    columns::Vector{Vector{Float64}}  # keeps the uploaded host columns alive
end

function GPUBatchedLikelihood(model; device::Integer=0)
    system = model.system
    rng = Random.Xoshiro(20260929)
    θs = [model.arr2nt(model.sample_priors(rng)) for _ in 1:3]
    eligible = Any[]; host_terms = Any[]
    add! = function (obs, ip, ctxkind)
        obs isa HIPObs && (obs = obs.wrapped_like)
        if !_has_epochs(obs)
            push!(host_terms, (obs, ip, ctxkind)); return
        end
        _eligible(obs, ip, θs) || throw(NotOnHIPPath("observation $(likelihoodname(obs)) is not on the HIP path (GP, or a trend_function that is not one θ_obs variable × a function of the epoch); keep using model.ℓπcallback"))
        push!(eligible, (obs, ip))
    end
    # evaluation order of the generated closure: planet observations planet by planet, then system ones (system.jl:229-235)
    for (ip, pl) in enumerate(system.planets), obs in pl.observations
        add!(obs, ip, :planet)
    end
    for obs in system.observations
        add!(obs, 0, :system)
    end
    ctx, ds, entries, columns = _upload(system, eligible, θs; device)
    g = GPUBatchedLikelihood(model, ctx, ds, length(system.planets), entries, host_terms, columns)
    finalizer(g) do x
        octo_dataset_destroy(x.ds); octo_ctx_destroy(x.ctx)
    end
    return g
end

kernel_inputs(g::GPUBatchedLikelihood, θ) = kernel_inputs(g.model.system, g.obs_entries, θ)

"Host-side (epoch-free) likelihood terms, summed exactly as the reference does."
function host_ll(g::GPUBatchedLikelihood, θ)
    ll = zero(Octofitter._system_number_type(θ))
    sysm = g.model.system
    for (obs, ip, kind) in g.host_terms
        key = normalizename(likelihoodname(obs))
        orbits = ntuple(i -> Octofitter.orbittype(sysm.planets[i])(; merge(θ, θ.planets[i])...), g.n_planets)
        if kind === :planet
            src = θ.planets[ip].observations
            θobs = hasproperty(src, key) ? getproperty(src, key) : (;)
            ll += Octofitter.ln_like(obs, Octofitter.PlanetObservationContext(θ, θ.planets[ip], θobs, orbits, ntuple(_ -> (), g.n_planets), ip, -1))
        else
            θobs = hasproperty(θ.observations, key) ? getproperty(θ.observations, key) : (;)
            ll += Octofitter.ln_like(obs, Octofitter.SystemObservationContext(θ, θobs, orbits, ntuple(_ -> (), g.n_planets), -1))
        end
    end
    return ll
end

"X :: inputs × W, column per walker. Returns ll[W] and, if grad, ∂ll/∂X of the same shape."
function eval_inputs(g::GPUBatchedLikelihood, X::Matrix{Float64}; grad::Bool=false)
    Xt = permutedims(X)                                  # [W, inputs]: walker index fastest, as the C ABI wants
    ll = Vector{Float64}(undef, size(X, 2))
    G = grad ? similar(Xt) : nothing
    octo_eval!(g.ctx, g.ds, Xt, g.n_planets * N_EL, ll, G)
    return grad ? (ll, permutedims(G)) : ll
end

"ln_like for a batch of structured parameter sets (what `make_ln_like(system, θ)(system, θ)` returns, W at a time)."
function ln_like_batch(g::GPUBatchedLikelihood, Θ::AbstractVector)
    X = reduce(hcat, (Float64.(kernel_inputs(g, θ)) for θ in Θ))
    ll = eval_inputs(g, X)
    isempty(g.host_terms) || (ll .+= (host_ll(g, θ) for θ in Θ))
    return ll
end

"Drop-in for the closure returned by make_ln_like (src/likelihoods/system.jl:206): one θ at a time."
(g::GPUBatchedLikelihood)(system, θ) = ln_like_batch(g, [θ])[1]

"Batched `pointwise_like` column (src/cross-validation.jl:34-46): the log-likelihood of every posterior sample, one device call."
pointwise_like_batch(g::GPUBatchedLikelihood, sample_nts::AbstractVector) = ln_like_batch(g, sample_nts)

# ==================================================================================================== (2) HIPLogDensityModel
function _octo_prior(d::Distribution)
    d isa Uniform && return OctoPrior(PRIOR_UNIFORM, 0, minimum(d), maximum(d), -Inf, Inf)
    d isa LogUniform && return OctoPrior(PRIOR_LOGUNIFORM, 0, minimum(d), maximum(d), -Inf, Inf)
    d isa Normal && return OctoPrior(PRIOR_NORMAL, 0, mean(d), std(d), -Inf, Inf)
    if d isa Truncated && d.untruncated isa Normal
        lo = d.lower === nothing ? -Inf : Float64(d.lower); hi = d.upper === nothing ? Inf : Float64(d.upper)
        return OctoPrior(PRIOR_TRUNCNORMAL, 0, mean(d.untruncated), std(d.untruncated), lo, hi)
    end
    nameof(typeof(d)) === :Sine && return OctoPrior(PRIOR_SINE, 0, 0.0, 0.0, -Inf, Inf)     # Octofitter.Sine, src/distributions.jl:14-39
    return nothing
end

"All priors in the reference's flattening order (src/variables.jl:1205-1347): system, system observations, then per planet its own and its observations'."
function _flat_priors(system)
    ds = Distribution[]
    append!(ds, values(system.priors.priors))
    for obs in system.observations
        hasproperty(obs, :priors) && append!(ds, values(obs.priors.priors))
    end
    for pl in system.planets
        append!(ds, values(pl.priors.priors))
        for obs in pl.observations
            hasproperty(obs, :priors) && append!(ds, values(obs.priors.priors))
        end
    end
    return ds
end

"""
How is kernel input k built from the natural θ? Found NUMERICALLY, not from the user's expressions: the Jacobian of
`kernel_inputs ∘ arr2nt` at a few prior draws tells which θ an input depends on, and the candidate closed forms (identity,
constant, atan(y, x)/2π·domain, θ_at_epoch_to_tperi) are checked against its values. Anything else: `nothing` (not a standard model).
"""
function _classify_sources(model, g::GPUBatchedLikelihood; n_probe::Int=4)
    D = model.D
    rng = Random.Xoshiro(20260929)
    θs = [model.sample_priors(rng) for _ in 1:n_probe]
    f(θ) = kernel_inputs(g, model.arr2nt(θ))
    vals = [Float64.(f(θ)) for θ in θs]
    jacs = [ForwardDiff.jacobian(f, collect(Float64, θ)) for θ in θs]
    n_in = length(vals[1]); n_el = g.n_planets * N_EL
    src = Vector{Union{Nothing,OctoSource}}(nothing, n_in)
    used_pairs = Set{Tuple{Int,Int}}()
    dep(k) = findall(d -> any(abs(J[k, d]) > 0 for J in jacs), 1:D)
    for k in 1:n_in
        deps = dep(k)
        if isempty(deps)
            all(v -> v[k] == vals[1][k], vals) || return nothing
            src[k] = OctoSource(SRC_CONST, 0, 0, 0, vals[1][k])
        elseif length(deps) == 1 && all(i -> vals[i][k] == θs[i][deps[1]], 1:n_probe)
            src[k] = OctoSource(SRC_THETA, deps[1] - 1, 0, 0, 0.0)
        elseif length(deps) == 2
            ix, iy = deps                                               # UniformCircular: (x, y) are declared in this order, variables.jl:290-293
            ang(i) = atan(θs[i][iy], θs[i][ix])
            dom = vals[1][k] / ang(1) * 2π
            all(i -> isapprox(vals[i][k], ang(i) / 2π * dom; rtol=1e-12, atol=1e-14), 1:n_probe) || return nothing
            flag = (ix, iy) in used_pairs ? Int32(0) : SRC_FLAG_UNITLEN
            push!(used_pairs, (ix, iy))
            src[k] = OctoSource(SRC_CIRCULAR, ix - 1, iy - 1, flag, dom)
        end
    end
    # what is left must be tp = θ_at_epoch_to_tperi(θ, epoch; …) of a planet whose other elements are already classified
    for k in 1:n_in
        src[k] === nothing || continue
        (k <= n_el && (k - 1) % N_EL + 1 == 6) || return nothing
        ip = (k - 1) ÷ N_EL + 1
        ti = Octofitter.orbittype(model.system.planets[ip]) <: ThieleInnesOrbit
        others = Set(Iterators.flatten(dep(j) for j in (ip-1)*N_EL+1:ip*N_EL if j != k))
        pair = sort(setdiff(dep(k), others))
        length(pair) == 2 || return nothing
        ix, iy = pair
        # epoch = tp + MA/n·year2day with MA, n from the planet's elements at θ = atan(y, x): the same for every probe
        function epoch_of(i)
            el = vals[i][(ip-1)*N_EL+1:ip*N_EL]; θang = atan(θs[i][iy], θs[i][ix])
            nt = ti ? (; plx=el[8], M=el[7], e=el[2], A=el[1], B=el[3], F=el[4], G=el[5]) : (; M=el[7], e=el[2], a=el[1], i=el[3], ω=el[4], Ω=el[5])
            return el[6] - (Octofitter.θ_at_epoch_to_tperi(θang, 0.0; nt...))     # linear in the epoch argument
        end
        ep = epoch_of(1)
        all(i -> isapprox(epoch_of(i), ep; rtol=0, atol=1e-6), 1:n_probe) || return nothing
        flag = ((ix, iy) in used_pairs ? Int32(0) : SRC_FLAG_UNITLEN) | (ti ? SRC_FLAG_TI : Int32(0))
        push!(used_pairs, (ix, iy))
        src[k] = OctoSource(SRC_TPERI, ix - 1, iy - 1, flag, round(ep; digits=6))
    end
    return OctoSource[s for s in src]
end

"""
    HIPLogDensityModel(model::Octofitter.LogDensityModel; device=0)

The whole log-posterior callback on the device for models made of the standard blocks; throws (and the caller keeps `model`,
possibly built from `accelerate(system)`) when a prior family, a Derived expression or an observation is outside them.
"""
mutable struct HIPLogDensityModel{TModel,Tℓπ,T∇ℓπ}
    const D::Int
    const ℓπcallback::Tℓπ
    const ∇ℓπcallback::T∇ℓπ
    const system::Any
    const link::Any
    const invlink::Any
    const arr2nt::Any
    const sample_priors::Any
    starting_points::Union{Nothing,Vector}
    const reference::TModel
    const batched::GPUBatchedLikelihood
    const m::Ptr{Cvoid}
    const lock::ReentrantLock
end

function HIPLogDensityModel(model; device::Integer=0, fallback::Bool=true, verbosity::Integer=1)
    # SURVEY.md §8(b): fall back to the reference's model instead of throwing — more planets than the kernels are compiled for, no usable
    # device, a dataset or model the library refuses, a prior / Derived expression / observation outside the standard blocks. `fallback = false`
    # (tests, debugging) lets the exception through.
    fallback || return _hip_log_density_model(model; device)
    why = _not_on_device(model.system)
    if why === nothing
        try
            return _hip_log_density_model(model; device)
        catch e
            is_fallback(e) || rethrow()
            why = sprint(showerror, e)
        end
    end
    verbosity >= 1 && @info "OctofitterHIP: $why — returning the reference's LogDensityModel unchanged"
    return model
end

function _hip_log_density_model(model; device::Integer=0)
    g = GPUBatchedLikelihood(model; device)
    # the UnitLengthPrior terms of UniformCircular variables are part of the device model; any other host term is not
    for (obs, _, _) in g.host_terms
        nameof(typeof(obs)) === :UnitLengthPrior || throw(NotOnHIPPath("$(typeof(obs)) is evaluated in Julia: use LogDensityModel(accelerate(system)) for this model"))
    end
    dists = _flat_priors(model.system)
    length(dists) == model.D || throw(NotOnHIPPath("model has multivariate or discrete priors: not a standard-parameterisation model"))
    priors = OctoPrior[]
    for d in dists
        p = _octo_prior(d)
        p === nothing && throw(NotOnHIPPath("prior $(d) has no device counterpart: use LogDensityModel(accelerate(system)) for this model"))
        push!(priors, p)
    end
    srcs = _classify_sources(model, g)
    srcs === nothing && throw(NotOnHIPPath("a Derived variable of this model is not one of the standard blocks: use LogDensityModel(accelerate(system))"))
    n_el = g.n_planets * N_EL
    m = octo_model_create(g.ctx, g.ds, priors, srcs[1:n_el], length(srcs) > n_el ? srcs[n_el+1:end] : nothing)
    lk = ReentrantLock()
    D = model.D
    function ℓπ(θ_t::AbstractVector)
        lock(lk) do
            lp = Vector{Float64}(undef, 1)
            octo_model_logpost!(g.ctx, m, reshape(collect(Float64, θ_t), 1, D), lp, nothing)[1]
        end
    end
    function ℓπ(Θ_t::AbstractMatrix)                                    # D × W, as the reference lays batches out
        lock(lk) do
            octo_model_logpost!(g.ctx, m, permutedims(Matrix{Float64}(Θ_t)), Vector{Float64}(undef, size(Θ_t, 2)), nothing)
        end
    end
    function ∇ℓπ(θ_t::AbstractVector)
        lock(lk) do
            lp = Vector{Float64}(undef, 1); G = Matrix{Float64}(undef, 1, D)
            octo_model_logpost!(g.ctx, m, reshape(collect(Float64, θ_t), 1, D), lp, G)
            (lp[1], vec(G))
        end
    end
    function ∇ℓπ(Θ_t::AbstractMatrix)
        lock(lk) do
            W = size(Θ_t, 2); lp = Vector{Float64}(undef, W); G = Matrix{Float64}(undef, W, D)
            octo_model_logpost!(g.ctx, m, permutedims(Matrix{Float64}(Θ_t)), lp, G)
            (lp, permutedims(G))
        end
    end
    hm = HIPLogDensityModel(D, ℓπ, ∇ℓπ, model.system, model.link, model.invlink, model.arr2nt, model.sample_priors, model.starting_points,
                            model, g, m, lk)
    finalizer(x -> octo_model_destroy(x.m), hm)
    return hm
end

# The interface AdvancedHMC / Pathfinder / Pigeons see (src/logdensitymodel.jl:252-256, OctofitterPigeonsExt.jl:10-12)
LogDensityProblems.logdensity(p::HIPLogDensityModel, θ) = p.ℓπcallback(θ)
LogDensityProblems.logdensity_and_gradient(p::HIPLogDensityModel, θ) = p.∇ℓπcallback(θ)
LogDensityProblems.dimension(p::HIPLogDensityModel) = p.D
LogDensityProblems.capabilities(::Type{<:HIPLogDensityModel}) = LogDensityProblems.LogDensityOrder{1}()
(p::HIPLogDensityModel)(θ) = p.ℓπcallback(θ)

"`guess_starting_position` (src/initialization.jl:14-66) with the N prior draws evaluated in device batches instead of one callback each."
function Octofitter.guess_starting_position(rng::Random.AbstractRNG, model::HIPLogDensityModel, N=500_000; batch::Int=250_000)
    bestparams = model.sample_priors(rng); bestlogpost = -Inf64
    done = 0
    while done < N
        n = min(batch, N - done)
        params = [model.sample_priors(rng) for _ in 1:n]
        Θt = reduce(hcat, (collect(Float64, model.link(p)) for p in params))
        logpost = model.ℓπcallback(Θt)
        k = argmax(logpost)
        if logpost[k] > bestlogpost
            bestlogpost = logpost[k]; bestparams = params[k]
        end
        done += n
    end
    return bestparams, bestlogpost
end

"`_rejection_evaluate_likelihoods` (src/sampling.jl:260-268) for a vector of prior draws, one device call."
function rejection_evaluate_likelihoods(model::HIPLogDensityModel, prior_samples::AbstractVector)
    ll = ln_like_batch(model.batched, [model.arr2nt(θ) for θ in prior_samples])
    return map(x -> isfinite(x) ? x : -Inf, ll)
end

# ---------------------------------------------------------------------------------------------------- parallel tempering, batched (BASELINE config 5)
# `octofit_pigeons` (ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:76-128) builds Pigeons.Inputs and leaves the replicas to Pigeons.jl, whose explorers
# call model(θ) ONE replica at a time (:10-12). This driver owns the loop instead, so that every exploration step evaluates ALL local replicas as one
# [D][W] batch on the device and every communication step is one collective:
#   protocol (shared with host/tempering.py: TemperedSwap, its executable twin — tests/test_multi_gpu.py::test_host_swap_step_is_the_device_swap_step):
#   * one process per GPU; temperatures (replicas) split contiguously over the ranks, every rank holds all `n_chains` independent chains of its replicas:
#     local walker w = r_local·n_chains + c  (column w of θ_t, 1-based w + 1 in Julia);
#   * bootstrap: rank 0 draws the 128-byte id (octo_comm_unique_id), `bcast_id` carries it to the others (MPI.Bcast!, a file, a socket), every rank joins
#     with octo_comm_create(ctx, id, rank, world); world = 1 needs no id and no RCCL;
#   * exploration: a random-walk Metropolis step on θ_t for every local replica at its current β — log u < (ℓprior′ + β ℓ′) − (ℓprior + β ℓ) — with
#     ℓπ = octo_model_logpost! on the whole batch and ℓprior from the reference's own make_ln_prior_transformed, so ℓ = ℓπ − ℓprior (Pigeons' reference chain
#     is the prior, OctofitterPigeonsExt.jl:61-67);
#   * communication: octo_pt_step(ctx, ℓ_local, β, slot2rep, …, parity = scan % 2, seed, scan, accepted): all-gather of the local ℓ (RCCL) + the
#     deterministic neighbour swap of β LABELS (states never move) — same seed and scan on every rank, so every rank ends with the same slot2rep.
# Returns (θ_t, ℓπ, slot2rep, accepted, β) of this rank; slot2rep[c, t] = the replica that sits at ladder slot t of chain c.
function octofit_pigeons_hip(hm::HIPLogDensityModel; n_rounds::Integer=8, n_temps::Integer=16, n_chains::Integer=64, seed::Integer=1,
                             explorer_steps::Integer=3, step_size::Real=0.05, rank::Integer=0, world::Integer=1,
                             bcast_id=identity, rng::Random.AbstractRNG=Random.Xoshiro(seed + rank), verbosity::Integer=1)
    n_temps % world == 0 || error("n_temps must divide evenly over the ranks")
    ctx, D = hm.batched.ctx, hm.D
    id = world > 1 ? bcast_id(rank == 0 ? octo_comm_unique_id() : zeros(UInt8, 128)) : nothing
    octo_comm_create(ctx, id, rank, world)
    n_loc = n_temps ÷ world
    W = n_loc * n_chains
    β = collect(range(1.0, 0.0; length=n_temps)) .^ 3                                   # β_1 = 1: the target; the last slot is the prior
    slot2rep = Matrix{Int32}(undef, n_temps, n_chains)                                  # column-major [t, c] = the C ABI's [n_chains][n_temps]
    for c in 1:n_chains, t in 1:n_temps
        slot2rep[t, c] = t - 1
    end
    accepted = zeros(Int32, n_temps)
    θ = Matrix{Float64}(undef, W, D)                                                    # column d = row d of the C ABI's [D][ld = W]
    for w in 1:W
        θ[w, :] .= hm.link(hm.sample_priors(rng))
    end
    θ′ = similar(θ); ℓπ = Vector{Float64}(undef, W); ℓπ′ = similar(ℓπ)
    ln_prior_transformed = Octofitter.make_ln_prior_transformed(hm.system)             # the reference's own prior density (src/logdensitymodel.jl:47,128)
    ℓprior(θm, w) = ln_prior_transformed(hm.invlink(θm[w, :]), true)
    lpr = [ℓprior(θ, w) for w in 1:W]; lpr′ = similar(lpr)
    lock(hm.lock) do; octo_model_logpost!(ctx, hm.m, θ, ℓπ, nothing); end
    ℓ = ℓπ .- lpr
    try
        scan = 0
        for round in 1:n_rounds, _ in 1:2^round
            scan += 1
            # β of every local walker under the current labels: replica r (0-based, global) of chain c sits at the slot t with slot2rep[t, c] == r
            βw = Vector{Float64}(undef, W)
            for c in 1:n_chains, t in 1:n_temps
                r = Int(slot2rep[t, c]) - rank * n_loc
                0 <= r < n_loc && (βw[r * n_chains + c] = β[t])
            end
            for _ in 1:explorer_steps
                θ′ .= θ .+ step_size .* randn(rng, W, D)
                lock(hm.lock) do; octo_model_logpost!(ctx, hm.m, θ′, ℓπ′, nothing); end   # ONE batched evaluation of every local replica
                for w in 1:W
                    lpr′[w] = ℓprior(θ′, w)
                    ℓ′ = ℓπ′[w] - lpr′[w]
                    if isfinite(ℓπ′[w]) && log(rand(rng)) < (lpr′[w] + βw[w] * ℓ′) - (lpr[w] + βw[w] * ℓ[w])
                        θ[w, :] .= view(θ′, w, :); ℓπ[w] = ℓπ′[w]; lpr[w] = lpr′[w]; ℓ[w] = ℓ′
                    end
                end
            end
            octo_pt_step(ctx, ℓ, β, slot2rep, n_temps, n_chains, scan % 2, UInt64(seed), UInt64(scan), accepted)
        end
        verbosity >= 1 && @info "octofit_pigeons_hip: rank $rank of $world, $scan scans, swap acceptances per slot $(accepted)"
    finally
        octo_comm_destroy(ctx)
    end
    return (; θ_t=θ, ℓπ, slot2rep, accepted, β)
end

export octofit_pigeons_hip
export accelerate, HIPObs, HIPLogDensityModel, GPUBatchedLikelihood, ln_like_batch, pointwise_like_batch, rejection_evaluate_likelihoods
end # module
