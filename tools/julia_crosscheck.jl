# tools/julia_crosscheck.jl — close the parity gap for someone WHO HAS JULIA (the build image does not):
# evaluate the committed fixtures with the REAL reference (Octofitter.jl + PlanetOrbits.jl + Distributions.jl) and write
# tests/golden/reference_dump.json. `pytest tests/test_oracle.py::test_reference_dump_if_present` then holds the CPU oracle
# (and, on a GPU box, tests/test_gpu_parity.py::test_reference_dump_if_present_gpu the HIP path) to those numbers.
#
#   julia --project=<env with Octofitter, OctofitterRadialVelocity, JSON, TypedTables, ForwardDiff> tools/julia_crosscheck.jl
#
# NOT EXECUTED HERE. What it covers, per fixture family (SURVEY.md §8c last row):
#   model.json / config1.json  the D = 11 model of test/integration/sampling.jl:29-64 built with the reference's own macros:
#                              model.ℓπcallback(θ_t) and model.∇ℓπcallback(θ_t) for every committed θ_t — pins priors, bijectors
#                              (logpdf_with_trans / TruncatedBijector), UniformCircular + UnitLengthPrior, θ_at_epoch_to_tperi,
#                              the orbit constructor, the Kepler solve, the projection and the Gaussian density in one number;
#   fixtures.json F1-F4, F8    ln_like of PlanetRelAstromObs (both table formats, cor, jitter / platescale / northangle) and the
#                              O'Neil wrapper, at the committed elements, through `Octofitter.ln_like(obs, ctx)` with solutions from
#                              `orbitsolve`;
#   fixtures.json F5, kep.json StarAbsoluteRVObs / MarginalizedStarAbsoluteRVObs / PlanetRelativeRVObs on RadialVelocityOrbit,
#                              Visual{KepOrbit} and KepOrbit planets;
#   trend.json F13             the same three RV types with a `trend_function` (the model of OctofitterRadialVelocity/test/runtests.jl:168-231).
# HGCA (needs the catalogue download) and the two-planet cases are left to the restatement; extend `run_case` the same way.
using Octofitter, OctofitterRadialVelocity, PlanetOrbits, TypedTables, ForwardDiff, JSON, Distributions

const ROOT = normpath(joinpath(@__DIR__, ".."))
golden(name) = JSON.parsefile(joinpath(ROOT, "tests", "golden", name))
col(ob, k) = Float64[x for x in ob[k]]

function d11_model(ob)
    cols = (epoch=col(ob, "epoch"), ra=col(ob, "y1"), dec=col(ob, "y2"), σ_ra=col(ob, "s1"), σ_dec=col(ob, "s2"))
    tab = ob["cor"] === nothing ? Table(; cols...) : Table(; cols..., cor=col(ob, "cor"))
    astrom_like = PlanetRelAstromLikelihood(tab, name="sampling_test")
    b = Planet(name="b", basis=Visual{KepOrbit}, observations=[astrom_like], variables=@variables begin
        a ~ Uniform(0, 100)
        e ~ Uniform(0.0, 0.99)
        i ~ Sine()
        ω ~ UniformCircular()
        Ω ~ UniformCircular()
        θ ~ UniformCircular()
        tp = θ_at_epoch_to_tperi(θ, 50000; M=system.M, e, a, i, ω, Ω)
    end)
    sys = System(name="TestSys", companions=[b], observations=[], variables=@variables begin
        M ~ truncated(Normal(1.2, 0.1), lower=0.1)
        plx ~ truncated(Normal(50.0, 0.02), lower=0.1)
    end)
    return Octofitter.LogDensityModel(sys; verbosity=0)
end

function model_dump(case)
    model = d11_model(case["obs"][1])
    @assert model.D == 11
    Θ = reduce(hcat, [Float64[x for x in row] for row in case["theta_t"]])'      # D × W
    W = size(Θ, 2)
    lp = Float64[]; grad = Vector{Float64}[]
    for w in 1:W
        θt = collect(Θ[:, w])
        l, g = model.∇ℓπcallback(θt)
        @assert l == model.ℓπcallback(θt)
        push!(lp, l); push!(grad, collect(g))
    end
    return Dict("lp" => lp, "grad" => grad, "names" => string.(keys(Octofitter.flatten_named_tuple(model.arr2nt(model.invlink(collect(Θ[:, 1])))))))
end

orbit_of(kind, el) = kind == 0 ? Visual{KepOrbit}(; a=el[1], e=el[2], i=el[3], ω=el[4], Ω=el[5], tp=el[6], M=el[7], plx=el[8]) :
                     kind == 1 ? RadialVelocityOrbit(; a=el[1], e=el[2], ω=el[4], tp=el[6], M=el[7]) :
                     kind == 3 ? KepOrbit(; a=el[1], e=el[2], i=el[3], ω=el[4], Ω=el[5], tp=el[6], M=el[7]) :
                                 ThieleInnesOrbit(; A=el[1], e=el[2], B=el[3], F=el[4], G=el[5], tp=el[6], M=el[7], plx=el[8])

function make_obs(ob)
    k = ob["kind"]
    if k in ("ASTROM_RADEC", "ONEIL_RADEC", "ASTROM_SEPPA", "ONEIL_SEPPA")
        seppa = endswith(k, "SEPPA")
        cols = seppa ? (epoch=col(ob, "epoch"), pa=col(ob, "y1"), sep=col(ob, "y2"), σ_pa=col(ob, "s1"), σ_sep=col(ob, "s2")) :
                       (epoch=col(ob, "epoch"), ra=col(ob, "y1"), dec=col(ob, "y2"), σ_ra=col(ob, "s1"), σ_dec=col(ob, "s2"))
        tab = ob["cor"] === nothing ? Table(; cols...) : Table(; cols..., cor=col(ob, "cor"))
        o = PlanetRelAstromObs(tab, name="astrom")
        return startswith(k, "ONEIL") ? ObsPriorAstromONeil2019(o) : o
    end
    tab = Table(epoch=col(ob, "epoch"), rv=col(ob, "y1"), σ_rv=col(ob, "s1"))
    # F13 (trend.json): the fixture's `extra` column is trend_function(θ_obs with the coefficient = 1, epoch_j); rebuilt here as a closure
    # over (epoch -> basis) so that the REAL trend_function code path of the reference is what gets evaluated (rv-absolute.jl:143 etc.)
    basis = get(ob, "extra", nothing)
    kw = basis === nothing ? (;) : (; trend_function=let d = Dict(zip(col(ob, "epoch"), Float64[x for x in basis]))
        (θ_obs, epoch) -> θ_obs.trend_coef * d[epoch]
    end)
    k == "RV_ABS" && return StarAbsoluteRVObs(tab; name="rv", kw...)
    k == "RV_ABS_MARG" && return MarginalizedStarAbsoluteRVObs(tab; name="rv", kw...)
    k == "RV_REL" && return PlanetRelativeRVObs(tab; name="rv", kw...)
    return nothing
end

"Single-planet fixture cases: ln_like of every observation at every committed walker, through the reference's ln_like methods."
function run_case(case)
    length(case["planets"]) == 1 || return nothing
    obs = [make_obs(ob) for ob in case["obs"]]
    any(isnothing, obs) && return nothing
    elems = reduce(hcat, [Float64[x === nothing ? NaN : x for x in row] for row in case["elems"]])'      # 9 × W
    nuis = case["nuis"] === nothing ? nothing : reduce(hcat, [Float64[x for x in row] for row in case["nuis"]])'
    kind = case["planets"][1]["orbit_kind"]; has_mass = case["planets"][1]["has_mass"]
    function ll_of(x, w)      # x = the walker's [elements; nuisances] (Real or Dual)
        el = x[1:9]
        orbit = orbit_of(kind, el)
        θpl = has_mass ? (a=el[1], e=el[2], i=el[3], ω=el[4], Ω=el[5], tp=el[6], mass=el[9]) : (a=el[1], e=el[2], i=el[3], ω=el[4], Ω=el[5], tp=el[6])
        tot = zero(eltype(x))
        for (io, (o, ob)) in enumerate(zip(obs, case["obs"]))
            epochs = col(ob, "epoch")
            sols = [orbitsolve(orbit, t) for t in epochs]
            nu = nuis === nothing ? nothing : x[9 + 3 * (io - 1) + 1 : 9 + 3 * io]
            if o isa PlanetRelAstromObs || o isa ObsPriorAstromONeil2019
                θobs = nu === nothing ? (;) : (jitter=nu[1], platescale=nu[2], northangle=nu[3])
                θsys = (M=el[7], plx=el[8], planets=(b=merge(θpl, (observations=(astrom=θobs,),)),), observations=(;))
                tot += Octofitter.ln_like(o, Octofitter.PlanetObservationContext(θsys, θsys.planets.b, θobs, (orbit,), (sols,), 1, 0))
            elseif o isa PlanetRelativeRVObs
                θobs = nu === nothing ? (offset=0.0, jitter=0.0, trend_coef=0.0) : (offset=nu[1], jitter=nu[2], trend_coef=nu[3])
                θsys = (M=el[7], plx=el[8], planets=(b=merge(θpl, (observations=(rv=θobs,),)),), observations=(;))
                tot += Octofitter.ln_like(o, Octofitter.PlanetObservationContext(θsys, θsys.planets.b, θobs, (orbit,), (sols,), 1, 0))
            else
                θobs = nu === nothing ? (offset=0.0, jitter=0.0, trend_coef=0.0) : (offset=nu[1], jitter=nu[2], trend_coef=nu[3])
                θsys = (M=el[7], plx=el[8], planets=(b=θpl,), observations=(rv=θobs,))
                tot += Octofitter.ln_like(o, Octofitter.SystemObservationContext(θsys, θobs, (orbit,), (sols,), 0))
            end
        end
        return tot
    end
    W = size(elems, 2)
    ll = Float64[]; grad = Vector{Float64}[]
    for w in 1:W
        x = nuis === nothing ? collect(elems[:, w]) : vcat(collect(elems[:, w]), collect(nuis[:, w]))
        x[isnan.(x)] .= 1.0                                # rows the basis ignores
        push!(ll, ll_of(x, w)); push!(grad, ForwardDiff.gradient(y -> ll_of(y, w), x))
    end
    return Dict("ll" => ll, "grad_inputs" => grad)
end

out = Dict{String,Any}("octofitter_version" => string(pkgversion(Octofitter)), "planetorbits_version" => string(pkgversion(PlanetOrbits)),
                       "constants" => Dict("kepler_year_to_julian_day" => PlanetOrbits.kepler_year_to_julian_day_conversion_factor,
                                           "year2day_julian" => PlanetOrbits.year2day_julian, "au2m" => PlanetOrbits.au2m,
                                           "sec2year_julian" => PlanetOrbits.sec2year_julian, "pc2au" => PlanetOrbits.pc2au,
                                           "rad2as" => PlanetOrbits.rad2as, "mjup2msol" => Octofitter.mjup2msol))
out["model.json/D11_reference_test_model"] = model_dump(golden("model.json")["cases"][1])
out["config1.json/config1_D11_50_epochs"] = model_dump(golden("config1.json")["cases"][1])
for file in ("fixtures.json", "kep.json", "trend.json", "dense.json", "gappy.json"), case in golden(file)["cases"]      # gappy.json: F16 (tables with gaps: the warm loop's per-row test); dense.json: F14 (dense tables: the warm-started row loop); its six-planet F15 is skipped like every multi-planet case
    r = try run_case(case) catch err; @warn "case failed" case["name"] err; nothing end
    r === nothing || (out["$file/$(case["name"])"] = r)
end
open(joinpath(ROOT, "tests", "golden", "reference_dump.json"), "w") do io
    JSON.print(io, out, 1)
end
println("wrote tests/golden/reference_dump.json with ", length(out) - 3, " cases")
