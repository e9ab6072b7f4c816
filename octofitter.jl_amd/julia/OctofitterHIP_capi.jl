# OctofitterHIP_capi.jl — the THIN layer of the reference-side binding (SURVEY.md §7: "pure ccall"): the constants and structs of
# include/octofitter_hip.h and ONE `ccall` per exported symbol, nothing else. No Octofitter type appears here: a maintainer who wants
# only the C ABI from Julia includes this file into a module of their own; OctofitterHIP.jl includes it and builds the integration on top
# (`accelerate(system)`, `HIPLogDensityModel`, `GPUBatchedLikelihood`).
#
# NOT EXECUTED IN THE BUILD CONTAINER (Julia is not installed there). tests/test_abi.py compares every struct below with the header
# (field order, types, sizes through a compiled C program), requires a `ccall` for every symbol the header declares, with its arity.

const LIB = get(ENV, "OCTOFITTER_HIP_LIB", "liboctofitter_hip.so")

# ---------------------------------------------------------------------------------------------------- constants of the header
const OCTO_OK, OCTO_EINVAL, OCTO_EHIP, OCTO_ENOMEM, OCTO_ENODEV, OCTO_ENOTSUP = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
const ASTROM_RADEC, ASTROM_SEPPA, RV_ABS, RV_ABS_MARG, RV_REL = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const ONEIL_RADEC, ONEIL_SEPPA, HGCA = Int32(5), Int32(6), Int32(7)
const ORBIT_VISUAL_KEP, ORBIT_RADVEL, ORBIT_THIELE_INNES, ORBIT_KEP = Int32(0), Int32(1), Int32(2), Int32(3)
const PRIOR_UNIFORM, PRIOR_LOGUNIFORM, PRIOR_NORMAL, PRIOR_TRUNCNORMAL, PRIOR_SINE = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const SRC_CONST, SRC_THETA, SRC_CIRCULAR, SRC_TPERI = Int32(0), Int32(1), Int32(2), Int32(3)
const SRC_FLAG_UNITLEN, SRC_FLAG_TI = Int32(1), Int32(2)
const STREAM_CTX = Ptr{Cvoid}(typemax(UInt))       # OCTO_STREAM_CTX = (void*)-1: the context's own stream
const N_EL, N_NUIS = 9, 3
const OCTO_MAX_PLANETS = 8
const OCTO_MAX_PLANETS_ALL_KINDS = 4      # beyond it: the planet-per-wave throughput kernels only (every observation kind since round 6)
const EL_KEYS = (:a, :e, :i, :ω, :Ω, :tp, :M, :plx, :mass)
const EL_KEYS_TI = (:A, :e, :B, :F, :G, :tp, :M, :plx, :mass)     # ThieleInnesOrbit: constants [mas] in the rows of a, i, ω, Ω

# ---------------------------------------------------------------------------------------------------- structs of the header
struct OctoConsts            # mirrors `octo_consts`
    kepler_year_to_julian_day::Float64
    year2day_julian::Float64
    au2m::Float64
    sec2year_julian::Float64
    pc2au::Float64
    rad2as::Float64
    mjup2msol::Float64
end
struct OctoObsDesc           # mirrors `octo_obs_desc`
    kind::Int32
    planet::Int32
    n_epochs::Int64
    epoch::Ptr{Float64}
    y1::Ptr{Float64}
    y2::Ptr{Float64}
    s1::Ptr{Float64}
    s2::Ptr{Float64}
    cor::Ptr{Float64}
    extra::Ptr{Float64}
    n_extra::Int64
end
struct OctoPlanetDesc        # mirrors `octo_planet_desc`
    orbit_kind::Int32
    has_mass::Int32
end
struct OctoPrior             # mirrors `octo_prior`
    kind::Int32
    pad::Int32
    p0::Float64
    p1::Float64
    lo::Float64
    hi::Float64
end
struct OctoSource            # mirrors `octo_source`
    kind::Int32
    i0::Int32
    i1::Int32
    flags::Int32
    value::Float64
end

# ---------------------------------------------------------------------------------------------------- one ccall per exported symbol
"A non-zero status of the library (include/octofitter_hip.h: OCTO_EINVAL, OCTO_EHIP, OCTO_ENOMEM, OCTO_ENODEV, OCTO_ENOTSUP) as a Julia exception the shim can catch by kind."
struct OctoError <: Exception
    status::Int32
    what::String
    msg::String
end
# What the shim answers by staying on the reference's own path: no usable device, a valid system / model the device path does not take (OCTO_ENOTSUP), or
# the shim's own refusal of something valid in the reference (NotOnHIPPath: an orbit type, a prior, a Derived expression). Bad input (OCTO_EINVAL),
# OCTO_EHIP and OCTO_ENOMEM are rethrown.
struct NotOnHIPPath <: Exception; msg::String; end
Base.showerror(io::IO, e::NotOnHIPPath) = print(io, e.msg)
is_fallback(e) = e isa NotOnHIPPath || (e isa OctoError && (e.status == OCTO_ENODEV || e.status == OCTO_ENOTSUP))
Base.showerror(io::IO, e::OctoError) = print(io, "$(e.what) failed with status $(e.status): $(e.msg)")
check(ctx, st, what) = st == OCTO_OK ? nothing :
    throw(OctoError(st, what, ctx == C_NULL ? "" : unsafe_string(ccall((:octo_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))))

octo_version() = (a = Ref{Int32}(0); b = Ref{Int32}(0); ccall((:octo_version, LIB), Int32, (Ref{Int32}, Ref{Int32}), a, b); (a[], b[]))
octo_consts_default() = (c = Ref{OctoConsts}(); ccall((:octo_consts_default, LIB), Int32, (Ref{OctoConsts},), c); c[])
function octo_ctx_create(device::Integer)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:octo_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32), ctx, device)
    st == OCTO_OK || throw(OctoError(st, "octo_ctx_create", st == OCTO_ENODEV ? "no usable HIP device $device" : ""))
    return ctx[]
end
octo_ctx_destroy(ctx) = ccall((:octo_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), ctx)
octo_consts_set(ctx, c::OctoConsts) = check(ctx, ccall((:octo_consts_set, LIB), Int32, (Ptr{Cvoid}, Ref{OctoConsts}), ctx, c), "octo_consts_set")
octo_ctx_set_small_batch(ctx, n::Integer) = check(ctx, ccall((:octo_ctx_set_small_batch, LIB), Int32, (Ptr{Cvoid}, Int32), ctx, n), "octo_ctx_set_small_batch")
# Context options (include/octofitter_hip.h: OCTO_OPT_*). OCTO_OPT_BATCH_INVARIANT = 1 gives the reference's per-θ determinism (src/logdensitymodel.jl:110-146):
# ll(θ) bit-identical whatever batch θ is evaluated in.
const OCTO_OPT_BATCH_INVARIANT, OCTO_OPT_WARM_START, OCTO_OPT_TILE_SORT, OCTO_OPT_TILE_MIN_WALKERS = Int32(1), Int32(2), Int32(3), Int32(4)
octo_ctx_set_option(ctx, option::Integer, value::Integer) = check(ctx, ccall((:octo_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Int32, Int64), ctx, option, value), "octo_ctx_set_option")
octo_ctx_get_option(ctx, option::Integer) = (v = Ref{Int64}(0); check(ctx, ccall((:octo_ctx_get_option, LIB), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}), ctx, option, v), "octo_ctx_get_option"); v[])
# Page-lock and map an Array the caller keeps alive (elements, log-likelihoods, gradients of a big batch): later host-buffer calls
# whose buffers are all registered skip the copy engine. `GC.@preserve` the array for as long as it is registered; unregister before it is freed.
octo_host_register(ctx, a::Array{Float64}) = check(ctx, ccall((:octo_host_register, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), ctx, pointer(a), sizeof(a)), "octo_host_register")
octo_host_unregister(ctx, a::Array{Float64}) = check(ctx, ccall((:octo_host_unregister, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx, pointer(a)), "octo_host_unregister")
function octo_dataset_create(ctx, descs::Vector{OctoObsDesc}, planets::Vector{OctoPlanetDesc})
    ds = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:octo_dataset_create, LIB), Int32, (Ptr{Cvoid}, Ptr{OctoObsDesc}, Int32, Ptr{OctoPlanetDesc}, Int32, Ref{Ptr{Cvoid}}),
                     ctx, descs, length(descs), planets, length(planets), ds), "octo_dataset_create")
    return ds[]
end
octo_dataset_destroy(ds) = ccall((:octo_dataset_destroy, LIB), Int32, (Ptr{Cvoid},), ds)
octo_dataset_n_rows(ds) = ccall((:octo_dataset_n_rows, LIB), Int64, (Ptr{Cvoid},), ds)
octo_sync(ctx) = check(ctx, ccall((:octo_sync, LIB), Int32, (Ptr{Cvoid},), ctx), "octo_sync")

"Host buffers, blocking. `Xt`: W×inputs (walker index fastest); `G`: same shape or nothing (forward only)."
function octo_eval!(ctx, ds, Xt::Matrix{Float64}, n_el::Int, ll::Vector{Float64}, G::Union{Nothing,Matrix{Float64}}; with_nuis::Bool=true)
    W = size(Xt, 1); n_nu = with_nuis ? size(Xt, 2) - n_el : 0      # without nuisances the kernels take the precomputed-Σ⁻¹ path (jitter == 0, relative-astrometry.jl:218-219)
    pel = pointer(Xt); pnu = n_nu > 0 ? pointer(Xt, n_el * W + 1) : Ptr{Float64}(C_NULL)
    gel = G === nothing ? Ptr{Float64}(C_NULL) : pointer(G)
    gnu = (G === nothing || n_nu == 0) ? Ptr{Float64}(C_NULL) : pointer(G, n_el * W + 1)
    GC.@preserve Xt ll G check(ctx, ccall((:octo_eval, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx, ds, pel, pnu, W, W, ll, gel, gnu), "octo_eval")
    return ll
end
"The same call in two halves (one host thread, several devices): enqueue / wait. The arrays must stay rooted until `octo_eval_end`."
octo_eval_begin(ctx, ds, pel, pnu, ld, W, pll, pgel, pgnu) = check(ctx, ccall((:octo_eval_begin, LIB), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx, ds, pel, pnu, ld, W, pll, pgel, pgnu), "octo_eval_begin")
octo_eval_end(ctx) = check(ctx, ccall((:octo_eval_end, LIB), Int32, (Ptr{Cvoid},), ctx), "octo_eval_end")
"One host batch over several devices: ctxs[i], dss[i] live on device i (dataset replicated)."
octo_eval_multi(ctxs::Vector{Ptr{Cvoid}}, dss::Vector{Ptr{Cvoid}}, pel, pnu, ld, W, pll, pgel, pgnu) = check(ctxs[1], ccall((:octo_eval_multi, LIB), Int32,
    (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
    ctxs, dss, length(ctxs), pel, pnu, ld, W, pll, pgel, pgnu), "octo_eval_multi")
"Device-resident buffers (raw device pointers, e.g. from a HIP allocation the host owns), asynchronous on `stream`."
octo_eval_device(ctx, ds, d_el, d_nu, ld, W, d_ll, d_gel, d_gnu, stream=STREAM_CTX) = check(ctx, ccall((:octo_eval_device, LIB), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    ctx, ds, d_el, d_nu, ld, W, d_ll, d_gel, d_gnu, stream), "octo_eval_device")

"Batched PlanetOrbits.kepler_solver(MA, e) (src/parameterizations.jl:340) on the device."
function octo_kepler_solve(ctx, MA::Vector{Float64}, e::Vector{Float64})
    n = length(MA); E = similar(MA); sE = similar(MA); cE = similar(MA)
    check(ctx, ccall((:octo_kepler_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                     ctx, MA, e, n, E, sE, cE), "octo_kepler_solve")
    return E, sE, cE
end
"The same through the throughput kernels' variant of the routine (sin/cos of the starter from the LDS table)."
function octo_kepler_solve_table(ctx, MA::Vector{Float64}, e::Vector{Float64})
    n = length(MA); E = similar(MA); sE = similar(MA); cE = similar(MA)
    check(ctx, ccall((:octo_kepler_solve_table, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                     ctx, MA, e, n, E, sE, cE), "octo_kepler_solve_table")
    return E, sE, cE
end

# OFTI marginal likelihood (src/parameterizations.jl:318-405)
function octo_ofti_create(ctx, epochs, ra, dec, σ_ra, σ_dec, cor, σ_ABFG)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    c = cor === nothing ? Ptr{Float64}(C_NULL) : pointer(cor)
    GC.@preserve cor check(ctx, ccall((:octo_ofti_create, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ref{Ptr{Cvoid}}),
        ctx, epochs, ra, dec, σ_ra, σ_dec, c, length(epochs), σ_ABFG, h), "octo_ofti_create")
    return h[]
end
octo_ofti_destroy(h) = ccall((:octo_ofti_destroy, LIB), Int32, (Ptr{Cvoid},), h)
"nl: W×5 columns e, a, tp, M, plx. Returns (ABFG W×4, log_marginal_likelihood W)."
function octo_ofti_eval(ctx, h, nl::Matrix{Float64})
    W = size(nl, 1); abfg = Matrix{Float64}(undef, W, 4); lm = Vector{Float64}(undef, W)
    check(ctx, ccall((:octo_ofti_eval, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
                     ctx, h, nl, W, W, abfg, lm), "octo_ofti_eval")
    return abfg, lm
end
octo_ofti_eval_device(ctx, h, d_nl, ld, W, d_abfg, d_lm, stream=STREAM_CTX) = check(ctx, ccall((:octo_ofti_eval_device, LIB), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), ctx, h, d_nl, ld, W, d_abfg, d_lm, stream), "octo_ofti_eval_device")

# standard parameterisation on the device
function octo_model_create(ctx, ds, priors::Vector{OctoPrior}, esrc::Vector{OctoSource}, nsrc::Union{Nothing,Vector{OctoSource}})
    m = Ref{Ptr{Cvoid}}(C_NULL)
    pn = nsrc === nothing ? Ptr{OctoSource}(C_NULL) : pointer(nsrc)
    GC.@preserve nsrc check(ctx, ccall((:octo_model_create, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{OctoPrior}, Int32, Ptr{OctoSource}, Ptr{OctoSource}, Ref{Ptr{Cvoid}}),
        ctx, ds, priors, length(priors), esrc, pn, m), "octo_model_create")
    return m[]
end
octo_model_destroy(m) = ccall((:octo_model_destroy, LIB), Int32, (Ptr{Cvoid},), m)
"Θt: W×D (walker index fastest). Returns lp (and fills G, W×D, when given)."
function octo_model_logpost!(ctx, m, Θt::Matrix{Float64}, lp::Vector{Float64}, G::Union{Nothing,Matrix{Float64}})
    W = size(Θt, 1)
    pg = G === nothing ? Ptr{Float64}(C_NULL) : pointer(G)
    GC.@preserve G check(ctx, ccall((:octo_model_logpost, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
                                    ctx, m, Θt, W, W, lp, pg), "octo_model_logpost")
    return lp
end
octo_model_logpost_device(ctx, m, d_θt, ld, W, d_lp, d_grad, stream=STREAM_CTX) = check(ctx, ccall((:octo_model_logpost_device, LIB), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), ctx, m, d_θt, ld, W, d_lp, d_grad, stream), "octo_model_logpost_device")

# measurement hooks
octo_timing_enable(ctx, every_n::Integer) = ccall((:octo_timing_enable, LIB), Int32, (Ptr{Cvoid}, Int32), ctx, every_n)
function octo_timing_read(ctx; reset::Bool=true)
    ms = Ref{Float64}(0); n = Ref{Int64}(0)
    ccall((:octo_timing_read, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}, Int32), ctx, ms, n, reset)
    return ms[], n[]
end
function octo_timing_stats(ctx)
    med = Ref{Float64}(0); lo = Ref{Float64}(0); hi = Ref{Float64}(0); n = Ref{Int64}(0)
    ccall((:octo_timing_stats, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}), ctx, med, lo, hi, n)
    return med[], lo[], hi[], n[]
end

# parallel tempering (BASELINE config 5): one process per GPU, one all-gather per swap step inside the library
"Rank 0: the 128-byte RCCL rendezvous id; hand it to the other ranks (MPI.Bcast!, a file, …)."
octo_comm_unique_id() = (id = zeros(UInt8, 128); st = ccall((:octo_comm_unique_id, LIB), Int32, (Ptr{UInt8},), id); st == OCTO_OK || throw(OctoError(st, "octo_comm_unique_id", "")); id)
octo_comm_create(ctx, id::Union{Nothing,Vector{UInt8}}, rank::Integer, world::Integer) = check(ctx, ccall((:octo_comm_create, LIB), Int32,
    (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), ctx, id === nothing ? Ptr{UInt8}(C_NULL) : pointer(id), rank, world), "octo_comm_create")
octo_comm_destroy(ctx) = ccall((:octo_comm_destroy, LIB), Int32, (Ptr{Cvoid},), ctx)
"ncclAllGather of the local replicas' log-likelihoods + deterministic neighbour swap of β labels, on `stream` (device pointers)."
octo_pt_step_device(ctx, d_ll_local, d_ll_all, d_beta, d_slot2rep, n_temps, n_chains, parity, seed, step, d_accepted, stream=STREAM_CTX) =
    check(ctx, ccall((:octo_pt_step_device, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Int32, Int64, Int32, UInt64, UInt64, Ptr{Int32}, Ptr{Cvoid}),
        ctx, d_ll_local, d_ll_all, d_beta, d_slot2rep, n_temps, n_chains, parity, seed, step, d_accepted, stream), "octo_pt_step_device")
octo_pt_step(ctx, ll_local::Vector{Float64}, beta::Vector{Float64}, slot2rep::Matrix{Int32}, n_temps, n_chains, parity, seed, step, accepted::Vector{Int32}) =
    check(ctx, ccall((:octo_pt_step, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Int32, Int64, Int32, UInt64, UInt64, Ptr{Int32}),
        ctx, ll_local, beta, slot2rep, n_temps, n_chains, parity, seed, step, accepted), "octo_pt_step")
octo_pt_swap_device(ctx, d_ll_by_replica, d_beta, d_slot2rep, n_temps, n_chains, parity, seed, step, d_accepted, stream=STREAM_CTX) =
    check(ctx, ccall((:octo_pt_swap_device, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Int32, Int64, Int32, UInt64, UInt64, Ptr{Int32}, Ptr{Cvoid}),
        ctx, d_ll_by_replica, d_beta, d_slot2rep, n_temps, n_chains, parity, seed, step, d_accepted, stream), "octo_pt_swap_device")

