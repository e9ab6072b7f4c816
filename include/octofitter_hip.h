/*
 * octofitter_hip.h — C ABI of the MI355X (gfx950) hot path for Octofitter.jl.
 *
 * One call evaluates, for a batch of W walkers, the epoch-vectorised
 *   Kepler solve -> sky-plane projection (RA/Dec, sep/PA, RV) -> Gaussian
 *   log-likelihood reduction over epochs and companions,
 * and (optionally) its reverse-mode gradient w.r.t. the orbital elements and
 * the per-observation nuisance parameters.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an
 * int32 status (0 = OCTO_OK) and never lets a C++ exception cross the boundary.
 * It is what a Julia `ccall` / Python `ctypes` binding for this path binds to
 * (see INTEGRATION.md for the reference-side stub).
 *
 * Reference interfaces each symbol stands in for (paths under the reference
 * tree of sefffal/Octofitter.jl v8.3.0):
 *
 *   octo_dataset_create   <- the observation tables captured by make_ln_like:
 *                            src/likelihoods/system.jl:35-54 (epoch gather),
 *                            src/likelihoods/relative-astrometry.jl:20-95 (table,
 *                            sorted by epoch :46-47, optional cor :67-81),
 *                            OctofitterRadialVelocity/src/rv-absolute.jl:56-113,
 *                            rv-absolute-margin.jl:60-84, rv-relative.jl:60-101;
 *                            src/likelihoods/prior-observable.jl:56-76 (O'Neil wrapper).
 *   octo_consts_set       <- PlanetOrbits.* physical constants used through
 *                            src/parameterizations.jl:62-64,215-216 and
 *                            src/Octofitter.jl:43 (mjup2msol).
 *   octo_eval / _device   <- make_ln_like's generated body
 *                            src/likelihoods/system.jl:206-241 applied to W
 *                            parameter sets at once: orbit constructors (:116-118),
 *                            _kepsolve_all! (:250-269), then every observation's
 *                            ln_like (relative-astrometry.jl:166-253,
 *                            rv-absolute.jl:172-204, rv-absolute-margin.jl:140-185,
 *                            rv-relative.jl:177-211), i.e. the body of
 *                            ℓπcallback/∇ℓπcallback below the prior
 *                            (src/logdensitymodel.jl:134, :169-177).
 *   per-walker -Inf       <- src/logdensitymodel.jl:120-124, system.jl:214-221.
 */
#ifndef OCTOFITTER_HIP_H
#define OCTOFITTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCTO_VERSION_MAJOR 0
#define OCTO_VERSION_MINOR 1

/* ---- status codes ------------------------------------------------------- */
#define OCTO_OK      0
#define OCTO_EINVAL  1   /* bad argument (null pointer, bad size, unsupported combination) */
#define OCTO_EHIP    2   /* a HIP runtime call failed; see octo_last_error */
#define OCTO_ENOMEM  3   /* host or device allocation failed */
#define OCTO_ENODEV  4   /* no usable gfx950 device */
#define OCTO_ENOTSUP 5   /* a VALID dataset or model that is not on the device path: more planets than OCTO_MAX_PLANETS, (nothing of the observation kinds since round 6), an RV table next to a ThieleInnesOrbit planet, a model beyond the size
                          * limits of octo_model_create. A host-side binding falls back to the reference's own path on THIS status (and on OCTO_ENODEV) —
                          * not on OCTO_EINVAL (bad input: σ <= 0, non-finite epochs, |cor| >= 1 …), OCTO_EHIP or OCTO_ENOMEM, which are the caller's to see. */

/* ---- observation kinds (one per reference AbstractObs type on the path) -- */
#define OCTO_ASTROM_RADEC 0  /* PlanetRelAstromObs with (ra, dec, σ_ra, σ_dec[, cor])   */
#define OCTO_ASTROM_SEPPA 1  /* PlanetRelAstromObs with (pa, sep, σ_pa, σ_sep[, cor])   */
#define OCTO_RV_ABS       2  /* StarAbsoluteRVObs (no GP; trend: see OCTO_NU_RV_TREND)   */
#define OCTO_RV_ABS_MARG  3  /* MarginalizedStarAbsoluteRVObs (trend likewise)           */
#define OCTO_RV_REL       4  /* PlanetRelativeRVObs (no GP; trend likewise)              */
#define OCTO_ONEIL_RADEC  5  /* ObsPriorAstromONeil2019 wrapping an (ra, dec) table: the wrapped ln_like PLUS the    */
#define OCTO_ONEIL_SEPPA  6  /*   observable-based prior 2 log(Σ_j |…|·∛P/√(1−e²)), src/likelihoods/prior-observable.jl:78-137 */
#define OCTO_HGCA         7  /* HGCAInstantaneousObs (src/likelihoods/hgca.jl:28-219): Hipparcos-Gaia proper-motion   */
                             /*   anomaly of the primary from the companions' reflex positions and velocities at a    */
                             /*   handful of epochs. System table (planet = -1). Rows: epoch, y1 = measured axis      */
                             /*   (OCTO_HGCA_RA / _DEC), y2 = mission (OCTO_HGCA_HIP / _GAIA); s1, s2, cor NULL.      */
                             /*   `extra` = the OCTO_HGCA_N_EXTRA catalogue numbers below. Its two per-walker          */
                             /*   "nuisances" are the system's proper motion: slot 0 = pmra, slot 1 = pmdec [mas/yr]   */
                             /*   (θ_system.pmra / .pmdec, hgca.jl:266-267), so `nuis` is required with this kind.     */
#define OCTO_N_KINDS      8

#define OCTO_HGCA_RA   0
#define OCTO_HGCA_DEC  1
#define OCTO_HGCA_HIP  0
#define OCTO_HGCA_GAIA 1
/* extra[] of an OCTO_HGCA table: three epochs' (pmra, pmdec, σ_pmra·factor, σ_pmdec·factor, correlation), mas/yr:
 * Hipparcos [0..4], Hipparcos-Gaia scaled position difference [5..9], Gaia [10..14]   (hgca.jl:127-145, 182-203) */
#define OCTO_HGCA_N_EXTRA 15
#define OCTO_NU_HGCA_PMRA  0
#define OCTO_NU_HGCA_PMDEC 1

/* ---- orbit parameterisations (PlanetOrbits.jl types) --------------------- */
#define OCTO_ORBIT_VISUAL_KEP 0  /* Visual{KepOrbit}: a,e,i,ω,Ω,tp,M,plx               */
#define OCTO_ORBIT_RADVEL     1  /* RadialVelocityOrbit: a,e,ω,tp,M (i,Ω,plx ignored)  */
#define OCTO_ORBIT_KEP        3  /* plain KepOrbit: a,e,i,ω,Ω,tp,M — no parallax, so RV tables only (K carries sin i)  */
#define OCTO_ORBIT_THIELE_INNES 2 /* ThieleInnesOrbit: e,tp,M,plx and the Thiele-Innes constants A,B,F,G [mas] in the element rows
                                   * OCTO_EL_TI_A/B/F/G below; a = α/plx with α from A,B,F,G (src/parameterizations.jl:15-19).
                                   * Astrometry, the O'Neil prior and HGCA only (no RV tables with this basis). */

/* ---- element rows: elems[(planet*OCTO_N_EL + k) * ld + w] ---------------- */
#define OCTO_EL_A     0   /* semi-major axis [AU]                 */
#define OCTO_EL_E     1   /* eccentricity, 0 <= e < 1             */
#define OCTO_EL_I     2   /* inclination [rad]                    */
#define OCTO_EL_W     3   /* argument of periastron ω [rad]       */
#define OCTO_EL_O     4   /* longitude of ascending node Ω [rad]  */
#define OCTO_EL_TP    5   /* epoch of periastron passage [MJD]    */
#define OCTO_EL_M     6   /* total mass [M_sun]                   */
#define OCTO_EL_PLX   7   /* parallax [mas]                       */
#define OCTO_EL_MASS  8   /* companion mass [M_jup]               */
#define OCTO_N_EL     9
/* planets per dataset. 1 … OCTO_MAX_PLANETS_ALL_KINDS: every observation kind, both kernel families. Up to OCTO_MAX_PLANETS: the
 * planet-per-wave throughput kernels (one planet per wave of a block, any batch size) — every observation kind since round 6; no small-batch kernels beyond
 * OCTO_MAX_PLANETS_ALL_KINDS (a one-θ call takes the throughput launches). octo_dataset_create refuses more planets than OCTO_MAX_PLANETS with OCTO_ENOTSUP. (The
 * reference unrolls over any number, src/likelihoods/system.jl:116-118: a host-side binding keeps a refused system on its own path.) */
#define OCTO_MAX_PLANETS 8
#define OCTO_MAX_PLANETS_ALL_KINDS 4
/* the same rows for an OCTO_ORBIT_THIELE_INNES planet: A, B, F, G replace a, i, ω, Ω */
#define OCTO_EL_TI_A  0
#define OCTO_EL_TI_B  2
#define OCTO_EL_TI_F  3
#define OCTO_EL_TI_G  4

/* ---- nuisance rows: nuis[(obs*OCTO_N_NUIS + k) * ld + w] ----------------- */
/* astrometry kinds */
#define OCTO_NU_JITTER     0   /* added in quadrature to both σ (relative-astrometry.jl:234-235) */
#define OCTO_NU_PLATESCALE 1   /* multiplies the data separation (:202,:211)                    */
#define OCTO_NU_NORTHANGLE 2   /* rotates the data (:198,:210)                                  */
/* RV kinds */
#define OCTO_NU_RV_OFFSET  0   /* rv-absolute.jl:139, rv-relative.jl:130 (ignored by RV_ABS_MARG) */
#define OCTO_NU_RV_JITTER  1   /* rv-absolute.jl:181,197; rv-absolute-margin.jl:149,174           */
#define OCTO_NU_RV_TREND   2   /* coefficient c of the table's trend: rv_model[j] += c · basis[j], the reference's
                                * `trend_function(θ_obs, epoch_j)` (rv-absolute.jl:69,143; rv-relative.jl:64,131;
                                * rv-absolute-margin.jl:52,111) for every trend that is LINEAR IN ONE θ_obs variable:
                                * the host evaluates the user's closure once per table row with that variable set to 1 and
                                * uploads the column as the table's `extra` (basis[j] = epoch_j − 57000 for the documented
                                * `θ_obs.trend_slope * (epoch - 57000)`, rv-absolute.jl:26). Ignored (and its gradient 0) when
                                * the table was uploaded without a basis column. Any other trend closure is not on this path. */
#define OCTO_N_NUIS        3

/* Physical constants, supplied by the host from PlanetOrbits.* so that parity
 * is a property of formulas, not of digits baked into a kernel. */
typedef struct octo_consts {
    double kepler_year_to_julian_day; /* PlanetOrbits.kepler_year_to_julian_day_conversion_factor */
    double year2day_julian;           /* PlanetOrbits.year2day_julian  (365.25)                   */
    double au2m;                      /* PlanetOrbits.au2m                                         */
    double sec2year_julian;           /* PlanetOrbits.sec2year_julian                              */
    double pc2au;                     /* PlanetOrbits.pc2au                                        */
    double rad2as;                    /* PlanetOrbits.rad2as                                       */
    double mjup2msol;                 /* Octofitter.mjup2msol = PlanetOrbits.mjup2msol_IAU         */
} octo_consts;

/* One observation table, post-constructor (sorted by epoch), host pointers.
 * The library copies the columns; the caller keeps ownership. */
typedef struct octo_obs_desc {
    int32_t kind;          /* OCTO_ASTROM_* / OCTO_RV_*                                     */
    int32_t planet;        /* 0-based planet the table is attached to; -1 for system tables */
    int64_t n_epochs;      /* rows; may be 0                                                */
    const double* epoch;   /* [n_epochs] MJD                                                */
    const double* y1;      /* ra | pa | rv                                                  */
    const double* y2;      /* dec | sep | NULL                                              */
    const double* s1;      /* σ_ra | σ_pa | σ_rv                                            */
    const double* s2;      /* σ_dec | σ_sep | NULL                                          */
    const double* cor;     /* correlation column or NULL (astrometry only)                  */
    const double* extra;   /* per-table constants (OCTO_HGCA: [OCTO_HGCA_N_EXTRA]; RV kinds: the trend basis  */
    int64_t n_extra;       /*   column [n_epochs], see OCTO_NU_RV_TREND) or NULL; n_extra = its length          */
} octo_obs_desc;

typedef struct octo_planet_desc {
    int32_t orbit_kind;    /* OCTO_ORBIT_*                                                  */
    int32_t has_mass;      /* planet declares a `mass` variable (relative-astrometry.jl:122) */
} octo_planet_desc;

typedef struct octo_ctx octo_ctx;
typedef struct octo_dataset octo_dataset;

/* Defaults: the PlanetOrbits.jl / Octofitter.jl constant values this build assumes
 * when the host does not call octo_consts_set. */
int32_t octo_consts_default(octo_consts* out);

int32_t octo_version(int32_t* major, int32_t* minor);

/* Context = one HIP device + one stream + scratch + the row-partition cache. Not thread-safe: one ctx per host thread
 * AND per stream of concurrent work — two evaluations through one context share its scratch, so they are ordered one after
 * the other (if the caller switches streams between calls the library inserts the event wait itself; it never overlaps
 * them). A dataset is immutable after octo_dataset_create and may be shared, without locking, by any number of contexts
 * and host threads on the same device. */
int32_t octo_ctx_create(octo_ctx** out, int32_t device_id);
int32_t octo_ctx_destroy(octo_ctx* ctx);
int32_t octo_consts_set(octo_ctx* ctx, const octo_consts* c);
/* Small and mid-size batches take the fused single-launch kernel: the EPOCHS across the lanes, a wavefront/LDS tree
 * reduction, one block (or a few) per parameter set — the latency path for samplers that evaluate one θ per call
 * (src/logdensitymodel.jl:169-177), Pigeons' replicas and an ensemble sampler's 10²-10³ walkers. A batch of W parameter sets
 * of a P-planet system takes it when W·P <= max_walkers (default OCTO_SMALL_BATCH_DEFAULT, at most OCTO_SMALL_BATCH_MAX:
 * the measured crossover with the throughput kernels, lane = walker, whose fixed cost is three launches). 0 sends every batch
 * through the throughput kernels. Host-buffer calls of up to 128 sets use mapped pinned memory (no copy engine), larger ones up
 * to 1 MiB one pinned DMA each way. Results of the two kernel families agree to rounding, not bitwise (different summation order). */
#define OCTO_SMALL_BATCH_MAX 1024
#define OCTO_SMALL_BATCH_DEFAULT 512
int32_t octo_ctx_set_small_batch(octo_ctx* ctx, int32_t max_walkers);
const char* octo_last_error(const octo_ctx* ctx);

/* Context options (round 6). The reference evaluates ONE θ at a time and its result is a function of θ alone (src/logdensitymodel.jl:110-146).
 * The batched throughput kernels keep that to rounding (~1e-15), not bitwise, by default: on tables dense enough for the warm-started row loop a
 * wave falls back to the cold Kepler starter as a whole, so the last bits of a walker's result depend on the 63 walkers it shares a wave with; the
 * row partition follows the batch size; and big single-planet batches may be re-tiled by a severity key (below). Results ARE bit-reproducible from
 * run to run for the same sequence of calls in every mode.
 *   OCTO_OPT_BATCH_INVARIANT  1: ll(θ) and its gradient are independent of the batch's size and composition, bit for bit — checkpoint / resume, a
 *                             1-GPU against an 8-GPU rerun of one chain, two batch sizes: the cold row loop, no tile sort, ONE row partition (64 rows
 *                             per wave) and the throughput kernels for every batch size (no small-batch route: a one-θ call then costs three launches,
 *                             ~40 µs instead of ~16). Default 0. Costs ~25 % of the throughput on dense tables (round 4's rate).
 *   OCTO_OPT_WARM_START       0: the cold row loop only (round 4's kernels), everything else as usual. Default 1 (environment OCTO_WARM=0 at
 *                             octo_ctx_create sets 0).
 *   OCTO_OPT_TILE_SORT        walkers of big single-planet batches grouped into tiles of 64 by how often their rows would fail the warm start's
 *                             a-priori test (period against the table's cadence, eccentricity): 0 never, 1 every eligible evaluation, 2 (default)
 *                             when a probe — every 64th eligible evaluation of a (dataset, batch size) — estimates that it saves more than the sort
 *                             launch costs. Inputs and outputs keep the caller's order; only the last bits of the results may differ (see above).
 *   OCTO_OPT_TILE_MIN_WALKERS batches below this size are never sorted (default 2048).
 * octo_ctx_set_option: OCTO_EINVAL for an unknown option or a value outside its range. */
#define OCTO_OPT_BATCH_INVARIANT  1
#define OCTO_OPT_WARM_START       2
#define OCTO_OPT_TILE_SORT        3
#define OCTO_OPT_TILE_MIN_WALKERS 4
int32_t octo_ctx_set_option(octo_ctx* ctx, int32_t option, int64_t value);
int32_t octo_ctx_get_option(const octo_ctx* ctx, int32_t option, int64_t* value_out);

/* Upload the observation tables (one-time). Observation likelihoods are summed
 * in the order given. OCTO_EINVAL for a non-finite epoch or measurement, an uncertainty that is not finite and > 0, or a
 * correlation outside the reference's bound (relative-astrometry.jl:70-72). */
int32_t octo_dataset_create(octo_ctx* ctx,
                            const octo_obs_desc* obs, int32_t n_obs,
                            const octo_planet_desc* planets, int32_t n_planets,
                            octo_dataset** out);
int32_t octo_dataset_destroy(octo_dataset* ds);
int64_t octo_dataset_n_rows(const octo_dataset* ds);   /* Σ n_epochs over tables */

/* Batched evaluation, HOST buffers (blocking).
 *   elems   [n_planets*OCTO_N_EL][ld]  walker index fastest (SoA)
 *   nuis    [n_obs*OCTO_N_NUIS][ld] or NULL (defaults: jitter 0, platescale 1,
 *           northangle 0, offset 0)
 *   ll_out  [W]
 *   g_elems [n_planets*OCTO_N_EL][ld] or NULL  -> forward only
 *   g_nuis  [n_obs*OCTO_N_NUIS][ld]  or NULL
 * Per-walker numerical invalidity is data, not an error: ll = -Inf and zero
 * gradients for NaN/Inf inputs, e outside [0,1), a <= 0, M <= 0. */
int32_t octo_eval(octo_ctx* ctx, const octo_dataset* ds,
                  const double* elems, const double* nuis, int64_t ld, int64_t W,
                  double* ll_out, double* g_elems, double* g_nuis);

/* The same call in two halves, for one host thread that drives several devices: octo_eval_begin enqueues the
 * copies and kernels on the context's stream and returns; octo_eval_end waits for them and finishes the copy-out. One
 * begin may be outstanding per context; the host buffers must stay valid and untouched until the matching end. */
int32_t octo_eval_begin(octo_ctx* ctx, const octo_dataset* ds,
                        const double* elems, const double* nuis, int64_t ld, int64_t W,
                        double* ll_out, double* g_elems, double* g_nuis);
int32_t octo_eval_end(octo_ctx* ctx);
/* Between octo_eval_begin and octo_eval_end the context's other HOST-buffer entry points (octo_eval, octo_model_logpost,
 * octo_ofti_eval, octo_kepler_solve*) return OCTO_EINVAL: they share its staging buffers. */

/* One host batch over n_dev devices from ONE host thread: walkers are split contiguously and evenly (device i gets
 * [i·W/n, (i+1)·W/n) up to rounding), every device's work is enqueued before any is waited for, and the outputs land in
 * the caller's arrays at the walkers' own columns. ctxs[i] and datasets[i] live on device i (the dataset is replicated:
 * create it once per context from the same tables). No communication: walkers are independent (system.jl:206-241). */
int32_t octo_eval_multi(octo_ctx* const* ctxs, const octo_dataset* const* datasets, int32_t n_dev,
                        const double* elems, const double* nuis, int64_t ld, int64_t W,
                        double* ll_out, double* g_elems, double* g_nuis);

/* `hip_stream` of every *_device entry point: a hipStream_t, handed to HIP as it is — so NULL is HIP's NULL (legacy
 * default) stream, exactly what a framework that has no stream of its own selected reports as its current stream — or
 * OCTO_STREAM_CTX for the context's own non-blocking stream (the one octo_sync waits for and every host-buffer entry
 * point uses). The kernels are ordered like any other work on that stream: after what the caller enqueued before the
 * call, before what it enqueues after it. */
#define OCTO_STREAM_CTX ((void*)(intptr_t)-1)

/* Same, DEVICE buffers already resident in HBM, enqueued on `hip_stream`. Asynchronous: synchronise that stream (or, for
 * OCTO_STREAM_CTX, call octo_sync) before reading outputs. */
int32_t octo_eval_device(octo_ctx* ctx, const octo_dataset* ds,
                         const double* d_elems, const double* d_nuis, int64_t ld, int64_t W,
                         double* d_ll_out, double* d_g_elems, double* d_g_nuis,
                         void* hip_stream);
int32_t octo_sync(octo_ctx* ctx);

/* Pinning the caller's arrays (optional). octo_eval / octo_eval_begin with pageable host buffers stage big batches through
 * the runtime's pageable copies (~120 µs per call for the 1.4 MB of a 1e4-walker gradient call). A host that keeps its
 * element / log-likelihood / gradient arrays alive across calls — a sampler's preallocated buffers; in Julia the Arrays the
 * shim passes to ccall — registers them ONCE: the range is page-locked and mapped into the device's address space
 * (hipHostRegister), and every later host-buffer call whose buffers ALL lie inside registered ranges skips the copy engine:
 * one copy kernel reads the inputs over PCIe, the kernels write ll and the gradients straight into the caller's arrays.
 * Results are bit-identical to the pageable path. Unregister before freeing the memory. The registry is process-wide and
 * thread-safe; a range is usable by contexts on the device it was registered for. octo_host_unregister waits for ALL work on that
 * device (any context, any stream) and then unmaps the range; no evaluation that uses the range may be started concurrently
 * with it. Mirrors nothing in the reference (its
 * arrays never leave the host): it is the boundary's cost model, next to `octo_eval`
 * (src/likelihoods/system.jl:206-241 is the call it replaces). */
int32_t octo_host_register(octo_ctx* ctx, void* ptr, int64_t bytes);
int32_t octo_host_unregister(octo_ctx* ctx, void* ptr);

/* Batched eccentric-anomaly solve, HOST buffers (blocking): E = kepler_solver(MA, e) for 0 <= e < 1, the
 * call the reference makes at src/parameterizations.jl:340 (PlanetOrbits.kepler_solver, Markley). Runs the device
 * routine of the small-batch likelihood kernel (half-angle polynomial sin/cos of the starter). sinE_out / cosE_out may be NULL. Invalid inputs give NaN. */
int32_t octo_kepler_solve(octo_ctx* ctx, const double* MA, const double* e, int64_t n,
                          double* E_out, double* sinE_out, double* cosE_out);
/* The same call through the THROUGHPUT kernels' variant of the routine (k_main, k_ofti_main): sin/cos of the FP32 starter from the
 * 1041-entry table in LDS + a rotation, instead of the half-angle polynomials of the small-batch kernel that octo_kepler_solve runs.
 * Both variants find the same root to rounding; the parity tests check each over the whole elliptic domain. */
int32_t octo_kepler_solve_table(octo_ctx* ctx, const double* MA, const double* e, int64_t n,
                                double* E_out, double* sinE_out, double* cosE_out);

/* OFTI marginal likelihood: batched `ofti_linear_solve(epochs, ra, dec, σ_ra, σ_dec, cor, σ_ABFG,
 * e, a, tp, M, plx)` — src/parameterizations.jl:318-405. The handle holds one RA/Dec table (cor may be NULL = 0) and
 * σ_ABFG; an evaluation takes the nonlinear parameters nl[(k)*ld + w], k ∈ {e, a, tp, M, plx}, and returns the
 * posterior-mean Thiele-Innes constants abfg[(k)*ld + w], k ∈ {A, B, F, G} (may be NULL) and log_marginal_likelihood[w].
 * Invalid walkers (non-finite input, e ∉ [0,1), a <= 0, M <= 0): log-likelihood -Inf, constants NaN. */
typedef struct octo_ofti octo_ofti;
int32_t octo_ofti_create(octo_ctx* ctx, const double* epochs, const double* ra, const double* dec,
                         const double* sigma_ra, const double* sigma_dec, const double* cor, int64_t n_epochs,
                         double sigma_abfg, octo_ofti** out);
int32_t octo_ofti_destroy(octo_ofti* h);
int32_t octo_ofti_eval(octo_ctx* ctx, const octo_ofti* h, const double* nl, int64_t ld, int64_t W,
                       double* abfg_out, double* logml_out);
int32_t octo_ofti_eval_device(octo_ctx* ctx, const octo_ofti* h, const double* d_nl, int64_t ld, int64_t W,
                              double* d_abfg_out, double* d_logml_out, void* hip_stream);

/* ---- Standard parameterisation on the device ------------------------------------------------
 * The whole log-posterior callback ℓπcallback / ∇ℓπcallback (src/logdensitymodel.jl:110-146, 169-177) for models
 * whose variables are the reference's standard building blocks, so that θ_t never leaves HBM:
 *   invlink of each prior (Bijectors: src/variables.jl:1449-1493), logpdf_with_trans of each prior in declaration
 *   order (src/variables.jl:1205-1369), UniformCircular angles and their UnitLengthPrior terms (:279-323),
 *   tp = θ_at_epoch_to_tperi(θ, epoch; M, e, a, i, ω, Ω) (src/parameterizations.jl:6-69), then the likelihood above.
 * A model is: D priors (one per θ_t entry, in the reference's flattening order) and, for every kernel input (element
 * row of each planet, nuisance row of each observation), a SOURCE saying how it is built from the natural θ.
 * Size limit: 1 <= D <= 64 and (4·D + 6·n_circular)·512 B of LDS per block must fit the device (160 KB on gfx950:
 * i.e. 4·D + 6·n_circular <= 320: D = 64 with up to 10 UniformCircular variables); octo_model_create returns OCTO_EINVAL beyond it. */
#define OCTO_PRIOR_UNIFORM     0   /* Uniform(p0, p1)                                                   */
#define OCTO_PRIOR_LOGUNIFORM  1   /* LogUniform(p0, p1)                                                */
#define OCTO_PRIOR_NORMAL      2   /* Normal(p0 = μ, p1 = σ)                                            */
#define OCTO_PRIOR_TRUNCNORMAL 3   /* truncated(Normal(p0, p1), lower = lo, upper = hi); ±INFINITY = open */
#define OCTO_PRIOR_SINE        4   /* Octofitter.Sine()  (src/distributions.jl:14-39)                   */
typedef struct octo_prior {
    int32_t kind, pad;
    double p0, p1, lo, hi;
} octo_prior;

#define OCTO_SRC_CONST    0   /* value                                                                       */
#define OCTO_SRC_THETA    1   /* θ[i0] (natural domain)                                                      */
#define OCTO_SRC_CIRCULAR 2   /* atan(θ[i1], θ[i0]) / 2π · value      — UniformCircular(value), + UnitLengthPrior */
#define OCTO_SRC_TPERI    3   /* θ_at_epoch_to_tperi(atan(θ[i1], θ[i0]), value; M, e, a, i, ω, Ω) of this planet  */
#define OCTO_SRC_FLAG_UNITLEN 1   /* this use of the (i0, i1) pair also contributes its UnitLengthPrior term (set it on exactly
                                     one source per UniformCircular variable; variables.jl:309-323) */
#define OCTO_SRC_FLAG_TI      2   /* OCTO_SRC_TPERI of a Thiele-Innes planet: θ_at_epoch_to_tperi(θ, value; plx, M, e, A, B, F, G) */
typedef struct octo_source {
    int32_t kind, i0, i1, flags;
    double value;
} octo_source;

typedef struct octo_model octo_model;
int32_t octo_model_create(octo_ctx* ctx, const octo_dataset* ds, const octo_prior* priors, int32_t D,
                          const octo_source* elem_src /* [n_planets*OCTO_N_EL] */,
                          const octo_source* nuis_src /* [n_obs*OCTO_N_NUIS] or NULL = defaults */,
                          octo_model** out);
int32_t octo_model_destroy(octo_model* m);
/* theta_t[(d)*ld + w], d < D: unconstrained parameters. lp_out[W]; grad_out[(d)*ld + w] or NULL. HOST buffers. */
int32_t octo_model_logpost(octo_ctx* ctx, octo_model* m, const double* theta_t, int64_t ld, int64_t W,
                           double* lp_out, double* grad_out);
/* DEVICE buffers, asynchronous on hip_stream (see OCTO_STREAM_CTX). */
int32_t octo_model_logpost_device(octo_ctx* ctx, octo_model* m, const double* d_theta_t, int64_t ld, int64_t W,
                                  double* d_lp_out, double* d_grad_out, void* hip_stream);

/* Measurement hook used by bench.py: average duration in milliseconds of the
 * dominant (epoch-loop) kernel over the launches since the last reset, from
 * hipEvents recorded on the launch stream. octo_timing_enable(ctx, n): n = 0 off, n >= 1 bracket the kernel of every
 * n-th evaluation (an event pair costs a few µs of stream time, so a timed region samples rather than brackets all);
 * n = -1: bracket every HOST-BUFFER evaluation (octo_eval / octo_eval_begin) whole, from ahead of its first copy to behind
 * its last kernel or copy — the device time of the call with the PCIe transfers inside (SURVEY.md 8d's clock). */
int32_t octo_timing_enable(octo_ctx* ctx, int32_t every_n);
int32_t octo_timing_read(octo_ctx* ctx, double* avg_ms, int64_t* n_launches, int32_t reset);
/* Median / min / max over the individual timed launches since the last reset (does not reset). */
int32_t octo_timing_stats(octo_ctx* ctx, double* median_ms, double* min_ms, double* max_ms, int64_t* n_launches);

/* Parallel-tempering swap step on device-resident log-likelihoods gathered from
 * all replicas (the one collective of the path; the gather itself is RCCL's
 * all_gather issued by the host). Deterministic even/odd neighbour swaps driven
 * by a counter-based RNG (seed, step) shared by all ranks:
 *   log A = (β_i − β_{i+1}) (ℓ_{i+1} − ℓ_i)     on log-likelihoods,
 * swapping the β-index of the two replicas, never their states.
 *   d_ll       [n_temps][n_chains]  log-likelihood of replica r of chain c — the layout an all_gather over ranks that
 *                                   each own a contiguous block of replicas produces, so no transpose is needed
 *   d_beta     [n_temps]            inverse-temperature ladder (slot order)
 *   d_slot2rep [n_chains][n_temps]  in/out permutation: which replica sits at ladder slot t
 * `parity` 0 swaps pairs (0,1),(2,3)…; 1 swaps (1,2),(3,4)…  */
int32_t octo_pt_swap_device(octo_ctx* ctx, const double* d_ll_by_replica, const double* d_beta,
                            int32_t* d_slot2rep, int32_t n_temps, int64_t n_chains,
                            int32_t parity, uint64_t seed, uint64_t step,
                            int32_t* d_accepted, void* hip_stream);

/* ---- Parallel tempering across processes (BASELINE config 5): one process per GPU, temperatures (replicas) split
 * contiguously over the ranks, ONE collective per swap step — an all-gather of the per-replica log-likelihoods over RCCL —
 * followed by the deterministic swap kernel above on every rank (same seed, same step: every rank derives the same
 * permutation, so nothing else is exchanged). Stands in for Pigeons' communication step reached through
 * ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:76-128 (docs/src/parallel-sampling.md:64-80).
 *   octo_comm_unique_id   rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks by
 *                         whatever the host has (MPI, a file, a socket);
 *   octo_comm_create      every rank joins (ncclCommInitRank) with its context's device; world = 1 needs no id and no RCCL;
 *   octo_pt_step_device   d_ll_local [n_temps/world][n_chains] -> all-gather into d_ll_all [n_temps][n_chains] (rank order
 *                         = replica order, so the layout is the one octo_pt_swap_device reads) -> swap, all on hip_stream.
 * RCCL (librccl.so.1) is bound at run time, preferring a copy the process already maps. */
int32_t octo_comm_unique_id(uint8_t* out128);
int32_t octo_comm_create(octo_ctx* ctx, const uint8_t* unique_id128, int32_t rank, int32_t world);
int32_t octo_comm_destroy(octo_ctx* ctx);
int32_t octo_pt_step_device(octo_ctx* ctx, const double* d_ll_local, double* d_ll_all /* [n_temps][n_chains]; unused when world = 1 */,
                            const double* d_beta, int32_t* d_slot2rep, int32_t n_temps, int64_t n_chains,
                            int32_t parity, uint64_t seed, uint64_t step, int32_t* d_accepted, void* hip_stream);

/* The same step on HOST arrays (blocking): ll_local [n_temps/world][n_chains], beta [n_temps], slot2rep [n_chains][n_temps] in/out, accepted
 * [n_temps] in/out or NULL — for a driver whose replicas live in host memory (Julia Vectors: julia/OctofitterHIP.jl: octofit_pigeons_hip; its
 * executable twin is host/tempering.py: TemperedSwap.swap_step_host). */
int32_t octo_pt_step(octo_ctx* ctx, const double* ll_local, const double* beta, int32_t* slot2rep, int32_t n_temps, int64_t n_chains,
                     int32_t parity, uint64_t seed, uint64_t step, int32_t* accepted);

#ifdef __cplusplus
}
#endif
#endif /* OCTOFITTER_HIP_H */
