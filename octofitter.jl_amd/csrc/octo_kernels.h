// octo_kernels.h — the kernels of one batched evaluation (gfx950):
//
//   k_setup   per (walker, planet): orbit-constructor constants (PlanetOrbits KepOrbit / Visual /
//             RadialVelocityOrbit ctor; witnesses src/parameterizations.jl:62-64, 215-216) and the
//             per-walker validity flag (src/logdensitymodel.jl:120-124, src/likelihoods/system.jl:214-221).
//   k_main    THE hot kernel. grid = (walker tiles of 64) × (row tasks); block = 4 waves that share a
//             walker tile and split the task's rows; lane = walker. Fuses _kepsolve_all!
//             (system.jl:250-269), simulate! + ln_like of every observation kind
//             (relative-astrometry.jl:104-142,166-253; rv-absolute.jl:135-204; rv-absolute-margin.jl:106-185;
//             rv-relative.jl:121-211) and the reverse sweep into ~9-12 running sums per planet; the four
//             waves' sums are combined through LDS in a fixed order, one partial per (tile, task).
//   k_finish  per walker: fixed-order sum of the task partials (deterministic), closed-form terms,
//             and the map from the running sums to ∂ll/∂(a,e,i,ω,Ω,tp,M,plx,mass) and nuisances.
#pragma once
#include <type_traits>

#include "octo_device.h"
#include "octofitter_hip.h"

namespace octo {

constexpr int MAXP = OCTO_MAX_PLANETS;                 // planets per dataset (array sizes)
constexpr int MAXP_T = OCTO_MAX_PLANETS_ALL_KINDS;      // … that the templated kernels (k_main<P>, k_small<P>, k_finish<P>, k_hgca<P>) are compiled for
constexpr int ROW_STRIDE = 8;   // doubles per observation row record (64 B = one s_load_dwordx16)
constexpr int WPB = 4;          // waves per k_main block (row split + LDS combine)
#ifndef OCTO_FIN_G
#define OCTO_FIN_G 16
#endif
#ifndef OCTO_FIN_UNROLL
#define OCTO_FIN_UNROLL 4
#endif
constexpr int FIN_G = OCTO_FIN_G;      // waves of a single-planet k_finish block (block = 64 walkers x FIN_G waves). The kernel is pure memory latency — a
                                         // tile's partials are tasks x NACC rows of 512 B written by other CUs — so what counts is loads in flight:
                                         // 16 waves x 4 tasks unrolled (8 x 2 in round 2: 10.7 us at 1 250 walkers x 76 tasks, 9.4 us at 1e4 x 34)
// several planets (finish_tile_multi): waves per planet that share the planet's tasks; block = 1 + P·fin_nwr(P) waves
constexpr int fin_nwr(int n_planets) { return n_planets == 2 ? 3 : (n_planets == 3 ? 2 : 1); }      // (four planets: 5 waves — a ninth wave would cap the block at 168 VGPRs per lane, which the O'Neil layouts spill)
#ifndef OCTO_FIN_CH
#define OCTO_FIN_CH 6
#endif
#ifndef OCTO_FIN_FAST
#define OCTO_FIN_FAST true
#endif
constexpr int FIN_CH = OCTO_FIN_CH;      // rows of the LDS combine per chunk: 6 x 16 x 512 B = 48 KB

// kind mask bits
constexpr int KM_RADEC = 1, KM_SEPPA = 2, KM_RVABS = 4, KM_MARG = 8, KM_RVREL = 16, KM_COR = 32, KM_ONEIL = 64;
constexpr int KM_RV = KM_RVABS | KM_MARG | KM_RVREL;
constexpr int KM_ALL = 127;      // every ROW kind (the epoch-loop kernels' kind sets)
// The dataset holds an OCTO_HGCA table (hgca.jl:155-400: no epoch loop). Only k_small looks at this bit — its variants compiled with it
// carry the proper-motion-anomaly block (extra blocks of the launch, one input direction per wave); k_main / k_finish / k_marg are
// instantiated with the bit stripped (the term reaches k_finish through `extra`, a run-time pointer). Rounds 2-3 compiled that block into
// EVERY k_small<NUIS> variant: 20 KB of code, registers and (through hgca_setup) a scratch allocation on calls that have no such table.
constexpr int KM_HGCA = 128;

struct DevObs {
    int32_t kind, planet, has_cor;
    float dm_max;        // largest 2π·(t_j − t_{j−1}) of the table's rows [rad·day]; 0 for a table of one row (or a context created with OCTO_WARM=0)
    int64_t n;
    const double* raw;   // [n][8]: astrom {t,y1,y2,s1,s2,cor,dm,key}; rv {t,rv,σ,trend basis,0,0,dm,key}; dm = 2π·(t − t of the previous row), 0 in row 0
    const double* pre;   // [n][8]: astrom {t,y1,y2,p11,p22,p12,dm,key} (Σ⁻¹ entries); rv {t,rv,1/σ²,0,0,0,dm,key}
    // The warm start's candidate step bounds [rad·day], in order of preference (octo_device.h: KWarm, warm_init below): quantiles of the table's
    // |dm| — the 97 % one first, then smaller ones — each rounded up to a float; 0 = no entry. A wave takes the FIRST entry that none of its
    // lanes vetoes; rows whose own step exceeds it (slot 7 of the record, `key`: |dm|, or +Inf for a row that must start cold) are solved cold
    // behind a scalar branch. One seasonal gap then costs one cold row, not the table (round 5 used dm_max for every row).
    float dm_ladder[WARM_LADDER];
};

struct Task {
    int32_t obs, row0, nrows, chunk;   // chunk = rows per wave: the block's WPB waves split the task's nrows
    float key_max;                     // largest |2π Δt| of the task's rows, each wave's first row aside (+Inf for a non-finite step): a wave whose bound
                                       // covers it — and whose chunk is within WARM_RESTART — takes the warm loop WITHOUT the per-row test
    int32_t pad[3];
};

struct DevConsts {
    double k_yr, yd, au2m, sec2yr, mas_per_au_per_plx /* rad2as/pc2au */, mjup2msol;
};

struct EvalArgs {
    const DevObs* obs;
    const Task* tasks;
    const double* task_const;     // [n_tasks] walker-independent additive constant of the task's rows
    const int32_t* obs_range;     // [n_obs][2] first and one-past-last task of each observation (tasks are grouped by observation)
    const double* obs_const;      // [n_obs] Σ task_const over the observation's tasks, in task order
    int32_t n_obs, n_tasks, n_planets, n_hblocks;      // n_hblocks: k_small only — extra blocks per walker that compute the HGCA term
    int32_t warm, fin_fused;                    // warm 1: k_main may take the warm-started row loop (octo_ctx option OCTO_OPT_WARM_START / _BATCH_INVARIANT);
                                                // fin_fused 1: the launch has ONE task and k_main's blocks finish their own tile (fin_in_main): no partials, no k_finish
    int32_t task0, n_rblocks;                   // first task of this launch (k_main grid.y is relative to it); k_small: blocks per walker that take row tasks
    int32_t orbit_kind[MAXP];
    int32_t has_mass[MAXP];
    const double* elems;          // [P*9][ld]
    const double* nuis;           // [n_obs*3][ld] or null
    int64_t ld, W;
    int64_t ws_in, ws_out;        // k_small only: walker stride of the inputs (elems, nuis) and of the outputs (ll, g_elems, g_nuis).
                                  // 1 = the C ABI's SoA layout (walker fastest); the host-buffer path stages walker-major
                                  // (ld = 1, ws = values per walker) so that one block's inputs are one contiguous PCIe read
    double* wc;                   // [P*NWC][ldw]
    int32_t* valid;               // [n_planets][ldw]
    double* partials;             // [n_tasks*NACC][ldw]
    const double* marg;           // [n_obs*2][ldw]: μ̂ and A of each marginalised-RV table (grad pass) or null
    double* marg_out;
    double* extra;                // [1 + P*9 + n_obs*3][ldw]: ll and input-gradient of the terms computed outside the epoch loop
                                  // (k_hgca), added by k_finish; or null
    const double* sctab;          // [SCT_N][2] sin/cos grid (octo_device.h: sincos_table), copied to LDS by every k_main block
    const int32_t* perm;          // [W] or null: walker of each tile position (octo_tile.h: k_tile_sort). One- and two-planet fused launches only: k_main gathers its
                                  // tile's inputs through it, the partials stay in tile order, k_finish<P <= 2, …, FROM_WC = false> scatters the results back
    int64_t ldw;
    double* ll_out; double* g_elems; double* g_nuis;
    // The tail of the standard parameterisation (octo_model.h; src/logdensitymodel.jl:110-146,169-177) for big batches, run by k_finish right
    // after a tile's adjoints are known: lp = prior + ll with the callback's rules, ∇θ_t[d] = ∂prior/∂θ_t[d] + Σ_k J[k][d]·ḡ[k]. Null mt_lpp:
    // a plain likelihood evaluation. (Rounds 1-3: a kernel of its own, k_model_bwd, behind k_finish: 7 µs + a launch boundary per callback.)
    const double* mt_Jc;          // [2·n_in][mt_ld]     compact Jacobian of the kernel inputs (elements, then nuisances) w.r.t. θ_t, from k_model_fwd:
                                  //                     ∂input k/∂θ_t[i0_k], ∂input k/∂θ_t[i1_k] (octo_model.h)
    const double* mt_gtp;         // [n_el][mt_ld]       ∂tp/∂(element) of the element's planet, 0 where tp is not derived from them
    const octo_source* mt_esrc;   // [n_el]              the model's element and nuisance sources (which θ_t an input reads); mt_nsrc may be null
    const octo_source* mt_nsrc;
    const double* mt_glp;         // [D][mt_ld]          ∂(prior + UnitLength terms)/∂θ_t
    const double* mt_lpp;         // [mt_ld]             prior + UnitLength terms (−Inf for a non-finite θ_t)
    double* mt_lp; double* mt_grad;      // outputs: lp[W]; grad[D][mt_ldo] or null
    int64_t mt_ld, mt_ldo;
    int32_t mt_D, mt_n_nu;        // θ_t dimension; nuisance rows that have a Jacobian (0: the model has no nuisance variables)
    DevConsts c;
};

template <int P, bool GRAD, bool NUIS, int KM>
struct Layout {
    static constexpr bool HAS_RV = (KM & KM_RV) != 0;
    static constexpr bool HAS_MARG = (KM & KM_MARG) != 0;
    static constexpr bool HAS_ASTROM = (KM & (KM_RADEC | KM_SEPPA)) != 0;
    static constexpr bool HAS_COR = (KM & KM_COR) != 0;
    static constexpr int OFF_S = 0;
    static constexpr int OFF_NU = 1;
    static constexpr int N_NU = (GRAD && NUIS) ? 3 : 0;
    static constexpr int OFF_MARG = OFF_NU + N_NU;
    static constexpr int N_MARG = HAS_MARG ? 3 : 0;
    // O'Neil observable-based prior (prior-observable.jl:123-135): Σ_j |t_j| and, for the gradient, Σ sgn·∂t/∂e, Σ sgn·∂t/∂M̄A,
    // Σ sgn·∂t/∂M̄A·(t−tp) — scaled by 2/Σ|t_j| in k_finish
    static constexpr bool HAS_ONEIL = (KM & KM_ONEIL) != 0;
    static constexpr int OFF_ONEIL = OFF_MARG + N_MARG;
    static constexpr int N_ONEIL = HAS_ONEIL ? (GRAD ? 4 : 1) : 0;
    static constexpr int OFF_PL = OFF_ONEIL + N_ONEIL;
    // per planet, all weighted by the planet's coefficient in the model:
    //   U1 Σ cosE·r̄a  U2 Σ sinE·r̄a  U3 Σ cosE·d̄ec  U4 Σ sinE·d̄ec  U5 Σ r̄a  U6 Σ d̄ec
    //   GE Σ M̄·sinE (+ direct RV terms)  GM Σ M̄  GT Σ M̄·(t−tp)  [GC ∂/∂(m/M)]  [GK ∂/∂K  GW ∂/∂ω direct]
    enum { U1 = 0, U2, U3, U4, U5, U6, GE, GM, GT, GC, GK, GW };
    static constexpr int PL_N = !GRAD ? 0 : (HAS_RV ? 12 : (P > 1 ? 10 : 9));
    static constexpr int NACC = OFF_PL + P * PL_N;
};

template <int P, bool GRAD, bool NUIS, int KM>
using AccArr = double[Layout<P, GRAD, NUIS, KM>::NACC];

// ------------------------------------------------------------------------------------ k_setup
// Orbit-constructor constants of one (walker, planet): everything the row loop and the finish need that depends on the walker
// only. Shared by k_setup (stores them in `wc` for the big-batch kernels) and k_small (keeps them in registers).
struct SetupOut {
    double v[NWC];
    double el[OCTO_N_EL];      // the element rows as read (the finish needs e, M, plx, mass and a ThieleInnesOrbit's A, B, F, G)
    bool ok;
};

// sin/cos of an angle given in radians, call-free: reduce to [−π, π] with a two-term 2π (k = rint(x/2π) is exact, x − k·2π_hi is ONE
// rounding of a number of size π, the second term adds k·2π_lo: absolute error ~4e-16 for |x| < 2^40, where k is still right to
// ±0 and k·|2π − hi − lo| < 1e-20), then the half-angle polynomials — ~35 instructions instead of ocml's ~180 per angle, and no
// function call: a call in a kernel clobbers memory for the compiler (the observation rows then stop being scalar loads) and costs
// the callee-save registers of the calling convention. An angle of 2^40 rad (1.1e12; its own rounding is already 1e-4 rad) or more,
// or a non-finite one, gives NaN: the walker is invalid (-Inf), like a non-finite element (deliberate deviations).
constexpr double SINCOS_MAX_ANGLE = 0x1p40;
__device__ __forceinline__ void sincos_reduced(double x, double& s, double& c) {
    const double k = rint(x * (1.0 / TWO_PI));
    double r = fma(-k, 0x1.921fb54442d18p+2, x);      // 2π hi
    r = fma(-k, 0x1.1a62633145c07p-52, r);            // 2π lo
    sincos_halfangle(r, s, c);
    const bool in_range = fabs(x) < SINCOS_MAX_ANGLE; // false for NaN too
    s = in_range ? s : NAN;
    c = in_range ? c : NAN;
}

// Several angles at once where the arguments are WAVE-UNIFORM (k_small: one parameter set per block): lane j evaluates angle j and the
// results are read back with v_readlane — one pass of the reduction + polynomials (~45 instructions) instead of one per angle. Per
// angle the arithmetic is that of sincos_reduced, so the values are bit-identical to separate calls.
template <int N>
__device__ __forceinline__ void sincos_lanes(const double (&x)[N], double (&s)[N], double (&c)[N]) {
    // angle k into lane k with v_writelane (a select chain on the lane index is turned into a scratch-memory lookup table by the compiler)
    int xlo = __double2loint(x[0]), xhi = __double2hiint(x[0]);
#pragma unroll
    for (int k = 1; k < N; ++k) {
        const int slo = __builtin_amdgcn_readfirstlane(__double2loint(x[k])), shi = __builtin_amdgcn_readfirstlane(__double2hiint(x[k]));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(xlo) : "s"(slo), "n"(k));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(xhi) : "s"(shi), "n"(k));
    }
    const double xl = __hiloint2double(xhi, xlo);
    double sl, cl;
    sincos_reduced(xl, sl, cl);
#pragma unroll
    for (int k = 0; k < N; ++k) { s[k] = lane_value(sl, k); c[k] = lane_value(cl, k); }
}

// √x from v_rsq_f64 + Newton (rsqrt_nr), with one correction step on the product: ≤ 1 ulp, 14 instructions (ocml: ~25).
__device__ __forceinline__ double sqrt_fast(double x) {
    const double y = rsqrt_nr(x);
    const double s0 = x * y;
    const double s1 = fma(fma(-s0, s0, x), 0.5 * y, s0);
    return (x > 0.0 && x < 1.0e300) ? s1 : sqrt(x);       // 0, Inf, NaN, negatives: the library's edge handling
}

// FAST (k_small): reciprocal-multiply instead of IEEE division, rsqrt-based roots, polynomial sincos. At the clock a
// mostly-idle GPU runs one short kernel at, every 100 serial FP64 instructions are about a microsecond of latency.
// LANES (k_small only): the elements are wave-uniform, so the three angles go through sincos_lanes in one pass.
//
// The constructor is written as four independent PIECES + an assembly, so that the fused k_main prologue can hand one piece to each
// of its four waves (round 4: the ~300-instruction chain ran in wave 0 alone while the block's other three SIMDs waited — ~3.5 µs of a
// one-round launch): the sin/cos of i, of ω, of Ω, and the scalars (a, period, β, K/sin i, m/M, starter constants). setup_planet_vals
// is the same pieces in sequence, so every kernel derives bit-identical constants.
struct SetupScalars { double sma, T, invP, beta, eob, K0, mu, f32a, f32b; };

template <bool FAST> __device__ __forceinline__ double setup_fdiv(double x, double y) { return FAST ? x * rcp_nr<2>(y) : x / y; }
template <bool FAST> __device__ __forceinline__ double setup_fsqrt(double x) { return FAST ? sqrt_fast(x) : sqrt(x); }

// the angle as the orbit constructor sees it. WHICH: 0 = i (KepOrbit ctor: rem(i, π, RoundDown)), 1 = ω, 2 = Ω (rem2pi(Ω, RoundDown));
// a RadialVelocityOrbit has neither i nor Ω (0), a ThieleInnesOrbit none of the three (the rows carry A, B, F, G)
template <int WHICH>
__device__ __forceinline__ double setup_angle_arg(const double (&elv)[OCTO_N_EL], int orbit_kind) {
    const bool radvel = orbit_kind == OCTO_ORBIT_RADVEL;
    if constexpr (WHICH == 0) { const double inc = radvel ? 0.0 : elv[OCTO_EL_I]; return inc - PI * floor(inc / PI); }
    else if constexpr (WHICH == 1) return elv[OCTO_EL_W];
    else { const double Om = radvel ? 0.0 : elv[OCTO_EL_O]; return Om - TWO_PI * floor(Om / TWO_PI); }
}
// sin and cos of that angle as the constants use them (RadialVelocityOrbit: sin i = 1, cos i = 0, sin Ω = 0, cos Ω = 1; ThieleInnesOrbit: 0)
template <int WHICH, bool FAST>
__device__ __forceinline__ void setup_angle(const double (&elv)[OCTO_N_EL], int orbit_kind, double& sn, double& cs) {
    const bool radvel = orbit_kind == OCTO_ORBIT_RADVEL;
    if (orbit_kind == OCTO_ORBIT_THIELE_INNES) { sn = 0.0; cs = 0.0; return; }
    const double x = setup_angle_arg<WHICH>(elv, orbit_kind);
    if constexpr (FAST) sincos_reduced(x, sn, cs);
    else sincos(x, &sn, &cs);
    if (radvel && WHICH == 0) { sn = 1.0; cs = 0.0; }
    if (radvel && WHICH == 2) { sn = 0.0; cs = 1.0; }
}

template <bool FAST>
__device__ __forceinline__ SetupScalars setup_scalars(const double (&elv)[OCTO_N_EL], const DevConsts& cst, int orbit_kind, int has_mass) {
    SetupScalars q;
    const bool radvel = orbit_kind == OCTO_ORBIT_RADVEL;
    const bool ti = orbit_kind == OCTO_ORBIT_THIELE_INNES;
    const bool kep = orbit_kind == OCTO_ORBIT_KEP;      // plain KepOrbit: no parallax, positions stay in AU (no astrometry tables)
    const double e = elv[OCTO_EL_E], Mt = elv[OCTO_EL_M];
    const double plx = (radvel || kep) ? 1.0 : elv[OCTO_EL_PLX];
    const double mass = has_mass ? elv[OCTO_EL_MASS] : 0.0;
    double sma = elv[OCTO_EL_A];
    if (ti) {
        // ThieleInnesOrbit: rows a, i, ω, Ω carry A, B, F, G [mas]; a = α/plx   (src/parameterizations.jl:14-19).
        // The reference writes α² = u + √((u+v)(u−v)), u = (A²+B²+F²+G²)/2, v = AG − BF, which cancels in u − v for near-face-on
        // orbits. Same quantity without the cancellation: u ± v are sums of squares, and u + √((u+v)(u−v)) = ½(√(u+v) + √(u−v))².
        const double A = elv[OCTO_EL_TI_A], B = elv[OCTO_EL_TI_B], F = elv[OCTO_EL_TI_F], G = elv[OCTO_EL_TI_G];
        const double pp = 0.5 * ((A + G) * (A + G) + (B - F) * (B - F)), mm = 0.5 * ((A - G) * (A - G) + (B + F) * (B + F));
        sma = setup_fdiv<FAST>((setup_fsqrt<FAST>(pp) + setup_fsqrt<FAST>(mm)) * 0.70710678118654752440, plx);
        q.T = 1.0;
    } else {
        // Thiele-Innes constants (parameterizations.jl:34-37) scaled to mas: T = a · cart2angle
        q.T = (radvel || kep) ? 0.0 : sma * plx * cst.mas_per_au_per_plx;   // parameterizations.jl:215-216
    }
    q.sma = sma;
    const double P_d = cst.k_yr * setup_fsqrt<FAST>(setup_fdiv<FAST>(sma * sma * sma, Mt));       // parameterizations.jl:62
    const double ome2 = 1.0 - e * e;
    q.beta = setup_fsqrt<FAST>(ome2);
    // K = ((2π a)/P_yr)/√(1−e²) · au2m · sec2year · sin i: everything but the sine here
    q.K0 = setup_fdiv<FAST>(setup_fdiv<FAST>(TWO_PI * sma, setup_fdiv<FAST>(P_d, cst.yd)), q.beta) * cst.au2m * cst.sec2yr;
    q.invP = setup_fdiv<FAST>(1.0, P_d);
    q.eob = setup_fdiv<FAST>(e, q.beta);
    q.f32a = pack_f32x2((float)e, (float)(1.0 - e));
    q.f32b = pack_f32x2((float)setup_fdiv<FAST>(MK_K1N, 1.0 + e), 0.0f);
    q.mu = setup_fdiv<FAST>(mass * cst.mjup2msol, Mt);
    return q;
}

// validity of one (walker, planet): every element finite and inside the domain (logdensitymodel.jl:120-124, system.jl:214-221)
template <bool FAST>
__device__ __forceinline__ bool setup_valid(const double (&elv)[OCTO_N_EL], int orbit_kind, int has_mass, double sma) {
    const bool radvel = orbit_kind == OCTO_ORBIT_RADVEL;
    const bool ti = orbit_kind == OCTO_ORBIT_THIELE_INNES;
    const bool kep = orbit_kind == OCTO_ORBIT_KEP;
    const double e = elv[OCTO_EL_E], om = elv[OCTO_EL_W], tp = elv[OCTO_EL_TP], Mt = elv[OCTO_EL_M];
    const double inc = radvel ? 0.0 : elv[OCTO_EL_I], Om = radvel ? 0.0 : elv[OCTO_EL_O];
    const double plx = (radvel || kep) ? 1.0 : elv[OCTO_EL_PLX];
    const double mass = has_mass ? elv[OCTO_EL_MASS] : 0.0;
    bool ok = isfinite(elv[OCTO_EL_A]) && isfinite(e) && isfinite(inc) && isfinite(om) && isfinite(Om) && isfinite(tp) &&
              isfinite(Mt) && isfinite(plx) && isfinite(mass);
    if (FAST && !ti) ok = ok && fabs(om) < SINCOS_MAX_ANGLE && fabs(inc) < SINCOS_MAX_ANGLE && fabs(Om) < SINCOS_MAX_ANGLE;   // sincos_reduced's domain
    return ok && (e >= 0.0) && (e < 1.0) && (sma > 0.0) && (Mt > 0.0) && (plx > 0.0);
}

// the per-walker constants from the pieces (o: NWC slots; the caller's dead stores fall away after inlining)
__device__ __forceinline__ void setup_assemble(double* o, const double (&elv)[OCTO_N_EL], int orbit_kind, double si, double ci, double sw, double cw,
                                               double sO, double cO, const SetupScalars& q) {
    const bool ti = orbit_kind == OCTO_ORBIT_THIELE_INNES;
    const double e = elv[OCTO_EL_E], T = q.T, beta = q.beta;
    double A, B, F, G;
    if (ti) { A = elv[OCTO_EL_TI_A]; B = elv[OCTO_EL_TI_B]; F = elv[OCTO_EL_TI_F]; G = elv[OCTO_EL_TI_G]; }
    else {
        A = cO * cw - sO * sw * ci; B = sO * cw + cO * sw * ci;
        F = -cO * sw - sO * cw * ci; G = -sO * sw + cO * cw * ci;
    }
    o[WC_INVP] = q.invP; o[WC_TP] = elv[OCTO_EL_TP]; o[WC_E] = e; o[WC_BETA] = beta;
    o[WC_EOB] = q.eob;
    o[WC_F32A] = q.f32a;
    o[WC_F32B] = q.f32b;
    o[WC_CB] = T * B; o[WC_CG] = T * G; o[WC_CA] = T * A; o[WC_CF] = T * F;
    o[WC_CGB] = T * G * beta; o[WC_CFB] = T * F * beta;
    o[WC_CBE] = T * B * e; o[WC_CAE] = T * A * e;
    o[WC_K] = q.K0 * si;      // 0 for a ThieleInnesOrbit (no RV tables there)
    o[WC_COSW] = cw; o[WC_SINW] = sw;
    o[WC_MU] = q.mu; o[WC_A] = q.sma;
    o[WC_SINI] = si; o[WC_COSI] = ci; o[WC_SINO] = sO; o[WC_COSO] = cO;
}

template <bool FAST = false, bool LANES = false>
__device__ __forceinline__ SetupOut setup_planet_vals(const double (&elv)[OCTO_N_EL], const DevConsts& cst, int orbit_kind, int has_mass) {
    SetupOut so;
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) so.el[k] = elv[k];
    double si, ci, sw, cw, sO, cO;
    if constexpr (FAST && LANES) {
        if (orbit_kind == OCTO_ORBIT_THIELE_INNES) { si = ci = sw = cw = sO = cO = 0.0; }
        else {
            const double xs[3] = {setup_angle_arg<0>(elv, orbit_kind), setup_angle_arg<1>(elv, orbit_kind), setup_angle_arg<2>(elv, orbit_kind)};
            double ss[3], cs[3];
            sincos_lanes<3>(xs, ss, cs);
            si = ss[0]; ci = cs[0]; sw = ss[1]; cw = cs[1]; sO = ss[2]; cO = cs[2];
            if (orbit_kind == OCTO_ORBIT_RADVEL) { si = 1.0; ci = 0.0; sO = 0.0; cO = 1.0; }
        }
    } else {
        setup_angle<0, FAST>(elv, orbit_kind, si, ci);
        setup_angle<1, FAST>(elv, orbit_kind, sw, cw);
        setup_angle<2, FAST>(elv, orbit_kind, sO, cO);
    }
    const SetupScalars q = setup_scalars<FAST>(elv, cst, orbit_kind, has_mass);
    setup_assemble(so.v, elv, orbit_kind, si, ci, sw, cw, sO, cO, q);
    so.ok = setup_valid<FAST>(elv, orbit_kind, has_mass, q.sma);
    return so;
}

template <bool FAST = false>
__device__ __forceinline__ SetupOut setup_planet(const EvalArgs& a, int p, int64_t woff) {
    const double* el = a.elems + (int64_t)p * OCTO_N_EL * a.ld + woff;
    double elv[OCTO_N_EL];
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = el[(int64_t)k * a.ld];
    return setup_planet_vals<FAST>(elv, a.c, a.orbit_kind[p], a.has_mass[p]);
}

static __global__ __launch_bounds__(256) void k_setup(EvalArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.W) return;
    const int p = blockIdx.y;                          // one thread per (walker, planet)
    const SetupOut so = setup_planet<true>(a, p, w);
    bool ok = so.ok;
    double* o = a.wc + (int64_t)p * NWC * a.ldw + w;
#pragma unroll
    for (int k = 0; k < NWC; ++k) o[(int64_t)k * a.ldw] = so.v[k];
    if (a.nuis && blockIdx.y == 0) {
        for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) ok = ok && isfinite(a.nuis[(int64_t)k * a.ld + w]);
    }
    a.valid[(int64_t)blockIdx.y * a.ldw + w] = ok ? 1 : 0;      // k_finish ANDs the planets' flags
}

// ------------------------------------------------------------------------------------ k_kepler
// Batched PlanetOrbits.kepler_solver(MA, e) (call site src/parameterizations.jl:340) through the same device
// routine k_main uses; exported as octo_kepler_solve so tests can check the solver itself.
// TAB: the throughput kernels' variant (sin/cos of the starter from the sin/cos table, copied to LDS by the block exactly as k_main
// does); otherwise the half-angle polynomials of k_small / k_hgca.
#ifdef OCTO_API_TU      // launched from octo_api.hip only
template <bool TAB>
static __global__ __launch_bounds__(256) void k_kepler(const double* __restrict__ MA, const double* __restrict__ ecc, int64_t n,
                                                       double* E, double* sE, double* cE, const double* __restrict__ sctab) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    SinCosTab tab{nullptr, 0.0, 0.0f, 0u};
    if constexpr (TAB) {
        tab = make_sincos_tab(reinterpret_cast<const double2*>(lds));
        const double2* __restrict__ g = reinterpret_cast<const double2*>(sctab);
        double2* t = reinterpret_cast<double2*>(lds);
        for (int i = threadIdx.x; i < SCT_N; i += blockDim.x) t[i] = g[i];
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PC pc = {};
    const double e = ecc[i];
    pc.invP = 1.0 / TWO_PI; pc.tp = 0.0; pc.e = e; pc.beta = sqrt(1.0 - e * e); pc.eob = e / pc.beta;
    set_starter(pc, (float)e, (float)(1.0 - e), (float)(MK_K1N / (1.0 + e)));
    const KSol s = kepler_solve<2, TAB>(MA[i], pc, tab);
    const bool ok = (e >= 0.0) && (e < 1.0) && isfinite(MA[i]);
    E[i] = ok ? s.E : NAN;
    if (sE) sE[i] = ok ? s.sE : NAN;
    if (cE) cE[i] = ok ? s.cE : NAN;
}

// The warm-started solve on its own (a test hook, octo_debug_kepler_warm): element i is solved cold at MA[i] — that solution is the
// "previous row" — and then advanced by dM[i] with kepler_solve_warm, the lane's bound computed from dM[i] as k_main computes it from the
// table's largest step. used[i] = 1 where the wave took the warm path (the ballot is the wave's: the caller groups its inputs).
static __global__ __launch_bounds__(256) void k_kepler_warm(const double* __restrict__ MA, const double* __restrict__ dMa, const double* __restrict__ ecc,
                                                            int64_t n, double* sE, double* cE, double* used, const double* __restrict__ sctab) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const SinCosTab tab = make_sincos_tab(reinterpret_cast<const double2*>(lds));
    {
        const double2* __restrict__ g = reinterpret_cast<const double2*>(sctab);
        double2* t = reinterpret_cast<double2*>(lds);
        for (int i = threadIdx.x; i < SCT_N; i += blockDim.x) t[i] = g[i];
        __syncthreads();
    }
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = i0 < n ? i0 : n - 1;
    PC pc = {};
    const double e = ecc[i];
    pc.invP = 1.0 / TWO_PI; pc.tp = 0.0; pc.e = e; pc.beta = sqrt(1.0 - e * e); pc.eob = e / pc.beta;
    set_starter(pc, (float)e, (float)(1.0 - e), (float)(MK_K1N / (1.0 + e)));
    const KSol s0 = kepler_solve<2, true>(MA[i], pc, tab);
    KWarm st{s0.sE, s0.cE, s0.invD};
    const float dmx = fabsf((float)dMa[i]);
    double thr = (double)__builtin_amdgcn_exp2f(0.2f * (__builtin_amdgcn_logf((float)WARM_TOL) - 3.0f * __builtin_amdgcn_logf(dmx)));
    if (thr < WARM_MIN_THR) thr = 0.0;      // k_main's veto (warm_init): a bound below WARM_MIN_THR never starts warm — the routine's polynomials count on |dE| < 0.066
    const bool warm = __builtin_amdgcn_ballot_w64(st.invD >= thr) == 0;
    // dm = 2π·Δt with 1/P = 1/2π: ΔM = dMa[i]; t = MA + dM so that the cold fallback solves the same row
    const KSol s = kepler_solve_warm<2>(MA[i] + dMa[i], pc, tab, st, thr, dMa[i] * TWO_PI);
    if (i0 < n) { sE[i] = s.sE; cE[i] = s.cE; used[i] = warm ? 1.0 : 0.0; }
}

#endif      // OCTO_API_TU

// ------------------------------------------------------------------------------------ row bodies
// One observation row for one walker: Kepler solve of every planet, projection, residual, density, and (GRAD) the reverse
// sweep into the running sums. Written once and inlined into both mappings of the epoch loop:
//   k_main   lane = walker: `pc`, the coefficients and nuisances are per-lane registers, the row record is wave-uniform (SGPRs);
//   k_small  lane = epoch:  the row record is per-lane, everything that depends on the walker is wave-uniform.
// TAB: sin/cos of the starter from the block's LDS table (k_main) or from the half-angle polynomials (k_small: no table fill).
template <int P>
struct AstromCoef {
    double f[P];                    // coefficient of each planet's sky offset in the model (relative-astrometry.jl:117-138)
    double jit, j2, ps, na, sn, cn; // θ_obs: jitter, jitter², platescale, northangle and its sin/cos (:170-172)
    bool seppa, oneil;
    int planet, has_cor;
};

template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ AstromCoef<P> astrom_coef_vals(double jit, double ps, double na, int ob_kind, int ob_planet, int ob_has_cor, const PC (&pc)[P]) {
    using L = Layout<P, GRAD, NUIS, KM>;
    AstromCoef<P> c;
    // 1 for the planet the table is attached to, +m/M for strictly-inner companions with a mass.
    if constexpr (P == 1) {
        c.f[0] = 1.0;
    } else {
        double a_this = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) a_this = (p == ob_planet) ? pc[p].a : a_this;
#pragma unroll
        for (int p = 0; p < P; ++p) c.f[p] = (p == ob_planet) ? 1.0 : ((pc[p].a < a_this) ? pc[p].mu : 0.0);
    }
    c.jit = 0.0; c.j2 = 0.0; c.ps = 1.0; c.na = 0.0; c.sn = 0.0; c.cn = 1.0;
    if constexpr (NUIS) {
        c.jit = jit; c.ps = ps; c.na = na;
        sincos_reduced(c.na, c.sn, c.cn);      // call-free, ~35 instructions (ocml: ~180 + a large-argument loop) in every block's prologue
        c.j2 = c.jit * c.jit;
    }
    c.seppa = (KM & KM_SEPPA) && (ob_kind == OCTO_ASTROM_SEPPA || ob_kind == OCTO_ONEIL_SEPPA);
    c.oneil = L::HAS_ONEIL && (ob_kind == OCTO_ONEIL_RADEC || ob_kind == OCTO_ONEIL_SEPPA);
    c.planet = ob_planet; c.has_cor = ob_has_cor;
    return c;
}

template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ AstromCoef<P> astrom_coef(const double* __restrict__ nuis, int64_t ld, int ob_kind, int ob_planet, int ob_has_cor,
                                                     int obs_index, const PC (&pc)[P], int64_t wl) {
    double jit = 0.0, ps = 1.0, na = 0.0;
    if constexpr (NUIS) {
        const double* nu = nuis + (int64_t)obs_index * OCTO_N_NUIS * ld + wl;
        jit = nu[OCTO_NU_JITTER * ld]; ps = nu[OCTO_NU_PLATESCALE * ld]; na = nu[OCTO_NU_NORTHANGLE * ld];
    }
    return astrom_coef_vals<P, GRAD, NUIS, KM>(jit, ps, na, ob_kind, ob_planet, ob_has_cor, pc);
}

// The warm start's per-wave state (octo_device.h: KWarm): the previous row's solution and the lane's bound on 1/D, per planet.
template <int P>
struct WarmState { KWarm st[P]; double thr[P]; uint32_t key_hi; bool row_ok; };

// WLAST (round 6, P >= 2): 1 — the LAST planet takes the warm step (octo_device.h: kepler_warm_step) from ws->st[P − 1], the others the cold solve;
// 2 — the same behind a per-row test (ws->row_ok, wave-uniform): a rejected row re-solves the last planet cold and records that solution.
template <int P, bool GRAD, bool NUIS, int KM, bool TAB, int WARM = 0, bool WCHECK = true, int WLAST = 0>
__device__ __forceinline__ void astrom_row(AccArr<P, GRAD, NUIS, KM>& acc, LogProd& lp, const PC (&pc)[P],
                                           const AstromCoef<P>& co, double t, double y1, double y2, double c3, double c4, double c5,
                                           const SinCosTab& tab, WarmState<P>* ws = nullptr, double dm = 0.0) {
    using L = Layout<P, GRAD, NUIS, KM>;
    // DEFER (k_main's mapping, lane = walker: round 6): every factor of an adjoint sum that is constant over the table's rows for one walker — the
    // planet's coefficient f_p, the jitter, the platescale in front of the northangle term — is applied ONCE per wave after the row loop
    // (astrom_finish_sums) instead of once per row; the ∂/∂(m/M) sum is accumulated for every planet and dropped afterwards where it does not
    // apply (two selects per planet and row before). 11 VALU instructions per two-planet nuisance row less; the value path is untouched.
    constexpr bool DEFER = TAB;
    const double (&f)[P] = co.f;
    const double jit = co.jit, j2 = co.j2, ps = co.ps, na = co.na, sn = co.sn, cn = co.cn;
    const bool seppa = co.seppa, oneil = co.oneil;
    KSol s[P];
    double rap[P], dep[P];      // each planet's own sky offset [mas]
    double ra_m, dec_m;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if constexpr (WARM != 0) s[p] = kepler_solve_warm<1, WARM == 1>(t, pc[p], tab, ws->st[p], ws->thr[p], dm, WCHECK ? ws->row_ok : true);
        else if constexpr (WLAST != 0) { if (p == P - 1) s[p] = kepler_warm_step<1>(t, pc[p], ws->st[p], dm); else s[p] = kepler_solve<1, TAB>(t, pc[p], tab); }
        else s[p] = kepler_solve<1, TAB>(t, pc[p], tab);
        if constexpr (WLAST == 2) {
            // the rare row the a-priori test rejects (wave-uniform, decided before the row: WarmState::row_ok): the warm step's result is dropped for the
            // cold solve — a triangle behind the straight-line code both planets share, not two arms
            if (p == P - 1 && __builtin_expect(!ws->row_ok, 0)) {
                s[p] = kepler_solve<1, TAB>(t, pc[p], tab);
                ws->st[p].sE = s[p].sE; ws->st[p].cE = s[p].cE; ws->st[p].invD = s[p].invD;
            }
        }
        rap[p] = fma(pc[p].cB, s[p].cE, fma(pc[p].cGb, s[p].sE, -pc[p].cBe));
        dep[p] = fma(pc[p].cA, s[p].cE, fma(pc[p].cFb, s[p].sE, -pc[p].cAe));
    }
    if constexpr (L::HAS_ONEIL) {
        if (oneil) {
            // M = meananom(sol) = E − e sin E; t = 3M(e + cos E) + 2(−2 + e² + e cos E) sin E   prior-observable.jl:129-133
            double sE = 0.0, cE = 0.0, ee = 0.0, invD = 0.0, dtp = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const bool me = (P == 1) || (p == co.planet);
                sE = me ? s[p].sE : sE; cE = me ? s[p].cE : cE; ee = me ? pc[p].e : ee; invD = me ? s[p].invD : invD; dtp = me ? s[p].dt : dtp;
            }
            double E = 0.0;                                  // E ∈ [−π, π], as the solver returns it (eccanom(sol))
#pragma unroll
            for (int p = 0; p < P; ++p) E = ((P == 1) || (p == co.planet)) ? s[p].E : E;
            const double Mm = fma(-ee, sE, E);
            const double c2 = fma(ee, ee + cE, -2.0);       // −2 + e² + e cos E
            const double tt = fma(3.0 * Mm, ee + cE, 2.0 * c2 * sE);
            acc[L::OFF_ONEIL] += fabs(tt);
            if constexpr (GRAD) {
                const double sg = tt < 0.0 ? -1.0 : 1.0;
                const double D = fma(-ee, cE, 1.0);
                const double tM = 3.0 * (ee + cE);                                        // ∂t/∂M
                const double tE = fma(-3.0 * Mm, sE, 2.0 * fma(c2, cE, -(ee * sE * sE))) + tM * D;      // total ∂t/∂E (M = E − e sinE)
                const double te = fma(2.0 * sE, 2.0 * ee + cE, 3.0 * Mm) - tM * sE;       // total ∂t/∂e at fixed E
                const double Mb = sg * tE * invD;                                          // through E(M̄A, e)
                acc[L::OFF_ONEIL + 1] += sg * te + Mb * sE;
                acc[L::OFF_ONEIL + 2] += Mb;
                acc[L::OFF_ONEIL + 3] = fma(Mb, dtp, acc[L::OFF_ONEIL + 3]);
            }
        }
    }
    if constexpr (P == 1) {
        ra_m = rap[0]; dec_m = dep[0];
    } else {
        ra_m = 0.0; dec_m = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) { ra_m = fma(f[p], rap[p], ra_m); dec_m = fma(f[p], dep[p], dec_m); }
    }
    // residuals
    double r1, r2, irho = 1.0, u1 = 0.0, u2 = 0.0;
    if (seppa) {
        // relative-astrometry.jl:192-202
        const double rho2 = fma(ra_m, ra_m, dec_m * dec_m);
        irho = rsqrt_nr(rho2);                         // ρ = ρ²·(1/ρ) enters r2 through one explicit FMA below: a separate
                                                       // product would be contracted differently by the forward-only and
                                                       // the gradient instantiation, and their values must agree bitwise
        const double pa = atan2_fast(ra_m, dec_m);
        double dpa = (y1 + na) - pa + PI;
        dpa = rem_2pi_trunc(dpa) - PI;                 // Julia `%`: truncated remainder
        dpa = dpa < -PI ? dpa + TWO_PI : dpa;
        r1 = dpa;
        r2 = fma(-rho2, irho, y2 * ps);
    } else {
        // relative-astrometry.jl:210-215: the data are rotated by −northangle and scaled
        if constexpr (NUIS) {
            u1 = fma(y1, cn, y2 * sn);
            u2 = fma(y2, cn, -(y1 * sn));
            r1 = fma(ps, u1, -ra_m);
            r2 = fma(ps, u2, -dec_m);
        } else {
            r1 = y1 - ra_m;
            r2 = y2 - dec_m;
        }
    }
    // density; g1, g2 = ∂ll/∂r1, ∂ll/∂r2
    double g1, g2;
    if constexpr (!NUIS) {
        // precomputed Σ⁻¹ (the jitter == 0 branch, relative-astrometry.jl:218-219)
        double a1, a2;
        if constexpr (L::HAS_COR) { a1 = fma(c3, r1, c5 * r2); a2 = fma(c5, r1, c4 * r2); }
        else { a1 = c3 * r1; a2 = c4 * r2; }
        acc[L::OFF_S] = fma(r1, a1, fma(r2, a2, acc[L::OFF_S]));   // Σ rᵀΣ⁻¹r ; ll = const − ½Σ
        g1 = -a1; g2 = -a2;
    } else {
        const double v1 = fma(c3, c3, j2), v2 = fma(c4, c4, j2);   // hypot(σ, jitter)², :234-235
        const double v12 = v1 * v2;
        // one reciprocal for 1/v1 and 1/v2; k_main (DEFER) takes ONE Newton step on v_rcp_f64 — 2^-46 = 1.4e-14 relative on a row's χ² term, three
        // orders below the golden-vector bar — k_small keeps two
        const double iv12 = rcp_nr<DEFER ? 1 : 2>(v12);
        const double iv1 = iv12 * v2, iv2 = iv12 * v1;
        double a1, a2;                                             // Σ⁻¹ r
        if (L::HAS_COR && co.has_cor) {
            const double cor = c5;
            const double omc = 1.0 - cor * cor;
            const double ic = rcp_nr<2>(omc);
            const double is = rsqrt(v12);                          // 1/(σ1 σ2)
            a1 = fma(r1, iv1, -(cor * r2 * is)) * ic;
            a2 = fma(r2, iv2, -(cor * r1 * is)) * ic;
            lp.mul(v12 * omc);
        } else {
            a1 = r1 * iv1; a2 = r2 * iv2;
            lp.mul(v12);
        }
        // ll = −n·log2π − ½Σ(log|Σ| + rᵀΣ⁻¹r); the logs via lp. Explicit FMAs: the forward-only and the gradient
        // instantiation must round this sum identically (the compiler would contract it differently around q1, q2)
        acc[L::OFF_S] = fma(r1, a1, fma(r2, a2, acc[L::OFF_S]));
        g1 = -a1; g2 = -a2;
        if constexpr (GRAD) {
            if constexpr (DEFER) acc[L::OFF_NU + OCTO_NU_JITTER] += fma(fma(r1, a1, -1.0), iv1, fma(r2, a2, -1.0) * iv2);      // × jitter after the loop
            else acc[L::OFF_NU + OCTO_NU_JITTER] += jit * fma(fma(r1, a1, -1.0), iv1, fma(r2, a2, -1.0) * iv2);
            if (seppa) {
                acc[L::OFF_NU + OCTO_NU_PLATESCALE] = fma(g2, y2, acc[L::OFF_NU + OCTO_NU_PLATESCALE]);
                acc[L::OFF_NU + OCTO_NU_NORTHANGLE] += g1;
            } else {
                acc[L::OFF_NU + OCTO_NU_PLATESCALE] += g1 * u1 + g2 * u2;
                if constexpr (DEFER) acc[L::OFF_NU + OCTO_NU_NORTHANGLE] += g1 * u2 - g2 * u1;      // × platescale after the loop
                else acc[L::OFF_NU + OCTO_NU_NORTHANGLE] += ps * (g1 * u2 - g2 * u1);
            }
        }
    }
    if constexpr (GRAD) {
        // adjoint of the model position
        double rab, deb;
        if (seppa) {
            const double pab = -g1, rhob = -g2;
            rab = (rhob * ra_m + pab * dec_m * irho) * irho;
            deb = (rhob * dec_m - pab * ra_m * irho) * irho;
        } else {
            rab = -g1; deb = -g2;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double* g = &acc[L::OFF_PL + p * L::PL_N];
            const double ra_f = (P == 1 || DEFER) ? rab : f[p] * rab, de_f = (P == 1 || DEFER) ? deb : f[p] * deb;      // DEFER: × f_p after the loop
            g[L::U1] = fma(s[p].cE, ra_f, g[L::U1]);
            g[L::U2] = fma(s[p].sE, ra_f, g[L::U2]);
            g[L::U3] = fma(s[p].cE, de_f, g[L::U3]);
            g[L::U4] = fma(s[p].sE, de_f, g[L::U4]);
            g[L::U5] += ra_f;
            g[L::U6] += de_f;
            if constexpr (P > 1) {
                if constexpr (DEFER) g[L::GC] += fma(rab, rap[p], deb * dep[p]);      // dropped after the loop for the attached planet and for f_p = 0
                else g[L::GC] += (p == co.planet) ? 0.0 : ((f[p] != 0.0) ? fma(rab, rap[p], deb * dep[p]) : 0.0);
            }
            // Ē = r̄a·∂ra/∂E + d̄ec·∂dec/∂E ;  M̄ = Ē/(1 − e cos E)
            const double dra = fma(pc[p].cGb, s[p].cE, -(pc[p].cB * s[p].sE));
            const double dde = fma(pc[p].cFb, s[p].cE, -(pc[p].cA * s[p].sE));
            const double Mb = fma(ra_f, dra, de_f * dde) * s[p].invD;
            g[L::GE] = fma(Mb, s[p].sE, g[L::GE]);          // ∂E/∂e = sin E/(1 − e cos E); the rest of ē in k_finish
            g[L::GM] += Mb;
            g[L::GT] = fma(Mb, s[p].dt, g[L::GT]);
        }
    }
}

template <int P>
struct RvCoef {
    double gc[P];                   // coefficient of K_p·V_p in the RV model
    double gK[P];                   // gc_p·K_p
    double ib2[P];                  // 1/β² of each planet (the gradient's closed forms)
    double off, jit, j2, mu_hat, iA;
    double trend;                   // coefficient of the table's trend basis column (OCTO_NU_RV_TREND)
    bool rel, marg;
    int planet;
};

template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ RvCoef<P> rv_coef_vals(double off, double jit, double trend, const double* __restrict__ margp, int64_t ldw,
                                                  int ob_kind, int ob_planet, int obs_index, const PC (&pc)[P], int64_t wl) {
    RvCoef<P> c;
    // RV_REL: +1 for this planet (rv-relative.jl:143), −m/M for strictly-inner massive companions (:148-156);
    // absolute RV: −m/M for every planet (rv-absolute.jl:146-155).
    c.rel = (KM & KM_RVREL) && ob_kind == OCTO_RV_REL;
    {
        double a_this = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) a_this = (p == ob_planet) ? pc[p].a : a_this;
#pragma unroll
        for (int p = 0; p < P; ++p)
            c.gc[p] = c.rel ? ((p == ob_planet) ? 1.0 : ((pc[p].a < a_this) ? -pc[p].mu : 0.0)) : -pc[p].mu;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) { c.ib2[p] = GRAD ? 1.0 / (pc[p].beta * pc[p].beta) : 0.0; c.gK[p] = c.gc[p] * pc[p].K; }
    c.marg = (KM & KM_MARG) && ob_kind == OCTO_RV_ABS_MARG;
    c.off = 0.0; c.jit = 0.0; c.j2 = 0.0; c.trend = 0.0;
    if constexpr (NUIS) {
        c.off = c.marg ? 0.0 : off;
        c.jit = jit;
        c.j2 = c.jit * c.jit;
        c.trend = trend;
    }
    c.mu_hat = 0.0; c.iA = 0.0;
    if (GRAD && c.marg && margp) {
        c.mu_hat = margp[((int64_t)obs_index * 2 + 0) * ldw + wl];
        c.iA = 1.0 / margp[((int64_t)obs_index * 2 + 1) * ldw + wl];
    }
    c.planet = ob_planet;
    return c;
}

template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ RvCoef<P> rv_coef(const double* __restrict__ nuis, int64_t ld, const double* __restrict__ margp, int64_t ldw,
                                             int ob_kind, int ob_planet, int obs_index, const PC (&pc)[P], int64_t wl) {
    double off = 0.0, jit = 0.0, trend = 0.0;
    if constexpr (NUIS) {
        const double* nu = nuis + (int64_t)obs_index * OCTO_N_NUIS * ld + wl;
        off = nu[OCTO_NU_RV_OFFSET * ld]; jit = nu[OCTO_NU_RV_JITTER * ld]; trend = nu[OCTO_NU_RV_TREND * ld];
    }
    return rv_coef_vals<P, GRAD, NUIS, KM>(off, jit, trend, margp, ldw, ob_kind, ob_planet, obs_index, pc, wl);
}

template <int P, bool GRAD, bool NUIS, int KM, bool TAB, int WARM = 0, bool WCHECK = true, int WLAST = 0>
__device__ __forceinline__ void rv_row(AccArr<P, GRAD, NUIS, KM>& acc, LogProd& lp, const PC (&pc)[P],
                                       const RvCoef<P>& co, double t, double rv, double c2, double basis, const SinCosTab& tab,
                                       WarmState<P>* ws = nullptr, double dm = 0.0) {
    using L = Layout<P, GRAD, NUIS, KM>;
    // DEFER (k_main's mapping: round 6, as in astrom_row): every planet sum of an RV row is linear in rvb·(a function of the solution) with a factor
    // gc_p·K_p, gc_p or K_p that is constant over the rows — applied once per wave after the loop (rv_finish_sums); ∂/∂K and ∂/∂(m/M) are the SAME
    // sum Σ V·rvb. Six VALU instructions per planet and row less.
    constexpr bool DEFER = TAB;
    const double (&gc)[P] = co.gc;
    const bool rel = co.rel, marg = co.marg;
    const double jit = co.jit, j2 = co.j2, mu_hat = co.mu_hat, iA = co.iA;
    KSol s[P];
    double V[P], cnu[P], snu[P];
    // offset + trend_function(θ_obs, t) (rv-absolute.jl:143, rv-relative.jl:131, rv-absolute-margin.jl:111): the trend as
    // coefficient × the row's basis value (include/octofitter_hip.h: OCTO_NU_RV_TREND); without nuisances both are zero
    double model = NUIS ? fma(co.trend, basis, co.off) : co.off;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if constexpr (WARM != 0) s[p] = kepler_solve_warm<2, WARM == 1>(t, pc[p], tab, ws->st[p], ws->thr[p], dm, WCHECK ? ws->row_ok : true);
        else if constexpr (WLAST != 0) { if (p == P - 1) s[p] = kepler_warm_step<2>(t, pc[p], ws->st[p], dm); else s[p] = kepler_solve<2, TAB>(t, pc[p], tab); }
        else s[p] = kepler_solve<2, TAB>(t, pc[p], tab);
        if constexpr (WLAST == 2) {
            if (p == P - 1 && __builtin_expect(!ws->row_ok, 0)) {
                s[p] = kepler_solve<2, TAB>(t, pc[p], tab);
                ws->st[p].sE = s[p].sE; ws->st[p].cE = s[p].cE; ws->st[p].invD = s[p].invD;
            }
        }
        cnu[p] = (s[p].cE - pc[p].e) * s[p].invD;             // cos ν
        snu[p] = pc[p].beta * s[p].sE * s[p].invD;            // sin ν
        V[p] = fma(cnu[p] + pc[p].e, pc[p].cw, -(snu[p] * pc[p].sw));   // cos(ν+ω) + e cos ω
        model = fma(DEFER ? co.gK[p] : gc[p] * pc[p].K, V[p], model);      // (co.gK is the same product, formed once per wave: the value is unchanged)
    }
    const double resid = rv - model;
    double iv, var = 1.0;
    if constexpr (NUIS) { var = fma(c2, c2, j2); iv = rcp_nr<DEFER ? 1 : 2>(var); lp.mul(var); } else { iv = c2; }      // (one Newton step in k_main: 1.4e-14, as in astrom_row)
    double rvb;   // ∂ll/∂model
    if (L::HAS_MARG && marg) {
        // rv-absolute-margin.jl:171-180
        if constexpr (L::HAS_MARG) {
            acc[L::OFF_MARG + 0] += iv;
            acc[L::OFF_MARG + 1] = fma(-2.0 * resid, iv, acc[L::OFF_MARG + 1]);
            acc[L::OFF_MARG + 2] = fma(resid * resid, iv, acc[L::OFF_MARG + 2]);
        }
        const double dm = resid - mu_hat;
        rvb = 2.0 * dm * iv;
        if constexpr (GRAD && NUIS) {
            if constexpr (DEFER) acc[L::OFF_NU + OCTO_NU_RV_JITTER] += iv * (dm * dm * iv - 1.0 + iv * iA);      // × 2·jitter after the loop
            else acc[L::OFF_NU + OCTO_NU_RV_JITTER] += 2.0 * jit * iv * (dm * dm * iv - 1.0 + iv * iA);
            acc[L::OFF_NU + OCTO_NU_RV_TREND] = fma(rvb, basis, acc[L::OFF_NU + OCTO_NU_RV_TREND]);
        }
    } else {
        acc[L::OFF_S] = fma(resid * resid, iv, acc[L::OFF_S]);
        rvb = resid * iv;
        if constexpr (GRAD && NUIS) {
            acc[L::OFF_NU + OCTO_NU_RV_OFFSET] += rvb;
            if constexpr (DEFER) acc[L::OFF_NU + OCTO_NU_RV_JITTER] += iv * (resid * resid * iv - 1.0);      // × jitter after the loop
            else acc[L::OFF_NU + OCTO_NU_RV_JITTER] += jit * iv * (resid * resid * iv - 1.0);
            acc[L::OFF_NU + OCTO_NU_RV_TREND] = fma(rvb, basis, acc[L::OFF_NU + OCTO_NU_RV_TREND]);
        }
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double* g = &acc[L::OFF_PL + p * L::PL_N];
            double Vb;
            if constexpr (DEFER) {
                g[L::GK] = fma(V[p], rvb, g[L::GK]);      // Σ V·rvb: × gc_p for ∂/∂K, × −K_p for ∂/∂(m/M), after the loop
                Vb = rvb;                                  // … and × gc_p·K_p for the four sums below
            } else {
                g[L::GK] = fma(gc[p] * V[p], rvb, g[L::GK]);
                const bool via_mu = rel ? (p != co.planet && gc[p] != 0.0) : true;
                g[L::GC] += via_mu ? -(pc[p].K * V[p] * rvb) : 0.0;
                Vb = gc[p] * pc[p].K * rvb;
            }
            // V = cos(ν+ω) + e cos ω with cos ν = (cE − e)/D, sin ν = β sE/D, D = 1 − e cE, in closed form:
            //   ∂V/∂ω = −sin(ν+ω) − e sin ω,   ∂V/∂E = −β sin(ν+ω)/D,   ∂V/∂e at fixed E = cos ω − sin ν · sin(ν+ω)/β²
            const double S = fma(snu[p], pc[p].cw, cnu[p] * pc[p].sw);      // sin(ν+ω)
            g[L::GW] = fma(Vb, -fma(pc[p].e, pc[p].sw, S), g[L::GW]);
            const double VS = Vb * S;
            const double Mb = -(VS * pc[p].beta) * (s[p].invD * s[p].invD);   // Ēᵥ/D = M̄
            double eb = Vb * pc[p].cw;
            eb = fma(-(snu[p] * co.ib2[p]), VS, eb);
            eb = fma(Mb, s[p].sE, eb);                                      // ∂E/∂e = sin E/D
            g[L::GE] += eb;
            g[L::GM] += Mb;
            g[L::GT] = fma(Mb, s[p].dt, g[L::GT]);
        }
    }
}

// What astrom_row<…, DEFER> / rv_row<…, DEFER> left out of the rows: applied once per wave (one task = one table, so the factors are the same for
// every row the sums hold).
template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ void astrom_finish_sums(AccArr<P, GRAD, NUIS, KM>& acc, const AstromCoef<P>& co) {
    using L = Layout<P, GRAD, NUIS, KM>;
    if constexpr (GRAD) {
        if constexpr (L::N_NU > 0) {
            acc[L::OFF_NU + OCTO_NU_JITTER] *= co.jit;
            acc[L::OFF_NU + OCTO_NU_NORTHANGLE] *= co.seppa ? 1.0 : co.ps;
        }
        if constexpr (P > 1) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                double* g = &acc[L::OFF_PL + p * L::PL_N];
                const double fp = co.f[p];
                g[L::U1] *= fp; g[L::U2] *= fp; g[L::U3] *= fp; g[L::U4] *= fp; g[L::U5] *= fp; g[L::U6] *= fp;
                g[L::GE] *= fp; g[L::GM] *= fp; g[L::GT] *= fp;
                g[L::GC] = (p == co.planet || fp == 0.0) ? 0.0 : g[L::GC];
            }
        }
    }
}
template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ void rv_finish_sums(AccArr<P, GRAD, NUIS, KM>& acc, const RvCoef<P>& co, const PC (&pc)[P]) {
    using L = Layout<P, GRAD, NUIS, KM>;
    if constexpr (GRAD && L::HAS_RV) {
        if constexpr (L::N_NU > 0) acc[L::OFF_NU + OCTO_NU_RV_JITTER] *= (L::HAS_MARG && co.marg) ? 2.0 * co.jit : co.jit;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double* g = &acc[L::OFF_PL + p * L::PL_N];
            const double svr = g[L::GK];                                                   // Σ V·rvb
            const bool via_mu = co.rel ? (p != co.planet && co.gc[p] != 0.0) : true;
            g[L::GK] = co.gc[p] * svr;
            g[L::GC] = via_mu ? -(pc[p].K * svr) : 0.0;
            g[L::GW] *= co.gK[p]; g[L::GE] *= co.gK[p]; g[L::GM] *= co.gK[p]; g[L::GT] *= co.gK[p];
        }
    }
}

// ------------------------------------------------------------------------------------ k_main
// The observation rows are immutable while a kernel runs (a dataset never changes after octo_dataset_create), so k_main reads them
// through the CONSTANT address space: a wave-uniform load from it is always a scalar load (s_load into SGPRs), whatever else the
// kernel contains. Left in the global address space the compiler issues scalar loads only while it can prove that nothing in the
// function may have written the memory — a call (the library sincos of the fused prologue), an argument struct passed by reference
// or a volatile asm silently demotes every row to per-lane vector loads (+45 % step time, measured).
typedef const double __attribute__((address_space(4))) * crow_t;
__device__ __forceinline__ crow_t constant_rows(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (crow_t)p;
#pragma clang diagnostic pop
}

// One row record in SGPRs. The row loops fetch row j+1 while row j is computed (the scalar load then has a whole iteration to land):
// every row is a new 64-byte line of the scalar cache, and a kernel that keeps only two or three waves per SIMD (the multi-planet
// nuisance variants) cannot hide that latency behind other waves.
// The loads are inline assembly on purpose: written as C++ loads the compiler sinks them to their first use and waits at once (they are
// pure), whatever the source order. row_issue starts the scalar loads and returns; row_wait is the s_waitcnt the consumers depend on.
// Scalar loads return out of order, so the only usable wait is lgkmcnt(0): a row is therefore waited for BEFORE the next one is
// issued (it was issued a whole row body earlier), and the next one lands behind the body's own LDS wait for the sin/cos table.
typedef int sgpr8_t __attribute__((ext_vector_type(8)));
typedef int sgpr4_t __attribute__((ext_vector_type(4)));
struct RowRegs { sgpr8_t lo; sgpr4_t hi; };      // doubles 0-3 and 4-5 of the record
__device__ __forceinline__ RowRegs row_issue(crow_t p) {
    RowRegs r;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20" : "=&s"(r.lo), "=&s"(r.hi) : "s"(p));
    return r;
}
// the last load issued (into `r`) has landed, with everything before it. `r` is an INPUT of the wait, so its registers stay allocated to
// it until the load has written them (a tuple that is dead in the program is free for the register allocator at once); an in/out
// operand made the compiler copy the in-flight registers ahead of the wait (round 3). That no instruction touches a load's destination
// before the wait is checked on the ISA of every k_main variant — tools/kernel_resources.py: scalar_load_hazards.
__device__ __forceinline__ void row_drain(const RowRegs& r) { asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(r.lo), "s"(r.hi)); }
// wait for `cur`, then start the loads of the following row — one asm block that `cur`'s consumers depend on, so that the compiler
// cannot schedule the first instructions of the row body ahead of the issue
__device__ __forceinline__ RowRegs row_wait_issue(RowRegs& cur, crow_t p) {
    RowRegs r;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x20"
                 : "=&s"(r.lo), "=&s"(r.hi), "+s"(cur.lo), "+s"(cur.hi) : "s"(p));
    return r;
}
__device__ __forceinline__ double row_get(const RowRegs& r, int k) {      // k: compile-time constant after inlining
    return k < 4 ? __hiloint2double(r.lo[2 * k + 1], r.lo[2 * k]) : __hiloint2double(r.hi[2 * (k - 4) + 1], r.hi[2 * (k - 4)]);
}
// The same with the whole 64-byte record (doubles 0-7): the warm-start loops also read slot 6, 2π·(t − t of the previous row).
struct RowRegs8 { sgpr8_t lo; sgpr8_t hi; };
__device__ __forceinline__ RowRegs8 row_issue8(crow_t p) {
    RowRegs8 r;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20" : "=&s"(r.lo), "=&s"(r.hi) : "s"(p));
    return r;
}
__device__ __forceinline__ void row_drain(const RowRegs8& r) { asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(r.lo), "s"(r.hi)); }

__device__ __forceinline__ RowRegs8 row_wait_issue(RowRegs8& cur, crow_t p) {
    RowRegs8 r;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20"
                 : "=&s"(r.lo), "=&s"(r.hi), "+s"(cur.lo), "+s"(cur.hi) : "s"(p));
    return r;
}
__device__ __forceinline__ double row_get(const RowRegs8& r, int k) {
    return k < 4 ? __hiloint2double(r.lo[2 * k + 1], r.lo[2 * k]) : __hiloint2double(r.hi[2 * (k - 4) + 1], r.hi[2 * (k - 4)]);
}

constexpr int NPC = WC_CAE + 1;     // the per-walker constants the row loop reads (PC): WC_INVP … WC_CAE; the rest of `wc` is the finish's

template <int P, bool GRAD, bool NUIS, int KM>
constexpr size_t main_lds_bytes() { return sizeof(double) * (2 * SCT_N + Layout<P, GRAD, NUIS, KM>::NACC * WAVE); }
// the fused-setup launch also stages the block's per-walker constants (P × NPC × 64 doubles, before the table is written) in the same allocation
// (NWV > 4, the wide block of one-round launches: room for all the other waves' sums at once — one combine round, two barriers)
template <int P, bool GRAD, bool NUIS, int KM, int NWV = WPB>
constexpr size_t fused_lds_bytes() {
    const size_t b = main_lds_bytes<P, GRAD, NUIS, KM>(), c = sizeof(double) * (size_t)P * NPC * WAVE;
    const size_t d = NWV > WPB ? sizeof(double) * (size_t)(NWV - 1) * Layout<P, GRAD, NUIS, KM>::NACC * WAVE : 0;
    return b > c ? (b > d ? b : d) : (c > d ? c : d);
}

// Every variant but the four-planet gradient kernels (and config 3's eight-wave block, below) keeps the register count it wants. (Rounds 3-4 held the
// nuisance-free single-planet RA/Dec gradient kernels to 72 VGPRs = seven waves per SIMD: −1 % then. With the warm-started loop next to the cold one the
// cap costs three moves per row — two polynomial coefficients re-fetched from SGPRs, one loop-carried copy — and e·cA recomputed; left alone the kernel
// took 79 registers, six waves per SIMD, and ran 2-3 % faster at every batch shape: 305.6 -> 299.2 µs (1e4 walkers), 88.7 -> 85.9 (2 500), 51.0 -> 49.7
// (1 250); profiles/r5_waves_ab.txt.)
template <int P, bool GRAD, bool NUIS, int KM, int NWV = WPB>
constexpr unsigned main_min_waves() {
    // Four planets with a gradient: left alone the kernel takes 312 registers (56 of them AGPRs, 8-62 moves per row) = ONE wave per SIMD, which
    // cannot cover the latency of its own dependent FP64 chains (0.21 of the FP64 peak against 0.41 for two planets). Held to 256 it parks
    // 2-4 doubles per astrometry row and ~17 per RV row in scratch memory and runs two waves: 906 -> 765 µs per step of the 4-planet probe
    // (same box, profiles/r4_p4_waves_ab.txt). Measured and NOT kept: the projection constants and K, cos ω, sin ω of every planet read from
    // an LDS copy inside the row bodies instead (72 registers fewer on paper): the volatile loads it needs cost the register allocator more
    // than they free — 11-21 scratch accesses per row instead of 4-17.
    if (P >= 4 && GRAD) return 2u;
    // The eight-wave block of a one-round launch puts two waves of every block on each SIMD: at the 84 registers config 3's kernel takes since its warm
    // rows carry their own (fourth-order) copy of the correction a CU holds two such blocks, at 80 (two dwords parked outside the loop) three —
    // 1 250 walkers x 1e4 epochs: 49.9 -> 48.9 µs per step (profiles/r5_d4_ab.txt). The four-wave kernel keeps its 84 (five waves: 280 µs against 285).
    if (NWV == 2 * WPB && P == 1 && GRAD && !NUIS && (KM & ~KM_COR) == KM_RADEC) return 6u;
#ifdef OCTO_P2_WAVES
    // experiment: two planets with nuisances, gradient — 172-186 VGPRs left alone (two waves per SIMD); three waves need <= 168
    if (P == 2 && GRAD && NUIS && !(KM & (KM_MARG | KM_ONEIL))) return (unsigned)OCTO_P2_WAVES;
#endif
    return 1u;
}

// Which k_main variants carry the warm-started row loop (octo_device.h: KWarm) next to the cold one: the single-planet fused launches
// (the O'Neil term reads E as a number; the multi-planet kernels have no registers to carry a second solution per planet).
// A wave chooses between the two loops once, from its lanes' bounds (WARM_MIN_THR): a table whose cadence is too coarse for its walkers'
// periods — any real astrometry table — runs the cold loop at no cost but the code's size.
#ifndef OCTO_WARM
#define OCTO_WARM 1
#endif
#ifndef OCTO_WARM_PLAIN
#define OCTO_WARM_PLAIN 0
#endif
template <int P, bool GRAD, bool NUIS, int KM, bool FUSED>
constexpr bool main_warm_plain() {
    // The nuisance kernels of the kind sets with sep/PA or RV rows: with the 16-dword row buffers of a second pair of prefetching loops they run out
    // of SGPRs, and the compiler then parks an in-flight prefetch tuple in VGPR lanes (tools/kernel_resources.py: scalar_load_hazards finds it).
    // Their warm loop reads its rows with plain scalar loads the compiler waits for itself (round 5 built that and found no gain on a probe whose
    // random epochs vetoed the warm loop through the table-wide bound; with the per-row test — round 6 — an RV table with offset and jitter,
    // the usual one, runs warm between its gaps). Round 6, late: with the uniform regions left unstructured the kind sets WITHOUT sep/PA rows — RA/Dec
    // [+ cor] + absolute | relative RV — have the SGPRs for the prefetching pair (no hazard: the ISA check passes) and take it: rv_gappy_nuis −2.5 %
    // (profiles/r6_noplain_ab.txt); the sets with sep/PA keep the plain loads (and the library its last 60 KB).
    return OCTO_WARM && FUSED && P == 1 && !(KM & KM_ONEIL) && NUIS && ((KM & KM_SEPPA) || ((KM & KM_RV) && OCTO_WARM_PLAIN));
}
template <int P, bool GRAD, bool NUIS, int KM, bool FUSED>
constexpr bool main_warm() {
    return OCTO_WARM && FUSED && P == 1 && !(KM & KM_ONEIL);
}

// Two and three planets (round 6): the LAST planet — the outer one in the usual order — starts from the previous row's solution. A warm/cold DIAMOND per planet ends the
// interleaving of the two solves that a kernel at two waves per SIMD lives on (round 5: −3 %, −8 %; and a diamond around the whole row body costs five scalar
// branches per row: −3 % where every row is warm). So the row body takes the warm step for that planet UNCONDITIONALLY, next to the other planet's cold solve
// in one basic block, and a rejected row — wave-uniform, decided before the row from the row's own step (slot 7) and every lane's previous 1/D against its
// bound — solves it again cold in a TRIANGLE behind that block (one branch, not taken on a warm row; the wasted warm step is 35 instructions on the few
// rows that are rejected). Config 4 as drawn (outer e ~ U(0, 0.95), a ~ 8-40 AU at a 4-day cadence): ~5 % of the wave-rows rejected, 160.9 -> 151.4 µs same
// box; where no row is rejected 150.3 µs against 149.9 for the loop without the test (profiles/r6_cfg4_dyn.txt).
#ifndef OCTO_WARM_LAST3
#define OCTO_WARM_LAST3 1
#endif
template <int P, bool GRAD, bool NUIS, int KM, bool FUSED>
constexpr bool main_warm_last() {
    // Three planets (round 6, late): the kind sets without sep/PA rows — RA/Dec [+ cor] [+ absolute | relative RV]: 204-242 VGPRs with the state of the last
    // planet, no SGPR spills; the sets with sep/PA park ~30 SGPRs with it (and the library has 0.1 MB to spare) — 343.5 -> 319.5 µs per step of the 3-planet probe
    return OCTO_WARM && FUSED && (P == 2 || (P == 3 && OCTO_WARM_LAST3 && !(KM & KM_SEPPA))) && !(KM & (KM_ONEIL | KM_MARG));
}

// the step bound as in warm_init; the lane's bound on 1/D of the previous row for the last planet. The chain starts cold (1/D = +Inf).
template <int P>
__device__ __forceinline__ float warm_last_init(WarmState<P>& ws, const PC& pc, const DevObs& ob, bool enabled) {
    float bound = 0.0f;
#pragma unroll
    for (int k = WARM_LADDER - 1; k >= 0; --k) {
        const float d = enabled ? ob.dm_ladder[k] : 0.0f;
        const bool veto = fabsf(d * (float)pc.invP) > WARM_DM_VETO;
        if (d > 0.0f && __builtin_amdgcn_ballot_w64(veto) == 0) bound = d;
    }
    ws.key_hi = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint((double)bound));
    const float dmx = fabsf(bound * (1.0f + 0x1p-18f) * (float)pc.invP);
    const float th = __builtin_amdgcn_exp2f(0.2f * (__builtin_amdgcn_logf((float)WARM_TOL) - 3.0f * __builtin_amdgcn_logf(dmx)));
    ws.thr[P - 1] = (double)th;
    ws.st[P - 1].sE = 0.0; ws.st[P - 1].cE = 1.0; ws.st[P - 1].invD = __builtin_huge_val();
    return bound;
}

// The wave's step bound: the first entry of the table's ladder (DevObs::dm_ladder, preferred first) that no lane vetoes, a veto being
// ΔM = bound/P > WARM_DM_VETO (thr would fall below WARM_MIN_THR; a NaN — an invalid walker — does not veto). thr = (tol / ΔM³)^(1/5) per lane from
// that bound (v_log_f32 / v_exp_f32 are base 2). No entry passes: the wave runs the cold loop. The first row of a wave is cold.
template <int P>
__device__ __forceinline__ float warm_init(WarmState<P>& ws, const PC (&pc)[P], const DevObs& ob, bool enabled) {
    float bound = 0.0f;
#pragma unroll
    for (int k = WARM_LADDER - 1; k >= 0; --k) {      // (last to first: the most preferred passing entry is written last)
        const float d = enabled ? ob.dm_ladder[k] : 0.0f;
        bool veto = false;
#pragma unroll
        for (int p = 0; p < P; ++p) veto = veto || (fabsf(d * (float)pc[p].invP) > WARM_DM_VETO);
        if (d > 0.0f && __builtin_amdgcn_ballot_w64(veto) == 0) bound = d;
    }
    const double bd = (double)bound;
    ws.key_hi = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(bd));
    const float bs = bound * (1.0f + 0x1p-18f);      // rows pass on the high dword of their step: up to 2^-20 above the bound
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float dmx = fabsf(bs * (float)pc[p].invP);
        const float th = __builtin_amdgcn_exp2f(0.2f * (__builtin_amdgcn_logf((float)WARM_TOL) - 3.0f * __builtin_amdgcn_logf(dmx)));
        ws.thr[p] = (double)th;
        ws.st[p].sE = 0.0; ws.st[p].cE = 1.0; ws.st[p].invD = __builtin_huge_val();
    }
    ws.row_ok = true;
    return bound;      // 0: the cold loop
}
// the row's own step against the wave's bound: slot 7 of the record (|2π Δt|, +Inf for a row that must start cold), high dwords, unsigned
template <int P, typename Row>
__device__ __forceinline__ bool warm_row_ok(const WarmState<P>& ws, const Row& r) { return (uint32_t)r.hi[7] <= ws.key_hi; }

// FUSED: the orbit constructors inside the launch — wave 0 of every block derives its tile's constants (what k_setup stores in `wc`) and
// hands them to the other waves through LDS while those fetch the sin/cos table. No k_setup launch, no `wc` round trip: one stream
// dependency less per evaluation (~3 µs of a 60 µs call at 1 250 walkers, SURVEY §8d's strong-scaling share; 17 -> 14 µs at 300 epochs);
// k_finish<FROM_WC = false> derives the handful of constants the finish needs again. Measured and NOT kept (profiles/r3_fused_ab.txt):
// the finish inside the same launch (last block of a tile, counter + write-through partials) — its tail, one block gathering 87 tasks'
// partials with the loop's register budget, is longer than the launch gap it saves, at every batch size.
// NWV: waves per block. 4 everywhere except the WIDE block of one-round launches (few walker tiles: a strong-scaled shard, a mid-size batch):
// there every block of the grid starts at the same moment and the number of waves a SIMD holds decides how well the row loop issues. Eight
// waves per block buy six waves per SIMD with the SAME number of blocks — table fills, orbit-constructor pieces, partials and k_finish work
// unchanged — where twice as many 4-wave blocks paid all of those twice (profiles/r4_chunk_sweep_1250.txt: no gain). 1 250 walkers x 1e4
// epochs: 53.25 -> 51.8 µs per step (same box, profiles/r4_wide_ab.txt); the planner (plan_key) offers it only while a wave keeps >= 32 rows.
// The finish of a ONE-TASK launch inside k_main (round 6; defined behind planet_finish below): a mid-size callback — 1 024 θ_t x 50
// epochs: a Pigeons round, a guess_starting_position chunk — is three mostly-fixed-cost kernels (k_model_fwd 11.7 µs, k_main ~8, k_finish ~9). Its
// table is one task, so after the LDS combine wave 0 of a k_main block already holds its tile's complete sums: it runs the per-walker tail itself
// (and all four waves the model's tail) — no partials, no k_finish launch. Round 3 measured a fused finish for EVERY launch and dropped it (one block
// gathering 87 tasks' partials); with a single task there is nothing to gather. Same routines on the same sums: bit-identical to the two-launch route.
template <int P, bool GRAD, bool NUIS, int KM, bool FUSED, int NWV>
constexpr bool main_fin_fused() {
    // (compiled into the three kind sets a one-table dataset can have — RA/Dec, RA/Dec + cor, RA/Dec | absolute RV: 8 KB of code per variant)
    return FUSED && P == 1 && NWV == WPB && (KM == KM_RADEC || KM == (KM_RADEC | KM_COR) || KM == (KM_RADEC | KM_RVABS));
}
template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ void fin_in_main(const EvalArgs& a, const AccArr<P, GRAD, NUIS, KM>& acc, int64_t w, int64_t wo, int wv);

// FINF: the instantiation that carries fin_in_main — a kernel of its own (launched for one-task launches only), so that the finish's registers and
// 8 KB of code are not every launch's: compiled into the standard kernel it took config 3's k_main from 84 to 93 VGPRs and cost it 0.5 %.
template <int P, bool GRAD, bool NUIS, int KM, bool FUSED = false, int NWV = WPB, bool FINF = false>
__attribute__((amdgpu_waves_per_eu(main_min_waves<P, GRAD, NUIS, KM, NWV>())))
static __global__ __launch_bounds__(64 * NWV) void k_main(EvalArgs a) {
    using L = Layout<P, GRAD, NUIS, KM>;
    static_assert(NWV == WPB || (FUSED && NWV == 2 * WPB), "k_main: four waves per block, or eight in the fused launch");
    // The hand-scheduled row prefetch (row_issue / row_wait_issue below) is used where the ISA check passes: the multi-planet variants of
    // the k_setup route (a marginalised-RV gradient and its forward pre-pass) run out of SGPRs, and the compiler spilled the prefetched
    // tuple to VGPR lanes WHILE the load was in flight (tools/kernel_resources.py: scalar_load_hazards found it) — they read their rows
    // with plain scalar loads the compiler waits for itself.
    constexpr bool ROW_PREFETCH = FUSED || P == 1;
    // the warm solve's shape (octo_device.h: kepler_solve_warm): 1 — the warm step unconditionally and a rejected row solved again cold behind it (a triangle);
    // 2 — warm or cold arm (a diamond): the eight-wave block's kernel, which held to 80 registers parks a double in the triangle's cold block — two scratch
    // round trips per rejected row, 48.7 -> 50.8 µs per step of the 1 250-walker shard (profiles/r6_tri_ab.txt)
    constexpr int WMODE = (OCTO_WARM_TRI && NWV == WPB) ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction; make it an SGPR
                                                                          // so the row pointer stays scalar (s_load, not global_load)
    const int64_t w = (int64_t)blockIdx.x * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;          // tail lanes recompute the last walker; results discarded
    int64_t wsrc = wl;                                  // the walker this tile position holds (octo_tile.h)
    if constexpr (FUSED && P <= 2) { if (a.perm) wsrc = a.perm[wl]; }
    static_assert(!FINF || main_fin_fused<P, GRAD, NUIS, KM, FUSED, NWV>(), "k_main<…, FINF>: a kind set fin_in_main is compiled for");
    const int task = a.task0 + (int)blockIdx.y;
    const Task tk = a.tasks[task];                      // wave-uniform: scalar loads
    const DevObs ob = a.obs[tk.obs];
    // this wave's slice of the task's rows
    const int r_lo = min(wv * tk.chunk, tk.nrows);
    const int r_hi = min(r_lo + tk.chunk, tk.nrows);
    const int row_first = tk.row0 + r_lo, n_rows = r_hi - r_lo;

    // LDS: [sin/cos table: SCT_N double2][combine buffer: NACC × 64 doubles]
    const SinCosTab tab = make_sincos_tab(reinterpret_cast<const double2*>(lds));
    double* const comb = lds + 2 * SCT_N;
    PC pc[P];
    if constexpr (!FUSED) {
        {
            const double2* __restrict__ g = reinterpret_cast<const double2*>(a.sctab);
            double2* t = reinterpret_cast<double2*>(lds);
            for (int i = threadIdx.x; i < SCT_N; i += WAVE * NWV) t[i] = g[i];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) load_pc(pc[p], a.wc, a.ldw, p, wl);
    } else {
        // The orbit constructors of the tile, one PIECE per wave (setup_planet_vals' four pieces: sin/cos of i, of ω, of Ω, the scalars):
        // in a one-round launch every block of the chip runs this prologue at the same moment, and as one wave's ~300-instruction
        // chain it left three of the four SIMDs idle for its duration (rounds 3: ~8.5 µs of fixed cost per launch, most of it here).
        // The pieces meet in LDS, and every wave assembles the constants it keeps in registers from them. The sin/cos table travels
        // global -> registers (issued first: one memory round trip under the pieces) -> LDS once the pieces have been read.
        static_assert(WPB == 4, "k_main<FUSED>: one setup piece per wave (waves 4-7 of a wide block have none)");
        constexpr int TPT = (SCT_N + WAVE * NWV - 1) / (WAVE * NWV);
        constexpr int NRAW = 15;      // si ci | sw cw | sO cO | sma T invP beta eob K0 mu f32a f32b
        static_assert(NRAW <= NPC, "the pieces fit in the LDS the fused launch allocates");
        double trs[TPT], trc[TPT];      // (scalars, not double2[]: the aggregate copies kept the array in scratch memory)
        {
            const double2* __restrict__ g = reinterpret_cast<const double2*>(a.sctab);
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                const int i = (int)threadIdx.x + k * WAVE * NWV;
                const double2 v = g[i < SCT_N ? i : SCT_N - 1];
                trs[k] = v.x; trc[k] = v.y;
            }
        }
        double elv[P][OCTO_N_EL];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double* el = a.elems + (int64_t)p * OCTO_N_EL * a.ld + wsrc;
#pragma unroll
            for (int k = 0; k < OCTO_N_EL; ++k) elv[p][k] = el[(int64_t)k * a.ld];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double* raw = lds + (p * NRAW) * WAVE + lane;
            if (wv == 0) { double sn, cs; setup_angle<0, true>(elv[p], a.orbit_kind[p], sn, cs); raw[0 * WAVE] = sn; raw[1 * WAVE] = cs; }
            else if (wv == 1) { double sn, cs; setup_angle<1, true>(elv[p], a.orbit_kind[p], sn, cs); raw[2 * WAVE] = sn; raw[3 * WAVE] = cs; }
            else if (wv == 2) { double sn, cs; setup_angle<2, true>(elv[p], a.orbit_kind[p], sn, cs); raw[4 * WAVE] = sn; raw[5 * WAVE] = cs; }
            else if (wv == 3) {
                const SetupScalars q = setup_scalars<true>(elv[p], a.c, a.orbit_kind[p], a.has_mass[p]);
                raw[6 * WAVE] = q.sma; raw[7 * WAVE] = q.T; raw[8 * WAVE] = q.invP; raw[9 * WAVE] = q.beta; raw[10 * WAVE] = q.eob;
                raw[11 * WAVE] = q.K0; raw[12 * WAVE] = q.mu; raw[13 * WAVE] = q.f32a; raw[14 * WAVE] = q.f32b;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double* raw = lds + (p * NRAW) * WAVE + lane;
            SetupScalars q;
            q.sma = raw[6 * WAVE]; q.T = raw[7 * WAVE]; q.invP = raw[8 * WAVE]; q.beta = raw[9 * WAVE]; q.eob = raw[10 * WAVE];
            q.K0 = raw[11 * WAVE]; q.mu = raw[12 * WAVE]; q.f32a = raw[13 * WAVE]; q.f32b = raw[14 * WAVE];
            double v[NWC];
            setup_assemble(v, elv[p], a.orbit_kind[p], raw[0 * WAVE], raw[1 * WAVE], raw[2 * WAVE], raw[3 * WAVE], raw[4 * WAVE], raw[5 * WAVE], q);
            pc[p].invP = v[WC_INVP]; pc[p].tp = v[WC_TP]; pc[p].e = v[WC_E]; pc[p].beta = v[WC_BETA]; pc[p].eob = v[WC_EOB];
            pc[p].cB = v[WC_CB]; pc[p].cG = v[WC_CG]; pc[p].cA = v[WC_CA]; pc[p].cF = v[WC_CF]; pc[p].K = v[WC_K]; pc[p].cw = v[WC_COSW];
            pc[p].sw = v[WC_SINW]; pc[p].mu = v[WC_MU]; pc[p].a = v[WC_A]; pc[p].cGb = v[WC_CGB]; pc[p].cFb = v[WC_CFB]; pc[p].cBe = v[WC_CBE];
            pc[p].cAe = v[WC_CAE];
            const float2 fa = *reinterpret_cast<const float2*>(&v[WC_F32A]);
            const float2 fb = *reinterpret_cast<const float2*>(&v[WC_F32B]);
            set_starter(pc[p], fa.x, fa.y, fb.x);
        }
        __syncthreads();                                // every wave has read the pieces: the table may overwrite them
        {
            double2* t = reinterpret_cast<double2*>(lds);
#pragma unroll
            for (int k = 0; k < TPT; ++k) { const int i = (int)threadIdx.x + k * WAVE * NWV; if (i < SCT_N) t[i] = make_double2(trs[k], trc[k]); }
        }
    }

    double acc[L::NACC];
#pragma unroll
    for (int k = 0; k < L::NACC; ++k) acc[k] = 0.0;
    LogProd lp;                                         // Π of the rows' variance terms (nuisance path only)
    __syncthreads();                                    // table filled

    const bool is_astrom = ob.kind == OCTO_ASTROM_RADEC || ob.kind == OCTO_ASTROM_SEPPA || ob.kind == OCTO_ONEIL_RADEC ||
                           ob.kind == OCTO_ONEIL_SEPPA;

    if (L::HAS_ASTROM && (!L::HAS_RV || is_astrom)) {
        const AstromCoef<P> co = astrom_coef<P, GRAD, NUIS, KM>(a.nuis, a.ld, ob.kind, ob.planet, ob.has_cor, tk.obs, pc, wsrc);
        const crow_t rows = constant_rows((NUIS ? ob.raw : ob.pre) + (int64_t)row_first * ROW_STRIDE);
        // two rows per trip through two SGPR buffers that swap roles (no copies): B is fetched while A is computed and vice versa
        auto body = [&](const RowRegs& r) {
            astrom_row<P, GRAD, NUIS, KM, true>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), row_get(r, 3), row_get(r, 4), row_get(r, 5), tab);
        };
        bool warm_loop = false, warm_unchecked = false;
        WarmState<P> ws;
        if constexpr (main_warm<P, GRAD, NUIS, KM, FUSED>()) {
            const float bound = warm_init<P>(ws, pc, ob, a.warm != 0);
            warm_loop = bound > 0.0f;
            // every row of the task within the wave's bound and no chain longer than WARM_RESTART: the loop without the per-row test (a uniform
            // cadence — config 3 — pays nothing for the gaps other tables have: the test is seven scalar instructions and two branches per row, 1 %)
            warm_unchecked = warm_loop && tk.key_max <= bound && tk.chunk <= WARM_RESTART;
        }
        bool warm_last = false;
        if constexpr (main_warm_last<P, GRAD, NUIS, KM, FUSED>()) {
            const float bound = warm_last_init<P>(ws, pc[P - 1], ob, a.warm != 0);
            warm_last = bound > 0.0f && n_rows > 0;
        }
        if (main_warm_last<P, GRAD, NUIS, KM, FUSED>() && warm_last) {
            if constexpr (main_warm_last<P, GRAD, NUIS, KM, FUSED>()) {
                // per row, wave-uniform: the row's own step within the wave's bound (slot 7) and every lane's previous 1/D below its bound — the last planet
                // starts from the previous row; otherwise (rare) it is solved again cold inside the row body.
                auto lbody = [&](const RowRegs8& r) {
                    const uint64_t over = __builtin_amdgcn_ballot_w64(ws.st[P - 1].invD >= ws.thr[P - 1]);
                    ws.row_ok = ((uint32_t)warm_row_ok<P>(ws, r) & (uint32_t)(over == 0)) != 0;
                    astrom_row<P, GRAD, NUIS, KM, true, 0, true, 2>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), row_get(r, 3), row_get(r, 4),
                                                                        row_get(r, 5), tab, &ws, row_get(r, 6));
                };
                RowRegs8 A = row_issue8(rows);
                for (int j = 0; j < n_rows; j += 2) {
                    RowRegs8 B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                    lbody(A);
                    if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                    A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                    lbody(B);
                }
                row_drain(A);
            }
        } else if constexpr (!ROW_PREFETCH) {
            for (int j = 0; j < n_rows; ++j) {
                const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
                astrom_row<P, GRAD, NUIS, KM, true>(acc, lp, pc, co, rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], tab);
            }
        } else if (main_warm<P, GRAD, NUIS, KM, FUSED>() && warm_loop) {
            if constexpr (main_warm_plain<P, GRAD, NUIS, KM, FUSED>()) {
                for (int j = 0; j < n_rows; ++j) {
                    const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
                    ws.row_ok = (uint32_t)__double2hiint(rw[7]) <= ws.key_hi;
                    astrom_row<P, GRAD, NUIS, KM, true, WMODE, true>(acc, lp, pc, co, rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], tab, &ws, rw[6]);
                }
            } else if constexpr (main_warm<P, GRAD, NUIS, KM, FUSED>()) {
                auto wbody_t = [&](const RowRegs8& r, auto checked) {
                    if constexpr (decltype(checked)::value) ws.row_ok = warm_row_ok<P>(ws, r);
                    astrom_row<P, GRAD, NUIS, KM, true, WMODE, decltype(checked)::value>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), row_get(r, 3),
                                                                                         row_get(r, 4), row_get(r, 5), tab, &ws, row_get(r, 6));
                };
                auto wloop = [&](auto checked) {
                    RowRegs8 A = row_issue8(rows);
                    for (int j = 0; j < n_rows; j += 2) {
                        RowRegs8 B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                        wbody_t(A, checked);
                        if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                        A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                        wbody_t(B, checked);
                    }
                    row_drain(A);      // the normal exit's spare prefetch sits in A: the wait names the tuple, so its registers stay allocated until it has landed
                };
                if (n_rows > 0) {
                    if (warm_unchecked) wloop(std::false_type{});
                    else wloop(std::true_type{});
                }
            }
        } else if (n_rows > 0) {
            // Two rows per trip through two SGPR buffers that swap roles (no copies): B is fetched while A is computed and vice versa. Either
            // exit leaves ONE load in flight (the spare re-read of the last row) and waits for it (row_drain). (Rounds 2-3 waited on the early
            // exit only, and with the tuple as an in/out operand: copies of the in-flight registers ahead of the wait — dead values, but
            // reads of a load in flight — while the normal exit did not wait at all.)
            RowRegs A = row_issue(rows);
            for (int j = 0; j < n_rows; j += 2) {
                RowRegs B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                body(A);
                if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                body(B);
            }
            row_drain(A);      // the normal exit's spare prefetch (in A; named, so that nothing is allocated over it before it lands)
        }
        astrom_finish_sums<P, GRAD, NUIS, KM>(acc, co);
    }
    if (L::HAS_RV && !is_astrom) {
        const RvCoef<P> co = rv_coef<P, GRAD, NUIS, KM>(a.nuis, a.ld, a.marg, a.ldw, ob.kind, ob.planet, tk.obs, pc, wsrc);      // (a.marg: the k_setup route only, where wsrc == wl)
        const crow_t rows = constant_rows((NUIS ? ob.raw : ob.pre) + (int64_t)row_first * ROW_STRIDE);
        auto body = [&](const RowRegs& r) {
            rv_row<P, GRAD, NUIS, KM, true>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), NUIS ? row_get(r, 3) : 0.0, tab);
        };
        bool warm_loop = false, warm_unchecked = false;
        WarmState<P> ws;
        if constexpr (main_warm<P, GRAD, NUIS, KM, FUSED>()) {
            const float bound = warm_init<P>(ws, pc, ob, a.warm != 0);
            warm_loop = bound > 0.0f;
            // every row of the task within the wave's bound and no chain longer than WARM_RESTART: the loop without the per-row test (a uniform
            // cadence — config 3 — pays nothing for the gaps other tables have: the test is seven scalar instructions and two branches per row, 1 %)
            warm_unchecked = warm_loop && tk.key_max <= bound && tk.chunk <= WARM_RESTART;
        }
        bool warm_last = false;
        if constexpr (main_warm_last<P, GRAD, NUIS, KM, FUSED>()) {
            const float bound = warm_last_init<P>(ws, pc[P - 1], ob, a.warm != 0);
            warm_last = bound > 0.0f && n_rows > 0;
        }
        if (main_warm_last<P, GRAD, NUIS, KM, FUSED>() && warm_last) {
            if constexpr (main_warm_last<P, GRAD, NUIS, KM, FUSED>()) {
                auto lbody = [&](const RowRegs8& r) {
                    const uint64_t over = __builtin_amdgcn_ballot_w64(ws.st[P - 1].invD >= ws.thr[P - 1]);
                    ws.row_ok = ((uint32_t)warm_row_ok<P>(ws, r) & (uint32_t)(over == 0)) != 0;
                    rv_row<P, GRAD, NUIS, KM, true, 0, true, 2>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), NUIS ? row_get(r, 3) : 0.0, tab, &ws, row_get(r, 6));
                };
                RowRegs8 A = row_issue8(rows);
                for (int j = 0; j < n_rows; j += 2) {
                    RowRegs8 B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                    lbody(A);
                    if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                    A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                    lbody(B);
                }
                row_drain(A);
            }
        } else if constexpr (!ROW_PREFETCH) {
            for (int j = 0; j < n_rows; ++j) {
                const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
                rv_row<P, GRAD, NUIS, KM, true>(acc, lp, pc, co, rw[0], rw[1], rw[2], NUIS ? rw[3] : 0.0, tab);
            }
        } else if (main_warm<P, GRAD, NUIS, KM, FUSED>() && warm_loop) {
            if constexpr (main_warm_plain<P, GRAD, NUIS, KM, FUSED>()) {
                for (int j = 0; j < n_rows; ++j) {
                    const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
                    ws.row_ok = (uint32_t)__double2hiint(rw[7]) <= ws.key_hi;
                    rv_row<P, GRAD, NUIS, KM, true, WMODE, true>(acc, lp, pc, co, rw[0], rw[1], rw[2], rw[3], tab, &ws, rw[6]);
                }
            } else if constexpr (main_warm<P, GRAD, NUIS, KM, FUSED>()) {
                auto wbody_t = [&](const RowRegs8& r, auto checked) {
                    if constexpr (decltype(checked)::value) ws.row_ok = warm_row_ok<P>(ws, r);
                    rv_row<P, GRAD, NUIS, KM, true, WMODE, decltype(checked)::value>(acc, lp, pc, co, row_get(r, 0), row_get(r, 1), row_get(r, 2), NUIS ? row_get(r, 3) : 0.0, tab,
                                                                                     &ws, row_get(r, 6));
                };
                auto wloop = [&](auto checked) {
                    RowRegs8 A = row_issue8(rows);
                    for (int j = 0; j < n_rows; j += 2) {
                        RowRegs8 B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                        wbody_t(A, checked);
                        if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                        A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                        wbody_t(B, checked);
                    }
                    row_drain(A);      // the normal exit's spare prefetch sits in A: the wait names the tuple, so its registers stay allocated until it has landed
                };
                if (n_rows > 0) {
                    if (warm_unchecked) wloop(std::false_type{});
                    else wloop(std::true_type{});
                }
            }
        } else if (n_rows > 0) {
            RowRegs A = row_issue(rows);
            for (int j = 0; j < n_rows; j += 2) {
                RowRegs B = row_wait_issue(A, rows + (int64_t)(j + 1 < n_rows ? j + 1 : j) * ROW_STRIDE);
                body(A);
                if (__builtin_expect(j + 1 >= n_rows, 0)) { row_drain(B); break; }
                A = row_wait_issue(B, rows + (int64_t)(j + 2 < n_rows ? j + 2 : j + 1) * ROW_STRIDE);
                body(B);
            }
            row_drain(A);
        }
        rv_finish_sums<P, GRAD, NUIS, KM>(acc, co, pc);
    }
    if constexpr (NUIS) {
        // Σ log|Σ_row| (astrometry), Σ log var (RV), Σ log(2π var) (marginalised RV, rv-absolute-margin.jl:179) of this wave's rows
        double lg = lp.log_value();
        if ((KM & KM_MARG) && ob.kind == OCTO_RV_ABS_MARG) lg = fma((double)n_rows, LOG2PI, lg);
        acc[L::OFF_S] += lg;
    }
    // ---- combine the block's waves through LDS in a fixed order (deterministic), one partial per (tile, task).
    // The sin/cos table is dead once every wave has left its row loop, so the waves' sums go through the WHOLE allocation (table +
    // the NACC×64 buffer behind it), as many waves per round as fit: all three at once for the narrow layouts (config 3: one round, two
    // barriers, where rounds 1-3 took turns through the buffer alone: three rounds, six barriers at the tail of every block). Wave 0
    // adds them in wave order, as before: bit-identical sums.
    constexpr int LDS_DOUBLES = (int)((FUSED ? fused_lds_bytes<P, GRAD, NUIS, KM, NWV>() : main_lds_bytes<P, GRAD, NUIS, KM>()) / sizeof(double));
    constexpr int WPR = (LDS_DOUBLES / (L::NACC * WAVE)) < (NWV - 1) ? (LDS_DOUBLES / (L::NACC * WAVE)) : (NWV - 1);      // waves per round
    static_assert(WPR >= 1, "k_main: the combine buffer holds one wave's sums");
    (void)comb;
#pragma unroll 1
    for (int q0 = 1; q0 < NWV; q0 += WPR) {
        __syncthreads();                                // table (first round) / previous round's sums no longer needed
        if (wv >= q0 && wv < q0 + WPR) {
#pragma unroll
            for (int k = 0; k < L::NACC; ++k) lds[((wv - q0) * L::NACC + k) * WAVE + lane] = acc[k];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll 1
            for (int q = q0; q < (q0 + WPR < NWV ? q0 + WPR : NWV); ++q) {
#pragma unroll
                for (int k = 0; k < L::NACC; ++k) acc[k] += lds[((q - q0) * L::NACC + k) * WAVE + lane];
            }
        }
    }
    if constexpr (FINF) {      // the launch has ONE task: this block finishes its tile itself
        fin_in_main<P, GRAD, NUIS, KM>(a, acc, w, wsrc, wv);
        return;
    }
    if (wv == 0 && w < a.W) {
        double* out = a.partials + (int64_t)task * L::NACC * a.ldw + w;
#pragma unroll
        for (int k = 0; k < L::NACC; ++k) out[(int64_t)k * a.ldw] = acc[k];
    }
}

// ------------------------------------------------------------------------------------ k_marg
// Pre-pass for marginalised-RV tables when a gradient is requested: μ̂ = −B/(2A) and A per walker.
template <int P, bool NUIS, int KM>
static __global__ __launch_bounds__(256) void k_marg(EvalArgs a) {
    using L = Layout<P, false, NUIS, KM>;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.W) return;
    if constexpr (L::HAS_MARG) {
        for (int o = 0; o < a.n_obs; ++o) {
            if (a.obs[o].kind != OCTO_RV_ABS_MARG) continue;
            double A = 0.0, B = 0.0;
            for (int t = a.obs_range[2 * o]; t < a.obs_range[2 * o + 1]; ++t) {
                const double* pt = a.partials + (int64_t)t * L::NACC * a.ldw + w;
                A += pt[(int64_t)(L::OFF_MARG + 0) * a.ldw];
                B += pt[(int64_t)(L::OFF_MARG + 1) * a.ldw];
            }
            a.marg_out[((int64_t)o * 2 + 0) * a.ldw + w] = -B / (2.0 * A);
            a.marg_out[((int64_t)o * 2 + 1) * a.ldw + w] = A;
        }
    }
}

// ------------------------------------------------------------------------------------ finish
// The per-walker tail of an evaluation, shared by k_finish (big batches: sums arrive from the task partials) and k_small
// (sums arrive from the block reduction): per-observation closed forms, then the map from the running sums to
// ∂ll/∂(a, e, i, ω, Ω, tp, M, plx, mass).
constexpr int NOBS_ACC = 11;             // S, 3 marg, 3 nuis, 4 O'Neil per observation

struct FinPC { double sma, P_d, beta, si, ci, sO, cO, sw, cw; };

template <int P, bool GRAD, bool NUIS, int KM>
constexpr int oneil_slots() { return (Layout<P, GRAD, NUIS, KM>::HAS_ONEIL && GRAD) ? P * 6 : 1; }

// v = {S, margA, margB, margC, nu0, nu1, nu2, on0..on3} summed over the observation's rows; returns its log-likelihood.
// `sma[p]`: semi-major axis of each planet (derived for a ThieleInnesOrbit). Writes the observation's g_nuis rows if `write`.
template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ double obs_finish(const DevObs* __restrict__ obs, int64_t ld, double* __restrict__ gn /* this observation's g_nuis rows at this walker */,
                                             const double* __restrict__ extra_w /* k_hgca's rows at this walker, or null */, int64_t ldw, double k_yr,
                                             int o, const double (&v)[NOBS_ACC],
                                             double cst, const double (&sma_p)[P], const double (&e_p)[P], const double (&M_p)[P], bool write,
                                             double (&oneil_g)[oneil_slots<P, GRAD, NUIS, KM>()], int n_planets_rt = P /* k_finishp: the system's planets (rows of `extra`) */) {
    using L = Layout<P, GRAD, NUIS, KM>;
    const int kind = obs[o].kind;
    double llo;
    if (L::HAS_MARG && kind == OCTO_RV_ABS_MARG) {
        // ll = −Σ log(2π var) − (−B²/(4A) + C + log A)      rv-absolute-margin.jl:179-181
        const double slog = NUIS ? v[0] : -cst;
        llo = (obs[o].n > 0) ? -slog - (-v[2] * v[2] / (4.0 * v[1]) + v[3] + log(v[1])) : 0.0;
    } else {
        // NUIS: S = Σ(log|Σ| + q) [astrom] or Σ(log var + r²/var) [rv]; cst = −n·log2π·(1 or ½)
        // !NUIS: S = Σ q, cst = Σ(−log2π·k − ½ log|Σ|)
        llo = cst - 0.5 * v[0];
    }
    if constexpr (L::HAS_ONEIL) {
        if ((kind == OCTO_ONEIL_RADEC || kind == OCTO_ONEIL_SEPPA) && obs[o].n > 0) {
            // ln_prior = 2 log(Σ|t_j| · ∛P / √(1−e²)), P = period/365.25   prior-observable.jl:96,136-139
            const int ip = obs[o].planet;
            double sma = sma_p[0], e = e_p[0], Mt = M_p[0];
#pragma unroll
            for (int p = 1; p < P; ++p) { sma = (p == ip) ? sma_p[p] : sma; e = (p == ip) ? e_p[p] : e; Mt = (p == ip) ? M_p[p] : Mt; }
            const double Pyr = k_yr * sqrt(sma * sma * sma / Mt) / 365.25;
            llo += 2.0 * log(v[7] * cbrt(Pyr) / sqrt(1.0 - e * e));
            if constexpr (GRAD) {
                const double f = 2.0 / v[7];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (p != ip) continue;
                    oneil_g[p * 6 + 0] += f * v[8]; oneil_g[p * 6 + 1] += f * v[9]; oneil_g[p * 6 + 2] += f * v[10];
                    oneil_g[p * 6 + 3] += 1.0 / sma; oneil_g[p * 6 + 4] += -1.0 / (3.0 * Mt); oneil_g[p * 6 + 5] += 2.0 * e / (1.0 - e * e);
                }
            }
        }
    }
    if constexpr (L::N_NU > 0) {
        if (write) {
            gn[0] = (kind == OCTO_RV_ABS_MARG) ? 0.0 : v[4];
            gn[(int64_t)ld] = v[5];
            gn[(int64_t)2 * ld] = v[6];      // northangle | trend coefficient (zero sums for a table without a basis column)
            if (kind == OCTO_HGCA) {      // ∂/∂(pmra, pmdec) from k_hgca
                const double* x = extra_w + (int64_t)(1 + n_planets_rt * OCTO_N_EL + o * OCTO_N_NUIS) * ldw;
                gn[0] = x[0]; gn[(int64_t)ld] = x[ldw]; gn[(int64_t)2 * ld] = 0.0;
            }
        }
    }
    return llo;
}

// Running sums of planet p -> its nine element adjoints, written to g_elems (zeros if !ok).
template <int P, bool GRAD, bool NUIS, int KM, bool FAST = false>
__device__ __forceinline__ void planet_finish(const double (&el)[OCTO_N_EL] /* the planet's element rows */, double* __restrict__ ge /* its g_elems rows, at this walker */,
                                              int64_t ld, const double* __restrict__ extra_w /* k_hgca's rows at this walker, or null */,
                                              int64_t ldw, const DevConsts& c, int orbit_kind, int has_mass, int p, const double* g,
                                              const double* og /* this planet's 6 O'Neil terms or null */, const FinPC& fp, bool ok) {
    using L = Layout<P, GRAD, NUIS, KM>;
    auto fdiv = [](double x, double y) { return FAST ? x * rcp_nr<2>(y) : x / y; };      // FAST: k_small (see setup_planet)
    const bool radvel = orbit_kind == OCTO_ORBIT_RADVEL;
    const bool ti = orbit_kind == OCTO_ORBIT_THIELE_INNES;
    const bool noplx = radvel || orbit_kind == OCTO_ORBIT_KEP;
    const double e = el[OCTO_EL_E];
    const double Mt = el[OCTO_EL_M];
    const double plx = noplx ? 1.0 : el[OCTO_EL_PLX];
    const double mass = has_mass ? el[OCTO_EL_MASS] : 0.0;
    // per-walker constants the setup already derived: no second round of sincos/sqrt in this latency-bound code
    const double sma = fp.sma;                 // the element itself, or α/plx of a ThieleInnesOrbit
    const double P_d = fp.P_d, beta = fp.beta;
    const double si = fp.si, ci = fp.ci, sO = fp.sO, cO = fp.cO, sw = fp.sw, cw = fp.cw;
    const double kappa = c.mas_per_au_per_plx;
    const double sm = plx * kappa, T = ti ? 1.0 : sma * sm;
    // unit Thiele-Innes constants of a Campbell orbit, or the [mas] constants a ThieleInnesOrbit is parameterised by
    const double A = ti ? el[OCTO_EL_TI_A] : cO * cw - sO * sw * ci, B = ti ? el[OCTO_EL_TI_B] : sO * cw + cO * sw * ci;
    const double F = ti ? el[OCTO_EL_TI_F] : -cO * sw - sO * cw * ci, G = ti ? el[OCTO_EL_TI_G] : -sO * sw + cO * cw * ci;
    double tiAb = 0, tiBb = 0, tiFb = 0, tiGb = 0;
    double ab = 0, eb = g[L::GE], ib = 0, wb = 0, Ob = 0, tpb, Mb = 0, plxb = 0, massb = 0, Pb = 0;
    double gM = g[L::GM], gT = g[L::GT];
    if constexpr (L::HAS_ONEIL) {
        eb += og[0] + og[5]; gM += og[1]; gT += og[2];
        ab += og[3]; Mb += og[4];
    }
    if (!radvel) {
        // adjoints of cB, cG, cA, cF (mas per unit X = cosE − e, Y = β sinE) from the running sums
        const double gB = g[L::U1] - e * g[L::U5], gG = beta * g[L::U2];
        const double gA = g[L::U3] - e * g[L::U6], gF = beta * g[L::U4];
        // ē: −Σ X̄ − (e/β) Σ sinE·Ȳ, with X̄ = cB r̄a + cA d̄ec, Ȳ = cG r̄a + cF d̄ec
        eb -= T * (B * g[L::U5] + A * g[L::U6]);
        eb -= fdiv(e, beta) * T * (G * g[L::U2] + F * g[L::U4]);
        const double Bb = T * gB, Gb = T * gG, Ab = T * gA, Fb = T * gF;
        tiAb = gA; tiBb = gB; tiFb = gF; tiGb = gG;
        const double Tb = ti ? 0.0 : B * gB + G * gG + A * gA + F * gF;
        ab += Tb * sm; plxb += Tb * sma * kappa;
        ib = Ab * (sO * sw * si) + Bb * (-cO * sw * si) + Fb * (sO * cw * si) + Gb * (-cO * cw * si);
        wb = Ab * (-cO * sw - sO * cw * ci) + Bb * (-sO * sw + cO * cw * ci) + Fb * (-cO * cw + sO * sw * ci) + Gb * (-sO * cw - cO * sw * ci);
        Ob = Ab * (-sO * cw - cO * sw * ci) + Bb * (cO * cw - sO * sw * ci) + Fb * (sO * sw - cO * cw * ci) + Gb * (-cO * sw - sO * cw * ci);
    }
    if constexpr (L::HAS_RV) {
        const double sieff = radvel ? 1.0 : si;
        const double Kc = TWO_PI * c.yd * c.au2m * c.sec2yr;      // K = Kc·a·sin i /(P_d·β)
        const double K = fdiv(Kc * sma * sieff, P_d * beta);
        const double Kb = g[L::GK];
        ab += fdiv(Kb * K, sma);
        if (!radvel) ib += fdiv(Kb * Kc * sma * ci, P_d * beta);
        Pb += -fdiv(Kb * K, P_d);
        eb += fdiv(Kb * K * e, beta * beta);
        wb += g[L::GW];
    }
    // M = 2π (t − tp)/P_d
    tpb = -fdiv(TWO_PI, P_d) * gM;
    Pb += -fdiv(TWO_PI, P_d * P_d) * gT;
    // P_d = k · a^{3/2} · M_tot^{−1/2}
    ab += fdiv(Pb * 1.5 * P_d, sma);
    Mb += fdiv(-0.5 * Pb * P_d, Mt);
    if constexpr (L::PL_N > L::GC) {
        const double mu = fdiv(mass * c.mjup2msol, Mt);
        if (has_mass) { massb = fdiv(g[L::GC] * c.mjup2msol, Mt); Mb += fdiv(-g[L::GC] * mu, Mt); }
    }
    double out[OCTO_N_EL] = {ab, eb, radvel ? 0.0 : ib, wb, radvel ? 0.0 : Ob, tpb, Mb, noplx ? 0.0 : plxb, massb};
    if (ti) {
        // a = α/plx with α = (√p + √m)/√2, p = u + v = ½((A+G)² + (B−F)²), m = u − v = ½((A−G)² + (B+F)²) — the reference's
        // α² = u + √(u² − v²) (src/parameterizations.jl:15-18) in a form that does not cancel near face-on. Push ā back:
        const double sAG = A + G, dAG = A - G, dBF = B - F, sBF = B + F;
        const double pp = 0.5 * (sAG * sAG + dBF * dBF), mm = 0.5 * (dAG * dAG + sBF * sBF);
        const double alphab = ab / plx;
        const double pb = alphab * 0.35355339059327376220 / sqrt(pp), mb = alphab * 0.35355339059327376220 / sqrt(mm);      // 1/(2√2 √·)
        out[OCTO_EL_TI_A] = tiAb + pb * sAG + mb * dAG;
        out[OCTO_EL_TI_B] = tiBb + pb * dBF + mb * sBF;
        out[OCTO_EL_TI_F] = tiFb - pb * dBF + mb * sBF;
        out[OCTO_EL_TI_G] = tiGb + pb * sAG - mb * dAG;
        out[OCTO_EL_PLX] = -ab * sma / plx;
    }
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) {
        const double x = extra_w ? extra_w[(int64_t)(1 + p * OCTO_N_EL + k) * ldw] : 0.0;
        ge[(int64_t)k * ld] = ok ? out[k] + x : 0.0;
    }
}

// ------------------------------------------------------------------------------------ model tail (k_finish, big-batch callbacks)
// ∇θ_t[d] = ∂prior/∂θ_t[d] + Σ_k ([i0_k = d]·Jc[2k] + [i1_k = d]·Jc[2k+1])·(ḡ[k] + ḡ[tp]·gtp[k]) for walker w — the tail of the callback
// (k_finish: model_tail; behind a k_small batch: k_model_bwd). ge / gn: the likelihood kernels' adjoints, [rows][ldg].
__device__ __forceinline__ double model_grad_row(int d, int64_t w, int n_planets, int n_nu, const octo_source* __restrict__ esrc,
                                                 const octo_source* __restrict__ nsrc, const double* __restrict__ Jc, const double* __restrict__ gtp,
                                                 const double* __restrict__ glp, int64_t ldj, const double* __restrict__ ge,
                                                 const double* __restrict__ gn, int64_t ldg) {
    double g = glp[(int64_t)d * ldj + w];
    const int n_el = n_planets * OCTO_N_EL, n_in = n_el + ((nsrc && gn) ? n_nu : 0);
#pragma unroll 1
    for (int k = 0; k < n_in; ++k) {
        const octo_source sc = k < n_el ? esrc[k] : nsrc[k - n_el];      // wave-uniform: scalar loads and scalar branches
        if (sc.kind == OCTO_SRC_CONST) continue;
        const bool h0 = sc.i0 == d, h1 = sc.kind != OCTO_SRC_THETA && sc.i1 == d;
        if (!h0 && !h1) continue;
        double gk;
        if (k < n_el) {
            gk = ge[(int64_t)k * ldg + w];
            const int ktp = k - k % OCTO_N_EL + OCTO_EL_TP;
            if (k != ktp && esrc[ktp].kind == OCTO_SRC_TPERI) gk = fma(ge[(int64_t)ktp * ldg + w], gtp[(int64_t)k * ldj + w], gk);
        } else {
            gk = gn[(int64_t)(k - n_el) * ldg + w];
        }
        if (h0) g = fma(Jc[(int64_t)(2 * k) * ldj + w], gk, g);
        if (h1) g = fma(Jc[(int64_t)(2 * k + 1) * ldj + w], gk, g);
    }
    return g;
}

// Called by every thread of a k_finish block once the tile's ll, ḡ_elems and ḡ_nuis are in memory (written by other waves of the SAME
// block: a block-scope fence + barrier make them visible). Wave g forms the gradient rows d = g, g + NG, … for its lane's walker.
__device__ __forceinline__ void model_tail_n(const EvalArgs& a, int64_t w, int grp, int n_waves, int P) {
    // ll_out, ḡ_elems, ḡ_nuis were written by OTHER waves of this block with plain global stores. A workgroup-scope release does not wait for
    // vmcnt on gfx950 (the stores are only ordered, not completed), which is enough while the waves share one CU's L1 — not in tgsplit mode, where
    // a block's waves may sit on two CUs. An explicit wait for the stores before the barrier costs nothing at the tail of a block and holds in
    // both modes (k_small's protocol does the same).
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (w >= a.W) return;
    const double lpp = a.mt_lpp[w], ll = a.ll_out[w];
    // ℓπcallback: non-finite θ_t or prior -> returned without the likelihood (logdensitymodel.jl:120-133)
    double lp = isfinite(lpp) ? lpp + ll : lpp;
    if (isnan(lp)) lp = -INFINITY;
    if (grp == 0) a.mt_lp[w] = lp;
    if (!a.mt_grad) return;
    const bool ok = isfinite(lp);
    const int D = a.mt_D, n_nu = a.g_nuis ? a.mt_n_nu : 0;
    for (int d = grp; d < D; d += n_waves) {
        const double g = model_grad_row(d, w, P, n_nu, a.mt_esrc, a.mt_nsrc, a.mt_Jc, a.mt_gtp, a.mt_glp, a.mt_ld, a.g_elems, a.g_nuis, a.ld);
        a.mt_grad[(int64_t)d * a.mt_ldo + w] = ok ? g : 0.0;
    }
}
template <int P>
__device__ __forceinline__ void model_tail(const EvalArgs& a, int64_t w, int grp, int n_waves) { model_tail_n(a, w, grp, n_waves, P); }

// fin_in_main (declared ahead of k_main): the per-walker tail of a one-task launch, by wave 0 of the k_main block that holds the tile's sums. What
// finish_tile does with one task's partials — obs_finish, validity, planet_finish, then the model's tail over the block's waves — on the same numbers.
template <int P, bool GRAD, bool NUIS, int KM>
__device__ __forceinline__ void fin_in_main(const EvalArgs& a, const AccArr<P, GRAD, NUIS, KM>& acc, int64_t w, int64_t wo, int wv) {
    using L = Layout<P, GRAD, NUIS, KM>;
    static_assert(P == 1 && !L::HAS_ONEIL && !L::HAS_MARG, "fin_in_main: single-planet kind sets without O'Neil / marginalised-RV sums");
    const bool live = w < a.W;
    if (wv == 0) {
        double v[NOBS_ACC];
#pragma unroll
        for (int k = 0; k < NOBS_ACC; ++k) v[k] = 0.0;
        v[0] = acc[L::OFF_S];
        if constexpr (L::N_NU > 0) { v[4] = acc[L::OFF_NU + 0]; v[5] = acc[L::OFF_NU + 1]; v[6] = acc[L::OFF_NU + 2]; }
        double sma_p[P] = {0.0}, e_p[P] = {0.0}, M_p[P] = {1.0};
        double og[oneil_slots<P, GRAD, NUIS, KM>()] = {0.0};
        const double ll = obs_finish<P, GRAD, NUIS, KM>(a.obs, a.ld, L::N_NU > 0 ? a.g_nuis + wo : nullptr, nullptr, a.ldw, a.c.k_yr, 0, v, a.obs_const[0],
                                                        sma_p, e_p, M_p, live, og);
        const SetupOut so = setup_planet<true>(a, 0, wo);      // the same routine k_finish<FROM_WC = false> derives its constants with
        bool ok = isfinite(ll) && so.ok;
        if (a.nuis)
            for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) ok = ok && isfinite(a.nuis[(int64_t)k * a.ld + wo]);
        if (live) {
            a.ll_out[wo] = ok ? ll : -INFINITY;
            if constexpr (GRAD && L::N_NU > 0) {
                if (!ok)
                    for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) a.g_nuis[(int64_t)k * a.ld + wo] = 0.0;
            }
            if constexpr (GRAD) {
                FinPC fp;
                fp.sma = so.v[WC_A]; fp.P_d = rcp_nr<2>(so.v[WC_INVP]); fp.beta = so.v[WC_BETA];
                fp.si = so.v[WC_SINI]; fp.ci = so.v[WC_COSI]; fp.sO = so.v[WC_SINO]; fp.cO = so.v[WC_COSO]; fp.sw = so.v[WC_SINW]; fp.cw = so.v[WC_COSW];
                planet_finish<P, GRAD, NUIS, KM, OCTO_FIN_FAST>(so.el, a.g_elems + wo, a.ld, nullptr, a.ldw, a.c, a.orbit_kind[0], a.has_mass[0], 0, &acc[L::OFF_PL], nullptr, fp, ok);
            }
        }
    }
    if (a.mt_lpp) model_tail<P>(a, live ? wo : a.W, wv, WPB);      // block-uniform (a kernel argument)
}

// ------------------------------------------------------------------------------------ finish_tile / k_finish
// The per-walker tail for one tile of 64 walkers, run by NG waves. It is a latency chain (a tile's partials were written by other CUs;
// what follows them is one wave's dependent arithmetic), so the work is spread over the waves instead of being left to wave 0:
//   * waves 0 .. NG−P−1 ("loaders"): wave g sums tasks g, g + (NG−P), … of each observation — many loads in flight;
//   * wave NG−P+p ("finisher" of planet p): derives that planet's constants (the orbit constructor again, or `wc`) WHILE the loaders
//     wait for their partials, receives the planet's combined sums and maps them to the nine element adjoints;
//   * wave 0 also receives the per-observation sums: log-likelihood, nuisance adjoints, validity.
// The loaders' sums are combined through LDS in a fixed order (deterministic). The forward-only instantiation uses the same split of
// the tasks (its finisher waves idle), so that the value it returns is bit-identical to the value returned with a gradient.
//   FROM_WC = true   after k_setup (big batches with a marginalised-RV gradient): constants and validity flags from `wc` / `valid`;
//   FROM_WC = false  after the fused k_main launch: derived again from the elements.
// LDS scratch: CH rows × NG × 64 doubles; sums are combined CH rows at a time (two barriers per chunk); row k is read by wave READER(k).
constexpr int largest_divisor(int n, int at_most) {
    int d = 1;
    for (int k = 1; k <= at_most && k <= n; ++k) d = (n % k == 0) ? k : d;
    return d;
}

template <int N, int NG, int CH, int R0, int RDIV>      // row k is summed (and kept) by wave R0 + k / RDIV
__device__ __forceinline__ void combine_rows(double (&v)[N], double* lds, int grp, int lane) {
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += CH) {
#pragma unroll
        for (int k = c0; k < (c0 + CH < N ? c0 + CH : N); ++k) lds[((k - c0) * NG + grp) * WAVE + lane] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = c0; k < (c0 + CH < N ? c0 + CH : N); ++k) {
            if (grp == R0 + k / RDIV) {
                double x = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) x += lds[((k - c0) * NG + g) * WAVE + lane];
                v[k] = x;
            }
        }
        __syncthreads();
    }
}

template <int P, bool GRAD, bool NUIS, int KM, int NG, int CH, bool FROM_WC>
__device__ __forceinline__ void finish_tile(const EvalArgs& a, int64_t tile, int grp, int lane, double* lds) {
    using L = Layout<P, GRAD, NUIS, KM>;
    constexpr int NPL = P * L::PL_N;
    constexpr int NGL = NG - P;                           // loader waves
    // tasks in flight per loader wave: OCTO_FIN_UNROLL, fewer for the widest layouts (1024 threads leave 128 VGPRs per lane; the
    // O'Neil variants spilled to scratch memory with four tasks of 20+ sums in flight)
    constexpr int UNR = (L::OFF_PL + NPL) > 24 ? 1 : (L::OFF_PL + NPL) > 16 ? (OCTO_FIN_UNROLL > 2 ? 2 : OCTO_FIN_UNROLL) : OCTO_FIN_UNROLL;
    static_assert(NGL >= 1, "finish_tile: more planets than waves");
    const int64_t w = tile * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    // wo: the walker this tile position holds (octo_tile.h; the identity without a permutation). The partials are indexed by the POSITION (wl),
    // everything that belongs to the caller — elements, nuisances, k_hgca's rows, ll, the adjoints, the model's tail — by the walker.
    int64_t wo = wl;
    if constexpr (!FROM_WC) { if (a.perm) wo = a.perm[wl]; }
    const int my_p = grp - NGL;                           // >= 0: this wave finishes planet my_p
    double gp[NPL > 0 ? NPL : 1];
#pragma unroll
    for (int k = 0; k < NPL; ++k) gp[k] = 0.0;
    double ll = 0.0;
    double oneil_g[oneil_slots<P, GRAD, NUIS, KM>()];     // per planet: ΔGE, ΔGM, ΔGT, Δā, ΔM̄tot, Δē from O'Neil terms
#pragma unroll
    for (int k = 0; k < oneil_slots<P, GRAD, NUIS, KM>(); ++k) oneil_g[k] = 0.0;
    double sma_p[P], e_p[P], M_p[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { sma_p[p] = 0.0; e_p[p] = 0.0; M_p[p] = 1.0; }
    if constexpr (L::HAS_ONEIL) {
        if (grp == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (FROM_WC) sma_p[p] = a.wc[((int64_t)p * NWC + WC_A) * a.ldw + wl];
                else sma_p[p] = setup_planet<true>(a, p, wo).v[WC_A];
                e_p[p] = a.elems[((int64_t)p * OCTO_N_EL + OCTO_EL_E) * a.ld + wo];
                M_p[p] = a.elems[((int64_t)p * OCTO_N_EL + OCTO_EL_M) * a.ld + wo];
            }
        }
    }
    // ---- finisher waves: this planet's constants — derived EARLY, while the loaders' partials are in flight, unless the layout is so
    // wide (O'Neil sums in a 1024-thread block: 128 VGPRs per lane) that 18 more live values across the load loop would spill: then
    // after the combine
    constexpr bool EARLY = true;      // (O'Neil layouts run 8-wave blocks: 256 VGPRs per lane; several planets: finish_tile_multi)
    FinPC fp = {};
    double elv[OCTO_N_EL];
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = 0.0;
    bool ok_mine = true;                                  // wave 0: ll and nuisances finite; finisher p: planet p's elements in the domain
    auto derive = [&]() {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (my_p != p) continue;
            if constexpr (FROM_WC) {
                const double* wc = a.wc + (int64_t)p * NWC * a.ldw + wl;
                fp.sma = wc[WC_A * a.ldw]; fp.P_d = rcp_nr<2>(wc[WC_INVP * a.ldw]); fp.beta = wc[WC_BETA * a.ldw];
                fp.si = wc[WC_SINI * a.ldw]; fp.ci = wc[WC_COSI * a.ldw]; fp.sO = wc[WC_SINO * a.ldw]; fp.cO = wc[WC_COSO * a.ldw];
                fp.sw = wc[WC_SINW * a.ldw]; fp.cw = wc[WC_COSW * a.ldw];
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = a.elems[((int64_t)p * OCTO_N_EL + k) * a.ld + wo];
                ok_mine = a.valid[(int64_t)p * a.ldw + wl] != 0;
            } else {
                const SetupOut so = setup_planet<true>(a, p, wo);      // the same routine, the same values k_setup would have stored
                fp.sma = so.v[WC_A]; fp.P_d = rcp_nr<2>(so.v[WC_INVP]); fp.beta = so.v[WC_BETA];
                fp.si = so.v[WC_SINI]; fp.ci = so.v[WC_COSI]; fp.sO = so.v[WC_SINO]; fp.cO = so.v[WC_COSO];
                fp.sw = so.v[WC_SINW]; fp.cw = so.v[WC_COSW];
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = so.el[k];
                ok_mine = so.ok;
            }
        }
    };
    if constexpr (GRAD && EARLY) derive();
    for (int o = 0; o < a.n_obs; ++o) {
        // the per-observation sums this layout carries (S [+ nuisance adjoints] [+ marginalised-RV sums] [+ O'Neil sums]): only those
        // rows go through the loads and the LDS combine (config 3: ONE row, not NOBS_ACC = 11)
        constexpr int NOB = L::OFF_PL;
        double vo[NOB];
#pragma unroll
        for (int k = 0; k < NOB; ++k) vo[k] = 0.0;
        const int t0 = a.obs_range[2 * o], t_end = a.obs_range[2 * o + 1];
        const double cst = a.obs_const[o];
#pragma clang loop unroll_count(UNR)
        for (int tt = t0 + grp; tt < (grp < NGL ? t_end : t0); tt += NGL) {
            const double* pt = a.partials + (int64_t)tt * L::NACC * a.ldw + wl;
            // all of the task's sums in flight, then the adds (left to itself the compiler reused one register pair for the
            // 36 loads of a 3-planet task and waited for each: 41 µs per launch instead of ~12)
            double to[NOB];
#pragma unroll
            for (int k = 0; k < NOB; ++k) to[k] = pt[(int64_t)k * a.ldw];
            double tmp[NPL > 0 ? NPL : 1];
#pragma unroll
            for (int k = 0; k < NPL; ++k) tmp[k] = pt[(int64_t)(L::OFF_PL + k) * a.ldw];
#pragma unroll
            for (int k = 0; k < NOB; ++k) vo[k] += to[k];
#pragma unroll
            for (int k = 0; k < NPL; ++k) gp[k] += tmp[k];
        }
        combine_rows<NOB, NG, (CH < NOB ? CH : NOB), 0, 1 << 20>(vo, lds, grp, lane);
        if (grp == 0) {
            double v[NOBS_ACC];      // S, margA, margB, margC, nu0, nu1, nu2, on0..on3
#pragma unroll
            for (int k = 0; k < NOBS_ACC; ++k) v[k] = 0.0;
            v[0] = vo[L::OFF_S];
            if constexpr (L::HAS_MARG) { v[1] = vo[L::OFF_MARG + 0]; v[2] = vo[L::OFF_MARG + 1]; v[3] = vo[L::OFF_MARG + 2]; }
            if constexpr (L::N_NU > 0) { v[4] = vo[L::OFF_NU + 0]; v[5] = vo[L::OFF_NU + 1]; v[6] = vo[L::OFF_NU + 2]; }
            if constexpr (L::HAS_ONEIL) {
                v[7] = vo[L::OFF_ONEIL + 0];
                if constexpr (GRAD) { v[8] = vo[L::OFF_ONEIL + 1]; v[9] = vo[L::OFF_ONEIL + 2]; v[10] = vo[L::OFF_ONEIL + 3]; }
            }
            // observations are summed in the order given (system.jl:93,186)
            ll += obs_finish<P, GRAD, NUIS, KM>(a.obs, a.ld, L::N_NU > 0 ? a.g_nuis + (int64_t)o * OCTO_N_NUIS * a.ld + wo : nullptr, a.extra ? a.extra + wo : nullptr,
                                                a.ldw, a.c.k_yr, o, v, cst, sma_p, e_p, M_p, w < a.W, oneil_g);
        }
    }
    // every planet's sums in one pass of chunks; planet p's rows are summed by its finisher wave (wave 0 without a gradient: no rows)
    // (several planets: chunks that do not straddle two planets — with rows of two readers in one chunk the compiler merged the two
    // readers' code and indexed the sums dynamically, which put them in scratch memory)
    if constexpr (NPL > 0) combine_rows<NPL, NG, (P == 1 ? (CH < NPL ? CH : NPL) : largest_divisor(L::PL_N, CH)), NGL, (L::PL_N > 0 ? L::PL_N : 1)>(gp, lds, grp, lane);
    // a finisher wave keeps its own planet's sums only (the other planets' registers are free for the chain that follows)
    double gmine[L::PL_N > 0 ? L::PL_N : 1];
#pragma unroll
    for (int k = 0; k < L::PL_N; ++k) {
        double x = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) x = (my_p == p) ? gp[p * L::PL_N + k] : x;
        gmine[k] = x;
    }
    if constexpr (GRAD && !EARLY) derive();
    if (grp == 0) {
        if (a.extra) ll += a.extra[wo];
        ok_mine = isfinite(ll);
        if constexpr (FROM_WC) {
            if constexpr (!GRAD) {
#pragma unroll
                for (int p = 0; p < P; ++p) ok_mine = ok_mine && a.valid[(int64_t)p * a.ldw + wl] != 0;
            }
        } else {
            // what k_setup records in `valid`: every planet's elements inside the domain (with a gradient: reported by the planet's
            // finisher wave), every nuisance finite
            if constexpr (!GRAD) {
#pragma unroll
                for (int p = 0; p < P; ++p) ok_mine = ok_mine && setup_planet<true>(a, p, wo).ok;
            }
            if (a.nuis)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) ok_mine = ok_mine && isfinite(a.nuis[(int64_t)k * a.ld + wo]);
        }
    }
    bool ok = ok_mine;
    if constexpr (GRAD) {
        // validity flags (and the O'Neil corrections wave 0 accumulated) cross the waves through LDS: rows 0 .. P of the scratch
        if (grp == 0 || my_p >= 0) lds[(grp == 0 ? 0 : 1 + my_p) * WAVE + lane] = ok_mine ? 1.0 : 0.0;
        if constexpr (L::HAS_ONEIL) {
            if (grp == 0) {
#pragma unroll
                for (int k = 0; k < P * 6; ++k) lds[(1 + P + k) * WAVE + lane] = oneil_g[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q <= P; ++q) ok = (q == 0 ? true : ok) && lds[q * WAVE + lane] != 0.0;
        if constexpr (L::HAS_ONEIL) {
            if (my_p >= 0) {
#pragma unroll
                for (int k = 0; k < P * 6; ++k) oneil_g[k] = lds[(1 + P + k) * WAVE + lane];
            }
        }
    }
    const bool live = w < a.W;      // (the tile's tail lanes recomputed the last walker: they store nothing)
    if (live && grp == 0) {
        a.ll_out[wo] = ok ? ll : -INFINITY;
        if constexpr (GRAD && L::N_NU > 0) {
            if (!ok)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) a.g_nuis[(int64_t)k * a.ld + wo] = 0.0;
        }
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (my_p != p || !live) continue;
            // FAST (reciprocal-multiply divisions, as in k_small): the tail of a launch is one wave's dependent chain, and an IEEE
            // FP64 division is a dozen dependent instructions
            planet_finish<P, GRAD, NUIS, KM, OCTO_FIN_FAST>(elv, a.g_elems + (int64_t)p * OCTO_N_EL * a.ld + wo, a.ld, a.extra ? a.extra + wo : nullptr, a.ldw, a.c,
                                                            a.orbit_kind[p], a.has_mass[p], p, gmine, L::HAS_ONEIL ? &oneil_g[p * 6] : nullptr, fp, ok);
        }
    }
    if (a.mt_lpp) model_tail<P>(a, live ? wo : a.W, grp, NG);      // block-uniform (a kernel argument); a dead tail lane passes W: it stores nothing
}

// ------------------------------------------------------------------------------------ finish_tile_multi (several planets)
// The same tail for P >= 2. finish_tile's loaders carry EVERY planet's sums (P·PL_N = 24-48 running sums + as many loads in flight per
// lane): the 3- and 4-planet instantiations spilled 170-418 VGPRs (round 3). Here a wave only ever holds ONE planet's PL_N sums:
//   * wave 0 ("observations"): the per-observation columns (S, nuisance adjoints, marginalised-RV and O'Neil sums) of every task, one
//     observation after the other -> obs_finish, log-likelihood, validity. No other wave touches those columns, so they need no combine.
//   * waves 1 + p·NWR + j, j < NWR ("planet p"): tasks j, j + NWR, … of ALL tasks, planet p's PL_N columns only. Wave j = 0 of the group
//     is the planet's finisher: it derives the planet's constants while its first loads are in flight, receives the other waves' sums
//     through LDS (one barrier, fixed order) and maps them to the nine element adjoints.
// The forward-only instantiation is wave 0 alone (a 64-thread block): it sums the same columns in the same order as wave 0 of the
// gradient instantiation, so the value is bit-identical with and without a gradient.
// LDS: rows 0 .. 7P of 64 doubles (validity flags, O'Neil corrections), then P·(NWR−1)·PL_N rows for the combine.
template <int P, bool GRAD, bool NUIS, int KM>
constexpr int fin_waves() {
    using L = Layout<P, GRAD, NUIS, KM>;
    if (P == 1) return L::HAS_ONEIL ? FIN_G / 2 : FIN_G;      // (the O'Neil sums next to a planet's in a 1024-thread block: 128 VGPRs per lane, 34 spilled)
    return GRAD ? 1 + P * fin_nwr(P) : 1;
}
template <int P, bool GRAD, bool NUIS, int KM>
constexpr size_t fin_lds_bytes() {
    using L = Layout<P, GRAD, NUIS, KM>;
    if (P == 1) return sizeof(double) * FIN_CH * fin_waves<P, GRAD, NUIS, KM>() * WAVE;
    return sizeof(double) * WAVE * (size_t)((1 + 7 * P) + (GRAD ? P * (fin_nwr(P) - 1) * L::PL_N : 0));
}

template <int P, bool GRAD, bool NUIS, int KM, bool FROM_WC>
__device__ __forceinline__ void finish_tile_multi(const EvalArgs& a, int64_t tile, int grp, int lane, double* lds) {
    using L = Layout<P, GRAD, NUIS, KM>;
    constexpr int NWR = fin_nwr(P);
    constexpr int PLN = L::PL_N;
    constexpr int NOB = L::OFF_PL;
    constexpr int FLAG_ROWS = 1 + 7 * P;
    constexpr int UNR_O = NOB <= 4 ? 8 : (NOB <= 8 ? 4 : 2);      // tasks in flight: wave 0 (NOB columns each) ...
    constexpr int UNR_P = 4;                                       // ... and a planet wave (PL_N columns each)
    const int64_t w = tile * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    int64_t wo = wl;      // the walker this tile position holds (octo_tile.h; as in finish_tile): the partials by position, everything of the caller's by walker
    if constexpr (!FROM_WC && P == 2) { if (a.perm) wo = a.perm[wl]; }
    const int role_p = grp == 0 ? -1 : (grp - 1) / NWR;           // the planet this wave works for
    const int sub = grp == 0 ? 0 : (grp - 1) % NWR;
    const int my_p = (grp > 0 && sub == 0) ? role_p : -1;        // >= 0: this wave finishes planet my_p
    double gp[PLN > 0 ? PLN : 1];
#pragma unroll
    for (int k = 0; k < PLN; ++k) gp[k] = 0.0;
    double ll = 0.0;
    double oneil_g[oneil_slots<P, GRAD, NUIS, KM>()];
#pragma unroll
    for (int k = 0; k < oneil_slots<P, GRAD, NUIS, KM>(); ++k) oneil_g[k] = 0.0;
    FinPC fp = {};
    double elv[OCTO_N_EL];
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = 0.0;
    bool ok_mine = true;
    if (grp == 0) {
        // ---- the observations' own sums -> log-likelihood and nuisance adjoints
        double sma_p[P], e_p[P], M_p[P];
#pragma unroll
        for (int p = 0; p < P; ++p) { sma_p[p] = 0.0; e_p[p] = 0.0; M_p[p] = 1.0; }
        if constexpr (L::HAS_ONEIL) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (FROM_WC) sma_p[p] = a.wc[((int64_t)p * NWC + WC_A) * a.ldw + wl];
                else sma_p[p] = setup_planet<true>(a, p, wo).v[WC_A];
                e_p[p] = a.elems[((int64_t)p * OCTO_N_EL + OCTO_EL_E) * a.ld + wo];
                M_p[p] = a.elems[((int64_t)p * OCTO_N_EL + OCTO_EL_M) * a.ld + wo];
            }
        }
        for (int o = 0; o < a.n_obs; ++o) {
            double vo[NOB];
#pragma unroll
            for (int k = 0; k < NOB; ++k) vo[k] = 0.0;
            const int t0 = a.obs_range[2 * o], t_end = a.obs_range[2 * o + 1];
#pragma clang loop unroll_count(UNR_O)
            for (int tt = t0; tt < t_end; ++tt) {
                const double* pt = a.partials + (int64_t)tt * L::NACC * a.ldw + wl;
                double to[NOB];
#pragma unroll
                for (int k = 0; k < NOB; ++k) to[k] = pt[(int64_t)k * a.ldw];
#pragma unroll
                for (int k = 0; k < NOB; ++k) vo[k] += to[k];
            }
            double v[NOBS_ACC];      // S, margA, margB, margC, nu0, nu1, nu2, on0..on3
#pragma unroll
            for (int k = 0; k < NOBS_ACC; ++k) v[k] = 0.0;
            v[0] = vo[L::OFF_S];
            if constexpr (L::HAS_MARG) { v[1] = vo[L::OFF_MARG + 0]; v[2] = vo[L::OFF_MARG + 1]; v[3] = vo[L::OFF_MARG + 2]; }
            if constexpr (L::N_NU > 0) { v[4] = vo[L::OFF_NU + 0]; v[5] = vo[L::OFF_NU + 1]; v[6] = vo[L::OFF_NU + 2]; }
            if constexpr (L::HAS_ONEIL) {
                v[7] = vo[L::OFF_ONEIL + 0];
                if constexpr (GRAD) { v[8] = vo[L::OFF_ONEIL + 1]; v[9] = vo[L::OFF_ONEIL + 2]; v[10] = vo[L::OFF_ONEIL + 3]; }
            }
            // observations are summed in the order given (system.jl:93,186)
            ll += obs_finish<P, GRAD, NUIS, KM>(a.obs, a.ld, L::N_NU > 0 ? a.g_nuis + (int64_t)o * OCTO_N_NUIS * a.ld + wo : nullptr, a.extra ? a.extra + wo : nullptr,
                                                a.ldw, a.c.k_yr, o, v, a.obs_const[o], sma_p, e_p, M_p, w < a.W, oneil_g);
        }
        if (a.extra) ll += a.extra[wo];
        ok_mine = isfinite(ll);
        if constexpr (FROM_WC) {
            if constexpr (!GRAD) {
#pragma unroll
                for (int p = 0; p < P; ++p) ok_mine = ok_mine && a.valid[(int64_t)p * a.ldw + wl] != 0;
            }
        } else {
            // what k_setup records in `valid`: every planet's elements inside the domain (with a gradient: reported by the planet's
            // finisher wave), every nuisance finite
            if constexpr (!GRAD) {
#pragma unroll
                for (int p = 0; p < P; ++p) ok_mine = ok_mine && setup_planet<true>(a, p, wo).ok;
            }
            if (a.nuis)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) ok_mine = ok_mine && isfinite(a.nuis[(int64_t)k * a.ld + wo]);
        }
    } else if constexpr (GRAD) {
        // ---- planet role_p's running sums over every task; the group's first wave derives the planet's constants meanwhile
        const double* pcol = a.partials + (int64_t)(L::OFF_PL + role_p * PLN) * a.ldw + wl;
        const int64_t tstride = (int64_t)L::NACC * a.ldw;
        auto load_sum = [&](int tt) {
            const double* pt = pcol + (int64_t)tt * tstride;
            double tmp[PLN];
#pragma unroll
            for (int k = 0; k < PLN; ++k) tmp[k] = pt[(int64_t)k * a.ldw];
#pragma unroll
            for (int k = 0; k < PLN; ++k) gp[k] += tmp[k];
        };
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (my_p != p) continue;
            if constexpr (FROM_WC) {
                const double* wc = a.wc + (int64_t)p * NWC * a.ldw + wl;
                fp.sma = wc[WC_A * a.ldw]; fp.P_d = rcp_nr<2>(wc[WC_INVP * a.ldw]); fp.beta = wc[WC_BETA * a.ldw];
                fp.si = wc[WC_SINI * a.ldw]; fp.ci = wc[WC_COSI * a.ldw]; fp.sO = wc[WC_SINO * a.ldw]; fp.cO = wc[WC_COSO * a.ldw];
                fp.sw = wc[WC_SINW * a.ldw]; fp.cw = wc[WC_COSW * a.ldw];
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = a.elems[((int64_t)p * OCTO_N_EL + k) * a.ld + wl];
                ok_mine = a.valid[(int64_t)p * a.ldw + wl] != 0;
            } else {
                const SetupOut so = setup_planet<true>(a, p, wo);      // the same routine, the same values k_setup would have stored
                fp.sma = so.v[WC_A]; fp.P_d = rcp_nr<2>(so.v[WC_INVP]); fp.beta = so.v[WC_BETA];
                fp.si = so.v[WC_SINI]; fp.ci = so.v[WC_COSI]; fp.sO = so.v[WC_SINO]; fp.cO = so.v[WC_COSO];
                fp.sw = so.v[WC_SINW]; fp.cw = so.v[WC_COSW];
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = so.el[k];
                ok_mine = so.ok;
            }
        }
#pragma clang loop unroll_count(UNR_P)
        for (int tt = sub; tt < a.n_tasks; tt += NWR) load_sum(tt);
    }
    bool ok = ok_mine;
    if constexpr (GRAD) {
        double* comb = lds + FLAG_ROWS * WAVE;
        if (grp > 0 && sub > 0) {
#pragma unroll
            for (int k = 0; k < PLN; ++k) comb[((role_p * (NWR - 1) + sub - 1) * PLN + k) * WAVE + lane] = gp[k];
        }
        // validity flags (and the O'Neil corrections wave 0 accumulated) cross the waves through LDS rows 0 .. 7P
        if (grp == 0 || my_p >= 0) lds[(grp == 0 ? 0 : 1 + my_p) * WAVE + lane] = ok_mine ? 1.0 : 0.0;
        if constexpr (L::HAS_ONEIL) {
            if (grp == 0) {
#pragma unroll
                for (int k = 0; k < P * 6; ++k) lds[(1 + P + k) * WAVE + lane] = oneil_g[k];
            }
        }
        __syncthreads();
        if (my_p >= 0) {
#pragma unroll
            for (int j = 1; j < NWR; ++j) {
#pragma unroll
                for (int k = 0; k < PLN; ++k) gp[k] += comb[((my_p * (NWR - 1) + j - 1) * PLN + k) * WAVE + lane];
            }
        }
#pragma unroll
        for (int q = 0; q <= P; ++q) ok = (q == 0 ? true : ok) && lds[q * WAVE + lane] != 0.0;
        if constexpr (L::HAS_ONEIL) {
            if (my_p >= 0) {
#pragma unroll
                for (int k = 0; k < P * 6; ++k) oneil_g[k] = lds[(1 + P + k) * WAVE + lane];
            }
        }
    }
    const bool live = w < a.W;
    if (live && grp == 0) {
        a.ll_out[wo] = ok ? ll : -INFINITY;
        if constexpr (GRAD && L::N_NU > 0) {
            if (!ok)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) a.g_nuis[(int64_t)k * a.ld + wo] = 0.0;
        }
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (my_p != p || !live) continue;
            planet_finish<P, GRAD, NUIS, KM, OCTO_FIN_FAST>(elv, a.g_elems + (int64_t)p * OCTO_N_EL * a.ld + wo, a.ld, a.extra ? a.extra + wo : nullptr, a.ldw, a.c,
                                                            a.orbit_kind[p], a.has_mass[p], p, gp, L::HAS_ONEIL ? &oneil_g[p * 6] : nullptr, fp, ok);
        }
    }
    if (a.mt_lpp) model_tail<P>(a, live ? wo : a.W, grp, (fin_waves<P, GRAD, NUIS, KM>()));
}

// one block per tile of 64 walkers. FROM_WC = false: after a k_main launch that derived the constants itself.
template <int P, bool GRAD, bool NUIS, int KM, bool FROM_WC = true>
static __global__ __launch_bounds__((64 * fin_waves<P, GRAD, NUIS, KM>())) void k_finish(EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (P == 1) finish_tile<P, GRAD, NUIS, KM, fin_waves<P, GRAD, NUIS, KM>(), FIN_CH, FROM_WC>(a, (int64_t)blockIdx.x, grp, lane, lds);
    else finish_tile_multi<P, GRAD, NUIS, KM, FROM_WC>(a, (int64_t)blockIdx.x, grp, lane, lds);
}

// ==================================================================================== OFTI
// ofti_linear_solve(epochs, ra, dec, σ_ra, σ_dec, cor, σ_ABFG, e, a, tp, M, plx) — src/parameterizations.jl:318-405:
// for fixed (e, a, tp, M) the sky position is linear in the Thiele-Innes constants (A, B, F, G); they are
// marginalised analytically under an isotropic Gaussian prior. Per walker: the same Kepler solve per epoch
// (:337-345), the 4x4 normal equations D'WD (13 running sums), a Cholesky solve, the marginal log-likelihood (:402).
constexpr int OFTI_NACC = 13;   // Sxx,Sxy,Syy for weights {rr, dd, rd}; Σxq, Σyq, Σxp, Σyp

struct OftiArgs {
    const double* rows;       // [n][8] {t, w_rr, w_dd, w_rd, p = w_rr·ra + w_rd·dec, q = w_dd·dec + w_rd·ra, 0, 0}
    int32_t n_rows, n_tasks, chunk, pad;
    const double* nl;         // [5][ld]: e, a, tp, M, plx
    int64_t ld, W, ldw;
    double* partials;         // [n_tasks*13][ldw]
    double* abfg; double* logml;
    const double* sctab;      // sin/cos grid, as EvalArgs::sctab
    double k_yr, lambda /* 1/σ_ABFG² */, data_quad, log_det_data_cov, log_det_prior_inv, n_log2pi;
};

#ifdef OCTO_API_TU      // launched from octo_api.hip only
static __global__ __launch_bounds__(64 * WPB) void k_ofti_main(OftiArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int task_rows = a.chunk * WPB;
    const int row0 = (int)blockIdx.y * task_rows;
    const int nrows = min(task_rows, a.n_rows - row0);
    const int r_lo = min(wv * a.chunk, nrows), r_hi = min(r_lo + a.chunk, nrows);
    // LDS: [sin/cos table][combine buffer], as in k_main
    const SinCosTab tab = make_sincos_tab(reinterpret_cast<const double2*>(lds));
    double* const comb = lds + 2 * SCT_N;
    {
        const double2* __restrict__ g = reinterpret_cast<const double2*>(a.sctab);
        double2* t = reinterpret_cast<double2*>(lds);
        for (int i = threadIdx.x; i < SCT_N; i += WAVE * WPB) t[i] = g[i];
    }
    PC pc = {};
    {
        const double e = a.nl[0 * a.ld + wl], sma = a.nl[1 * a.ld + wl], tp = a.nl[2 * a.ld + wl], Mt = a.nl[3 * a.ld + wl];
        const double P_d = a.k_yr * sqrt(sma * sma * sma / Mt);      // parameterizations.jl:322
        pc.invP = 1.0 / P_d; pc.tp = tp; pc.e = e; pc.beta = sqrt(1.0 - e * e);   // sqrt1me2, :325
        set_starter(pc, (float)e, (float)(1.0 - e), (float)(MK_K1N / (1.0 + e)));
    }
    double acc[OFTI_NACC];
#pragma unroll
    for (int k = 0; k < OFTI_NACC; ++k) acc[k] = 0.0;
    __syncthreads();                                    // table filled
    const crow_t rows = constant_rows(a.rows + (int64_t)(row0 + r_lo) * ROW_STRIDE);
    for (int j = 0; j < r_hi - r_lo; ++j) {
        const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
        const double t = rw[0], wrr = rw[1], wdd = rw[2], wrd = rw[3], pp = rw[4], qq = rw[5];
        const KSol s = kepler_solve<-1, true>(t, pc, tab);
        const double x = s.cE - pc.e, y = s.sE * pc.beta;            // :343-345
        const double xx = x * x, xy = x * y, yy = y * y;
        acc[0] = fma(wrr, xx, acc[0]); acc[1] = fma(wrr, xy, acc[1]); acc[2] = fma(wrr, yy, acc[2]);
        acc[3] = fma(wdd, xx, acc[3]); acc[4] = fma(wdd, xy, acc[4]); acc[5] = fma(wdd, yy, acc[5]);
        acc[6] = fma(wrd, xx, acc[6]); acc[7] = fma(wrd, xy, acc[7]); acc[8] = fma(wrd, yy, acc[8]);
        acc[9] = fma(x, qq, acc[9]); acc[10] = fma(y, qq, acc[10]);
        acc[11] = fma(x, pp, acc[11]); acc[12] = fma(y, pp, acc[12]);
    }
#pragma unroll 1
    for (int q = 1; q < WPB; ++q) {
        if (wv == q) {
#pragma unroll
            for (int k = 0; k < OFTI_NACC; ++k) comb[k * WAVE + lane] = acc[k];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int k = 0; k < OFTI_NACC; ++k) acc[k] += comb[k * WAVE + lane];
        }
        __syncthreads();
    }
    if (wv == 0 && w < a.W) {
        double* out = a.partials + (int64_t)blockIdx.y * OFTI_NACC * a.ldw + w;
#pragma unroll
        for (int k = 0; k < OFTI_NACC; ++k) out[(int64_t)k * a.ldw] = acc[k];
    }
}

static __global__ __launch_bounds__(64) void k_ofti_finish(OftiArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.W) return;
    double v[OFTI_NACC];
#pragma unroll
    for (int k = 0; k < OFTI_NACC; ++k) v[k] = 0.0;
#pragma unroll 4
    for (int t = 0; t < a.n_tasks; ++t) {      // independent loads: unrolled so that several tasks' partials are in flight
        const double* pt = a.partials + (int64_t)t * OFTI_NACC * a.ldw + w;
#pragma unroll
        for (int k = 0; k < OFTI_NACC; ++k) v[k] += pt[(int64_t)k * a.ldw];
    }
    const double e = a.nl[0 * a.ld + w], sma = a.nl[1 * a.ld + w], tp = a.nl[2 * a.ld + w], Mt = a.nl[3 * a.ld + w], plx = a.nl[4 * a.ld + w];
    const bool ok = isfinite(e) && isfinite(sma) && isfinite(tp) && isfinite(Mt) && isfinite(plx) && e >= 0.0 && e < 1.0 && sma > 0.0 && Mt > 0.0;
    // Σ_post⁻¹ = D'WD + Λ  (order A, B, F, G)                       parameterizations.jl:370-374
    const double lam = a.lambda;
    double S[4][4];
    S[0][0] = v[3] + lam; S[0][1] = v[6]; S[0][2] = v[4]; S[0][3] = v[7];
    S[1][1] = v[0] + lam; S[1][2] = v[7]; S[1][3] = v[1];
    S[2][2] = v[5] + lam; S[2][3] = v[8];
    S[3][3] = v[2] + lam;
    const double b[4] = {v[9], v[11], v[10], v[12]};
    // Cholesky S = L L'
    double L[4][4];
    bool pd = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = S[j][i];
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= L[i][k] * L[j][k];
            if (i == j) { pd = pd && (sum > 0.0); L[i][i] = sqrt(sum); }
            else L[i][j] = sum / L[j][j];
        }
    }
    double z[4], mu[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double sum = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= L[i][k] * z[k];
        z[i] = sum / L[i][i];
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        double sum = z[i];
#pragma unroll
        for (int k = i + 1; k < 4; ++k) sum -= L[k][i] * mu[k];
        mu[i] = sum / L[i][i];
    }
    const double post_quad = z[0] * z[0] + z[1] * z[1] + z[2] * z[2] + z[3] * z[3];   // μ'Σ⁻¹μ = |L⁻¹b|²
    const double log_det_post_inv = 2.0 * (log(L[0][0]) + log(L[1][1]) + log(L[2][2]) + log(L[3][3]));
    // :402
    const double lm = -0.5 * (a.data_quad - post_quad + log_det_post_inv - a.log_det_prior_inv + a.log_det_data_cov) - a.n_log2pi;
    const bool fin = ok && pd && isfinite(lm);
    a.logml[w] = fin ? lm : -INFINITY;
    if (a.abfg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.abfg[(int64_t)k * a.ld + w] = fin ? mu[k] : NAN;
    }
}

#endif      // OCTO_API_TU

}  // namespace octo
