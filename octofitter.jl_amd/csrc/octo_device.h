// octo_device.h — device-side math of the epoch-loop likelihood path (gfx950, FP64 VALU).
//
// Lane = walker. Everything that depends only on the walker lives in VGPRs for the whole row
// loop; everything that depends only on the observation row is wave-uniform and arrives through
// scalar loads (SGPRs), so a row costs no vector-memory traffic at all.
//
// Kepler solve. The reference's solver is PlanetOrbits.jl `kepler_solver(MA, e, Markley())` (call site
// src/parameterizations.jl:340; provenance docs/src/kepler.md:15-19): Markley's cubic starter E1 followed by
// ONE fifth-order correction δ5 — non-iterative, hence divergence-free across the 64 lanes. The root of
// Kepler's equation is unique, so any E1 inside the correction's basin gives the same E to rounding. That is
// what makes a CDNA4-shaped implementation legal:
//   * the starter is evaluated in FP32 (2x the FP64 issue rate, single-instruction v_sqrt/v_rcp/v_log/v_exp):
//     its own error (~4e-4 by construction) dwarfs FP32 rounding;
//   * sin/cos(E1): in k_main / k_ofti_main from a 1041-entry table in LDS plus a rotation by the remainder (sincos_table
//     below: 13 FP64 instructions); in k_small and k_hgca from one half-angle polynomial pair on |E1|/2 <= 1.59
//     (sincos_halfangle: 24 instructions, no table fill). Neither needs range reduction, quadrant selects or a
//     Payne-Hanek slow path: |E1| <= π by construction;
//   * the three divisions of the correction share ONE v_rcp_f64 (2^-23): δ3 uses it as it is, δ4 and δ5 are one correction step each
//     on the quotient, δ' = δ − r·(den·δ + f0) (round 3; rounds 1-2 refined the reciprocal instead: 3 instructions per division);
//   * sin/cos(E) follow from (sin E1, cos E1) by a rotation through δ5 (|δ5| < 5e-4: Taylor to δ^6).
// tools/kepler_proto.py measures this scheme against an 80-bit Newton solve over 2e6 (M, e) pairs incl.
// e -> 1 − 1e-9 and |M| -> 0, π: residual-weighted error 5.1e-16 max vs 7.1e-16 for the all-FP64 reference
// algorithm. tests/test_gpu_parity.py::test_kepler_* check the device code itself.
//
// Projection: the Thiele-Innes form the reference itself uses in ofti_linear_solve
// (src/parameterizations.jl:343-353): X = cos E − e, Y = √(1−e²) sin E, ra = cB X + cG Y,
// dec = cA X + cF Y — algebraically identical to orbitsolve's 2·atan(ν_fact·tan(E/2)) route.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace octo {

constexpr int WAVE = 64;   // gfx950 wavefront
constexpr double PI = 3.141592653589793238462643383279502884;
constexpr double TWO_PI = 6.283185307179586476925286766559005768;
constexpr double PI2 = PI * PI;
constexpr double LOG2PI = 1.8378770664093454835606594728112;
constexpr double MK_K0 = 3.0 * PI2 / (PI2 - 6.0);   // Markley eq. (20): α = K0 + k1(e)·(π − |M|)
constexpr double MK_K1N = 8.0 * PI / (5.0 * (PI2 - 6.0));

// Per-(walker, planet) derived constants, produced once per evaluation by k_setup.
enum : int {
    WC_INVP = 0,  // 1 / P_d               [orbits per day]
    WC_TP,        // epoch of periastron   [MJD]
    WC_E,         // eccentricity
    WC_BETA,      // √(1−e²)
    WC_EOB,       // e / √(1−e²)
    WC_CB, WC_CG, WC_CA, WC_CF,   // mas per unit (X, Y):  ra = CB·X + CG·Y,  dec = CA·X + CF·Y
    WC_K,         // RV semi-amplitude     [m/s]
    WC_COSW, WC_SINW,
    WC_MU,        // m_planet / M_tot  (0 when the planet declares no mass)
    WC_A,         // semi-major axis       [AU]
    WC_F32A,      // packed {float e, float 1−e}   (starter constants, FP32)
    WC_F32B,      // packed {float k1 = MK_K1N/(1+e), float 0}
    WC_CGB, WC_CFB,   // CG·β, CF·β          ra = CB·cosE + CGB·sinE − CBE,  dec = CA·cosE + CFB·sinE − CAE
    WC_CBE, WC_CAE,   // CB·e, CA·e
    WC_SINI, WC_COSI, WC_SINO, WC_COSO,   // of the reduced angles (i mod π, Ω mod 2π); read by k_finish only
    NWC
};

struct PC {   // one planet's constants for one walker, in registers
    double invP, tp, e, beta, eob, cB, cG, cA, cF, K, cw, sw, mu, a;
    double cGb, cFb, cBe, cAe;
    float ef, omef, k1f;
    // starter constants derived from (ef, omef, k1f) once per walker (set_starter): every multiple the Markley starter
    // needs is a ready operand of an FMA, so the row loop carries no FP32 scaling instructions
    float A0, A1;      // α = K0 + k1·(π − |M|) = A0 + A1·|frac|,  A0 = K0 + k1·π,  A1 = −2π·k1   (M = 2π·frac)
    float o2, no3;     // 2(1−e), −3(1−e)
    double he;         // e/2
};

// PIN: after the loop-invariant constants are computed, an empty asm makes them opaque to the compiler, which otherwise
// rematerialises cheap ones (3·(1−e), e/2) inside the row loop to save a register.
template <bool PIN = true>
__device__ __forceinline__ void set_starter(PC& pc, float ef, float omef, float k1f) {
    pc.ef = ef; pc.omef = omef; pc.k1f = k1f;
    pc.A0 = fmaf(k1f, (float)PI, (float)MK_K0); pc.A1 = -(float)TWO_PI * k1f;
    pc.o2 = 2.0f * omef; pc.no3 = -3.0f * omef;
    pc.he = 0.5 * pc.e;
    if constexpr (PIN) {
        asm("" : "+v"(pc.A0), "+v"(pc.A1), "+v"(pc.o2), "+v"(pc.no3));
        asm("" : "+v"(pc.he));
    }
}

struct KSol { double sE, cE, invD, dt, E; };

// WARM START from the previous row (k_main, round 5). Rows of a table are epoch-sorted (relative-astrometry.jl:46-47, rv-absolute.jl:98-99) and a
// wave walks a contiguous slice of them for the same 64 walkers, so the previous row's (sin E, cos E, 1/(1 − e cos E)) IS a starter: with
// ΔM = 2π Δt / P and x = ΔM / D,   dE = x − ½ (e sin E / D) x²   (second-order Taylor of E(M)),   (sin, cos)(E + dE) by a rotation through dE,
// f0 = E1 − e sin E1 − M = dE − ΔM − e (sin E1 − sin E)   — no E, no M as numbers, no FP32 Markley starter, no table lookup — and then the SAME
// fifth-order correction. The predictor's error is <= x³/D² (third derivative of E(M): (3e² sin²E − e cos E · D)/D⁵), and the correction is as
// good from there as from Markley's starter while that stays below ~5e-4 (tools/kepler_warm_proto.py; a sample of 4e6 (e, E, ΔM) incl.
// e -> 1 − 1e-9: D-weighted error of (sin E, cos E) 3.9e-16 below 1e-3, 2.2e-16 below 1e-4). Where it does not (a fast orbit near the periastron
// of a high-e walker) the wave falls back to the Markley starter for that row — WAVE-UNIFORMLY (one ballot, one scalar branch: the loop stays
// divergence-free), and the root is unique, so the result is the same E to rounding either way.
// The test is a priori and one compare: x³/D² < tol  <=>  1/D < thr with thr = (tol / ΔM³)^(1/5) per lane, from the TABLE's largest
// 2π Δt (DevObs::dm_max; exact for a uniform cadence, conservative otherwise). A wave starts warm only where every lane's thr >= WARM_MIN_THR, i.e.
// ΔM_max <= (tol/32)^(1/3) = 0.0315, so a pass bounds |x| = ΔM/D < ΔM_max · thr = tol^(1/5) ΔM_max^(2/5) <= 0.063 and |dE| < 0.066:
// the rotation's polynomials (sin to dE⁷, cos to dE⁸) are exact to 1e-16 there. tol = 1e-3 with the FOURTH-order correction of the warm rows
// (kepler_correct): its truncation is ~2e-16 there (tools/kepler_warm_proto.py: D-weighted maximum 1.1e-15 at 4e-4 and at 1e-3 — the rounding
// level of the cold solve — 3.4e-15 at 2e-3); config 3 falls back on 8 % of its wave-rows at 1e-3, on 12.5 % at 4e-4.
// What the chain gives up: E and M never appear, so the solve of row j starts from the SOLUTION of row j−1 and its rounding (~1e-16 in M per
// row) accumulates until the next cold row — at most a wave's chunk of rows (tens to a few hundred: < 1e-13 in M, the size of the
// rounding of (t − tp)/P itself for a walker a few orbits from tp; the prototype measures 8e-15 after 72 rows).
// ROUND 6 — the bound is per WAVE and the veto per ROW. Round 5 took ΔM_max from the table's single largest step: one seasonal gap in an RV table
// (or the 120-day steps of a real astrometry table) sent every wave to the cold loop, and so did one short-period walker in a tile of 64. Now the
// dataset carries a short LADDER of candidate step bounds per table (DevObs::dm_ladder: quantiles of |2π Δt|, preferred first), a wave takes the
// first one none of its lanes vetoes (ΔM <= WARM_DM_VETO for every lane), thr follows from THAT bound, and a row whose own step exceeds it is
// solved cold behind a SCALAR branch: slot 7 of the row record holds |2π Δt| (+Inf for rows that must start cold: row 0, every
// WARM_RESTART-th row — which bounds the chain whatever the planner's chunk is, ADVICE r5 — and non-finite steps), and the test is an unsigned
// compare of its high dword with the bound's (both SGPRs; positive doubles order like their high dwords, and thr is computed from the
// bound stretched by 2^-18 to cover the low dword). The loop stays divergence-free; a gap costs one cold row.
struct KWarm { double sE, cE, invD; };
constexpr double WARM_TOL = 1.0e-3;
constexpr double WARM_MIN_THR = 2.0;      // a wave takes the warm loop only if every lane passes at least wherever D >= 1/2
constexpr float WARM_DM_VETO = 0.0314f;   // (WARM_TOL/32)^(1/3) = 0.03150, less the 2^-18 stretch and the float roundings: thr >= WARM_MIN_THR
constexpr int WARM_LADDER = 8;            // candidate step bounds per table (DevObs::dm_ladder)
constexpr int WARM_RESTART = 256;         // every WARM_RESTART-th row of a table starts cold

__device__ __forceinline__ void load_pc(PC& pc, const double* __restrict__ wc, int64_t ldw, int p, int64_t w) {
    const double* b = wc + (int64_t)p * NWC * ldw + w;
    pc.invP = b[WC_INVP * ldw]; pc.tp = b[WC_TP * ldw]; pc.e = b[WC_E * ldw]; pc.beta = b[WC_BETA * ldw];
    pc.eob = b[WC_EOB * ldw]; pc.cB = b[WC_CB * ldw]; pc.cG = b[WC_CG * ldw]; pc.cA = b[WC_CA * ldw];
    pc.cF = b[WC_CF * ldw]; pc.K = b[WC_K * ldw]; pc.cw = b[WC_COSW * ldw]; pc.sw = b[WC_SINW * ldw];
    pc.mu = b[WC_MU * ldw]; pc.a = b[WC_A * ldw];
    pc.cGb = b[WC_CGB * ldw]; pc.cFb = b[WC_CFB * ldw]; pc.cBe = b[WC_CBE * ldw]; pc.cAe = b[WC_CAE * ldw];
    const float2 fa = *reinterpret_cast<const float2*>(&b[WC_F32A * ldw]);
    const float2 fb = *reinterpret_cast<const float2*>(&b[WC_F32B * ldw]);
    set_starter(pc, fa.x, fa.y, fb.x);
}

__device__ __forceinline__ double pack_f32x2(float x, float y) {
    float2 v = make_float2(x, y);
    return *reinterpret_cast<double*>(&v);
}

// Polynomial / Taylor coefficients live in constant memory on purpose: a wave-uniform load puts them in SGPRs
// once per wave, and every Horner step becomes one VOP3 `v_fma_f64 v, v, v, s[k]`. As literals hipcc keeps
// copies in VGPRs and re-materialises the addend with a v_mov_b64 in front of a VOP2 v_fmac per step
// (+15 VALU instructions per row, measured in the r1 ISA dump).
__constant__ double OCTO_KT[24] = {
    // sin(h)/h = 1 + u·Q(u), Q degree 7 (Chebyshev fit on |h| <= 1.59, tools/gen_sincos_poly.py; error 1.5e-17)
    2.7193985903359714e-15, -7.642822688610201e-13, 1.6058933154149512e-10, -2.5052106738943426e-08,
    2.7557319209590407e-06, -0.0001984126984119912, 0.00833333333333316, -0.16666666666666666,
    // cos(h) = 1 + u·Qc(u), Qc degree 8 (error 8.7e-18)
    -1.51077353714095e-16, 4.776743872294959e-14, -1.1470664505957302e-11, 2.0876755532557785e-09,
    -2.7557319207812853e-07, 2.480158730147785e-05, -0.0013888888888888464, 0.04166666666666666, -0.5,
    // Taylor: sin δ = δ(1 + δ²(−1/6 + δ²/120)), cos δ − 1 = δ²(−1/2 + δ²(1/24 − δ²/720))
    1.0 / 120.0, -1.0 / 6.0, -1.0 / 720.0, 1.0 / 24.0,
    // warm-start rotation (|dE| <= 0.117): sin dE/dE − 1 = u(−1/6 + u(1/120 + u(−1/5040 + u/362880))), cos dE − 1 = u(−1/2 + u(1/24 + u(−1/720 + u/40320)))
    1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 40320.0};

// sin and cos of x for |x| <= 3.18 via the half angle: h = x/2, polynomials in u = h², then
// sin x = 2 s c, cos x = 1 − 2 s². 24 FP64 instructions, branch-free, no range reduction.
__device__ __forceinline__ void sincos_halfangle(double x, double& s, double& c) {
    const double* __restrict__ K = OCTO_KT;
    const double h = 0.5 * x;
    const double u = h * h;
    double ps = fma(K[0], u, K[1]);
    ps = fma(ps, u, K[2]);
    ps = fma(ps, u, K[3]);
    ps = fma(ps, u, K[4]);
    ps = fma(ps, u, K[5]);
    ps = fma(ps, u, K[6]);
    ps = fma(ps, u, K[7]);
    double pc = fma(K[8], u, K[9]);
    pc = fma(pc, u, K[10]);
    pc = fma(pc, u, K[11]);
    pc = fma(pc, u, K[12]);
    pc = fma(pc, u, K[13]);
    pc = fma(pc, u, K[14]);
    pc = fma(pc, u, K[15]);
    pc = fma(pc, u, K[16]);
    const double sh = fma(h * u, ps, h);
    const double ch = fma(u, pc, 1.0);
    const double t = sh + sh;
    s = t * ch;
    c = fma(-t, sh, 1.0);
}

// sin and cos of the FP32-valued starter x through a table in LDS: x = k·SCT_STEP + r with k = round(x/SCT_STEP), the grid
// point's (sin, cos) read as one ds_read_b128, and a rotation by r (|r| <= 3.1e-3: sin r to r⁵, cos r − 1 to r⁴, error
// < 1e-17). 12 FP64 instructions + 5 cheap ones instead of 24: the hot loop is FP64-issue bound and LDS is otherwise idle.
//   * SCT_STEP is 2π/1024 rounded to 13 significant bits, so k·SCT_STEP and the remainder r = x − k·SCT_STEP are EXACT in
//     FP32 (r is a multiple of ulp(x) below 2^24 of them): one v_fma_f32 and one conversion give r, no FP64 subtraction;
//     the host fills the table with the sin/cos of exactly those grid points (octo_ctx_create).
//   * k comes from the magic-number trick: x/STEP + 1.5·2^23 leaves k in the low mantissa bits (one v_fma_f32 instead of
//     mul + rndne + cvt), and those bits shifted left by 4, plus a constant, ARE the byte offset of the table entry.
// The index is NOT clamped (a clamp costs two more instructions per row): for a valid walker (0 <= e < 1) the starter lies
// within ±(π + 1e-3), inside the guard entries; an invalid walker (e >= 1, non-finite elements) may produce any offset,
// which reads garbage from the block's LDS or, beyond the allocation, zero (out-of-range DS reads return 0 and never
// fault) — k_finish discards that walker's sums.
constexpr int SCT_HALF = 512;                         // grid steps per π
constexpr int SCT_PAD = 8;                            // guard steps beyond ±π (the FP32 starter may land a hair outside)
constexpr int SCT_N = 2 * (SCT_HALF + SCT_PAD) + 1;   // 1041 entries, 16.3 KB
constexpr double SCT_STEP = 0x1.922p-8;               // 0.00613594…; 2π/1024 = 0.00613592…
constexpr float SCT_STEP_F = 0x1.922p-8f;
constexpr float SCT_INV_STEP_F = 162.9742f;
constexpr float SCT_MAGIC_F = 12582912.0f;            // 1.5·2^23 = 0x4B400000

// c5 = 1/120 is carried in a VGPR by the caller: both coefficients of the inner Horner step as SGPRs would exceed the
// one-scalar-operand limit of a VOP3 and cost a v_mov_b64 per row.
struct SinCosTab {
    const double2* tab;
    double c5;
    float inv_step;         // in a VGPR: v_fmaak_f32 takes the magic number as its literal, and a literal excludes a scalar operand
    uint32_t off0;          // in an SGPR: the third operand of v_lshl_add_u32
};

__device__ __forceinline__ SinCosTab make_sincos_tab(const double2* lds_tab) {
    SinCosTab t{lds_tab, OCTO_KT[17], SCT_INV_STEP_F, (uint32_t)(16u * (SCT_HALF + SCT_PAD) - (0x4B400000u << 4))};
    asm("" : "+v"(t.c5));      // not volatile: a volatile asm counts as a memory clobber and demotes the s_loads of the rows
    asm("" : "+v"(t.inv_step));
    asm("" : "+s"(t.off0));
    return t;
}

__device__ __forceinline__ void sincos_table(float xf, const SinCosTab& T, double& s, double& c) {
    const float km = fmaf(xf, T.inv_step, SCT_MAGIC_F);
    const float kf = km - SCT_MAGIC_F;                                   // k, exact
    const uint32_t off = (__float_as_uint(km) << 4) + T.off0;
    const double2 g = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(T.tab) + off);
    const double r = (double)fmaf(-kf, SCT_STEP_F, xf);                   // exact
    const double r2 = r * r;
    const double sr = r * fma(r2, fma(r2, T.c5, OCTO_KT[18]), 1.0);
    const double cm1 = r2 * fma(r2, OCTO_KT[20], -0.5);
    // sin(kΔ + r) = sin kΔ·cos r + cos kΔ·sin r, as two chained FMAs (the small products enter last-but-one: <= 1 ulp on the result,
    // and one instruction less per output than mul + fma + add)
    s = fma(g.x, cm1, fma(g.y, sr, g.x));
    c = fma(g.y, cm1, fma(-g.x, sr, g.y));
}

// v_rcp_f64 (≈2^-23) + NR Newton steps (each doubles the correct bits): NR = 1 -> 2^-46, NR = 2 -> full.
template <int NR>
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int k = 0; k < NR; ++k) r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1/√x: v_rsq_f64 (≈2^-23) + two Newton steps y ← y(1.5 − 0.5·x·y²).
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
#pragma unroll
    for (int k = 0; k < 2; ++k) y = y * fma(-hx * y, y, 1.5);
    return y;
}

// atan(t)/t as a polynomial in u = t² on |t| <= tan(π/8) (Chebyshev fit at 60 digits, tools/gen_sincos_poly.py's fitter;
// error 2.5e-18), highest power first. In constant memory for the same reason as OCTO_KT.
__constant__ double OCTO_AT[12] = {
    -0.017802395576940664, 0.03796254872615211, -0.0503499712321774, 0.058468552036934525, -0.06662948677808485,
    0.07692045058151757, -0.09090896793876997, 0.11111110744394355, -0.1428571427923969, 0.19999999999940782,
    -0.3333333333333312, 1.0};

// atan2(y, x) for finite arguments, not both zero: octant reduction to t = min/max, one more reduction to
// |t'| <= tan(π/8) through (t − 1)/(t + 1) with ONE division for both, the polynomial above, and the octant fix-ups.
// ~45 instructions instead of ocml's 105 (which also handles ±Inf/NaN/±0 arguments the position angle of a sky offset
// cannot have). Absolute error < 2e-16.
__device__ __forceinline__ double atan2_fast(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool big = mn > 0.41421356237309503 * mx;          // t > tan(π/8)
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    const double t = num * rcp_nr<2>(den);
    const double u = t * t;
    double p = OCTO_AT[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) p = fma(p, u, OCTO_AT[k]);
    double r = fma(t, p, big ? 0.78539816339744830962 : 0.0);   // atan(min/max) in [0, π/4]
    r = ay > ax ? 1.57079632679489661923 - r : r;
    r = x < 0.0 ? PI - r : r;
    return copysign(r, y);
}

// Julia's `x % 2π` (truncated remainder, sign of x) for |x| < 2^50: q = trunc(x/2π), r = x − q·fl(2π) by one FMA, which
// is the exact remainder whenever q is the right integer; a q that is off by one (x within an ulp of a multiple) is
// repaired by one add. Replaces ocml fmod on the sep/PA path (relative-astrometry.jl:196).
__device__ __forceinline__ double rem_2pi_trunc(double x) {
    const double q = trunc(x * (1.0 / TWO_PI));
    double r = fma(-q, TWO_PI, x);
    r = (x >= 0.0 && r < 0.0) ? r + TWO_PI : r;
    r = (x < 0.0 && r > 0.0) ? r - TWO_PI : r;
    r = (r >= TWO_PI) ? r - TWO_PI : r;
    r = (r <= -TWO_PI) ? r + TWO_PI : r;
    return r;
}

// The value lane `src_lane` holds (src_lane wave-uniform), in every lane.
__device__ __forceinline__ double lane_value(double x, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}

// Σ_rows log(x_row) as log(Π x_row): one multiply and a mantissa/exponent split per row instead of one FP64 log (≈45
// instructions) per row; one real log per wave at the end. The running mantissa stays in [0.5, 1), so nothing
// over- or underflows whatever the σ's; 0, Inf and NaN factors propagate to the final log as they would through a sum
// of logs. Rounding: N multiplies ≈ N·2^-53 relative on the product = N·1e-16 ABSOLUTE on the sum (1e-12 for 1e4 rows),
// tighter than summing N rounded logs.
struct LogProd {
    double m = 1.0;
    int e = 0;
    __device__ __forceinline__ void mul(double x) {
        m *= x;
        e += __builtin_amdgcn_frexp_exp(m);
        m = __builtin_amdgcn_frexp_mant(m);
    }
    __device__ __forceinline__ double log_value() const { return fma((double)e, 0.69314718055994530942, log(m)); }
};

// Markley (1995) starter, eqs (20),(5),(9),(10),(14),(15), in FP32; frac = M/2π in [−1/2, 1/2]
__device__ __forceinline__ float markley_starter_f32(double frac, const PC& pc) {
    const float ff = (float)frac;
    const float Mf = ff * (float)TWO_PI;              // scale in FP32: one FP64 multiply less per row
    const float alpha = fmaf(pc.A1, fabsf(ff), pc.A0);
    const float d = fmaf(alpha, pc.ef, -pc.no3);      // 3(1−e) + αe
    const float ad = alpha * d;
    const float M2 = Mf * Mf;
    const float q = fmaf(ad, pc.o2, -M2);             // 2αd(1−e) − M²
    const float r = Mf * fmaf(ad, fmaf(d, 3.0f, pc.no3), M2);       // 3αd(d − 1 + e)M + M³
    const float q2 = q * q;
    const float disc = fmaf(q2, q, r * r);            // > 0 for 0 <= e < 1: r² dominates wherever q < 0 (q >= −M², r² >= 9α⁶M²)
    const float x = fabsf(r) + __builtin_amdgcn_sqrtf(disc);
    const float w = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (2.0f / 3.0f));   // cbrt(x²)
    const float den = fmaf(w, w + q, q2);
    return fmaf(Mf, den, 2.0f * r * w) * __builtin_amdgcn_rcpf(den * d);     // (2rw/den + M)/d, one reciprocal
}

// The part both starters share: Markley's fifth-order correction, eqs (21)-(29), from (sin E1, cos E1, f0 = E1 − e sin E1 − M) in FP64, then
// sin/cos(E1 + δ5) by rotation. E1 itself only feeds s.E (dead code unless a caller reads it).
// INV_NR: Newton steps on 1/(1 − e cos E) (1 when it only feeds adjoints, 2 when it feeds a model value, -1 when nobody needs it).
// FOURTH_ORDER: stop at δ4 — for a start within WARM_TOL of the root (the warm rows of k_main) δ4's error is already below the rounding
// of the rotation that follows (tools/kepler_warm_proto.py: the same 1.1e-15 D-weighted maximum as δ5 over config 3's walkers; 1.4e-14 at four times the tolerance).
template <int INV_NR, bool FOURTH_ORDER = false>
__device__ __forceinline__ void kepler_correct(KSol& s, const PC& pc, double E1, double s1, double c1, double f0) {
    const double e = pc.e;
    // f2/2, f2/24, f3/6 of Markley's (21)-(27) from the loop-invariant e/2 and e/6 (f2 = e sin E1 itself is not needed)
    const double hf2 = pc.he * s1, q24 = hf2 * (1.0 / 12.0);
    const double sf3 = (e * (1.0 / 6.0)) * c1;
    const double f1 = fma(-e, c1, 1.0);
    // One hardware reciprocal for the three divisions: the denominators are f1·(den4 + O(δ²)), den4, den4 + O(δ³), so
    // each reciprocal is a Newton step away from the previous one (prototype: tools/kepler_proto.py, same error).
    const double r3 = __builtin_amdgcn_rcp(fma(f1, f1, -(f0 * hf2)));                // ≈2^-23
    const double r4 = f1 * r3;                                                       // ≈ 1/den4, and f1·r3 is also Halley's factor:
    const double d3 = -f0 * r4;                                                      // δ3 = −f0·f1/(f1² − f0 f2/2)
    // δ4 = −f0/den4 and δ5 = −f0/den5 as ONE correction step each on the quotient instead of on the reciprocal:
    // δ' = δ − r·(den·δ + f0) with the crude r = r4 (≈ 1/den to 2^-23 + O(δ²)). The residual den·δ + f0 comes out of one FMA
    // without rounding before the cancellation, and the previous δ is already right to r4's relative error, so the error of δ' is
    // that squared — the same 2^-46 the refined reciprocals gave — for 2 instructions per division instead of 3 (tools/kepler_proto.py's
    // samples: identical 5.13e-16 weighted maximum over 2e6 (M, e) pairs, e -> 1 − 1e-9).
    const double den4 = fma(d3, fma(d3, sf3, hf2), f1);
    const double d4 = fma(-r4, fma(den4, d3, f0), d3);
    double d5 = d4;
    if constexpr (!FOURTH_ORDER) {
        const double den5 = fma(d4, fma(d4, fma(-d4, q24, sf3), hf2), f1);
        d5 = fma(-r4, fma(den5, d4, f0), d4);
    }
    s.E = E1 + d5;                                                   // eq. (29); dead code unless a caller reads it
    // ---- sin/cos(E1 + δ5) by rotation; |δ5| < 5e-4: sin δ = δ(1 − δ²/6) (+1e-19), cos δ − 1 = δ²(−1/2 + δ²/24) (+1e-23)
    const double dd = d5 * d5;
    const double sd = d5 * fma(dd, OCTO_KT[18], 1.0);
    const double cm1 = dd * fma(dd, OCTO_KT[20], -0.5);
    s.sE = fma(s1, cm1, fma(c1, sd, s1));
    s.cE = fma(c1, cm1, fma(-s1, sd, c1));
    if constexpr (INV_NR >= 0) s.invD = rcp_nr<(INV_NR >= 0 ? INV_NR : 0)>(fma(-e, s.cE, 1.0));
}

// Eccentric anomaly and the quantities every projection needs.
// tab: the block's LDS copy of the sin/cos table, or null for the polynomial sincos (kernels without the table).
template <int INV_NR, bool TAB = false>
__device__ __forceinline__ KSol kepler_solve(double t, const PC& pc, const SinCosTab& tab = SinCosTab{nullptr, 0.0, 0.0f, 0u}) {
    KSol s;
    // mean anomaly reduced to [-π, π]: work in orbits, subtract the nearest integer (exact), scale.
    s.dt = t - pc.tp;
    const double u = s.dt * pc.invP;
    const double frac = u - rint(u);                  // M = 2π·frac enters f0 through an FMA below
    const float E1f = markley_starter_f32(frac, pc);
    const double E1 = (double)E1f;
    double s1, c1;
    if constexpr (TAB) sincos_table(E1f, tab, s1, c1);
    else sincos_halfangle(E1, s1, c1);
    const double f0 = fma(-pc.e, s1, fma(-frac, TWO_PI, E1));           // E1 − e sin E1 − M
    // M == 0: E1f = 0 exactly and f0 = 0, so E = 0 like the reference's early return; e == 0: f2 = f3 = 0,
    // d5 = −(E1 − M) exactly, E = M to rounding, like the reference's early return.
    kepler_correct<INV_NR>(s, pc, E1, s1, c1, f0);
    return s;
}

// The warm-started solve (KWarm above). st: the previous row's solution of this (walker, planet), updated; thr: the lane's bound on 1/D;
// dm = 2π (t − t of the previous row), wave-uniform (row record slot 6); row_ok: the row's own step is within the wave's bound (scalar: slot 7
// against WarmState::key_hi). A wave's first row enters with st.invD = +Inf: cold.
// The warm step alone, UNCONDITIONAL (round 6: the two-planet kernels' "last planet always warm" loop): for a wave whose lanes all pass the a-priori test on
// every row — 1/(1 − e) < thr: the bound holds at periastron itself — and whose task has no row beyond the wave's step bound, there is nothing to test
// and nothing to fall back to: no ballot, no branch, the two planets' solves stay in one basic block.
template <int INV_NR>
__device__ __forceinline__ KSol kepler_warm_step(double t, const PC& pc, KWarm& st, double dm) {
    static_assert(INV_NR >= 0, "the warm start needs 1/(1 − e cos E) of every row");
    KSol s;
    s.dt = t - pc.tp;
    const double dM = dm * pc.invP;
    const double x = dM * st.invD;
    const double z = x * st.invD;
    const double dE = fma(-((pc.he * st.sE) * z), x, x);
    const double u = dE * dE;
    const double sr = dE * fma(u, fma(u, fma(u, OCTO_KT[22], OCTO_KT[17]), OCTO_KT[18]), 1.0);
    const double cm1 = u * fma(u, fma(u, fma(u, OCTO_KT[23], OCTO_KT[19]), OCTO_KT[20]), -0.5);
    const double ds = fma(st.sE, cm1, st.cE * sr);
    const double s1 = st.sE + ds;
    const double c1 = fma(st.cE, cm1, fma(-st.sE, sr, st.cE));
    const double f0 = fma(-pc.e, ds, dE - dM);
    kepler_correct<INV_NR, true>(s, pc, 0.0, s1, c1, f0);
    st.sE = s.sE; st.cE = s.cE; st.invD = s.invD;
    return s;
}

#ifndef OCTO_WARM_TRI
#define OCTO_WARM_TRI 1
#endif
template <int INV_NR, bool TRI = (OCTO_WARM_TRI != 0)>
__device__ __forceinline__ KSol kepler_solve_warm(double t, const PC& pc, const SinCosTab& tab, KWarm& st, double thr, double dm, bool row_ok = true) {
    static_assert(INV_NR >= 0, "the warm start needs 1/(1 − e cos E) of every row");
    KSol s;
    s.dt = t - pc.tp;
    double E1 = 0.0, s1, c1, f0;
    const bool warm_row = row_ok && __builtin_amdgcn_ballot_w64(st.invD >= thr) == 0;      // every lane of the wave passes (a NaN bound — an invalid walker — passes: its sums are discarded)
    if constexpr (TRI) {
    // The warm step UNCONDITIONALLY, a rejected row solved again cold behind it — a triangle instead of a diamond (round 6): the straight-line code of a warm
    // row crosses one branch that is not taken, where the diamond's warm arm ended in a taken one and the scheduler could not move anything across either
    // arm; a rejected row wastes the ~40 instructions of the step. Same box: config 3 −1.5 to −2.5 %, the nuisance kernels −2.5 to −3 %, rv_gappy −2 %,
    // wide_prior (20 % of its wave-rows rejected) ±0 (profiles/r6_tri_ab.txt). The first row of a chain enters with 1/D = +Inf: the step's NaNs are dropped.
    {
        const double dM = dm * pc.invP;
        const double x = dM * st.invD;
        const double z = x * st.invD;
        const double dE = fma(-((pc.he * st.sE) * z), x, x);
        const double u = dE * dE;
        const double sr = dE * fma(u, fma(u, fma(u, OCTO_KT[22], OCTO_KT[17]), OCTO_KT[18]), 1.0);
        const double cm1 = u * fma(u, fma(u, fma(u, OCTO_KT[23], OCTO_KT[19]), OCTO_KT[20]), -0.5);
        const double ds = fma(st.sE, cm1, st.cE * sr);
        s1 = st.sE + ds;
        c1 = fma(st.cE, cm1, fma(-st.sE, sr, st.cE));
        f0 = fma(-pc.e, ds, dE - dM);
        kepler_correct<INV_NR, true>(s, pc, E1, s1, c1, f0);
    }
#ifndef OCTO_WARM_NOREDO      // (timing diagnostic only, wrong results: no cold re-solve at all — profiles/r6_tri_ab.txt, section E)
    if (__builtin_expect(!warm_row, 0)) {
        const double uo = s.dt * pc.invP;
        const double frac = uo - rint(uo);
        const float E1f = markley_starter_f32(frac, pc);
        E1 = (double)E1f;
        sincos_table(E1f, tab, s1, c1);
        f0 = fma(-pc.e, s1, fma(-frac, TWO_PI, E1));
        kepler_correct<INV_NR>(s, pc, E1, s1, c1, f0);
    }
#endif
    } else {
    if (warm_row) {
        const double dM = dm * pc.invP;
        const double x = dM * st.invD;
        const double z = x * st.invD;
        const double dE = fma(-((pc.he * st.sE) * z), x, x);
        const double u = dE * dE;
        const double sr = dE * fma(u, fma(u, fma(u, OCTO_KT[22], OCTO_KT[17]), OCTO_KT[18]), 1.0);      // |dE| < 0.066 here (thr >= WARM_MIN_THR): dE⁹/9! < 7e-17
        const double cm1 = u * fma(u, fma(u, fma(u, OCTO_KT[23], OCTO_KT[19]), OCTO_KT[20]), -0.5);
        const double ds = fma(st.sE, cm1, st.cE * sr);               // sin E1 − sin E, without cancellation
        s1 = st.sE + ds;
        c1 = fma(st.cE, cm1, fma(-st.sE, sr, st.cE));
        f0 = fma(-pc.e, ds, dE - dM);
        kepler_correct<INV_NR, true>(s, pc, E1, s1, c1, f0);      // (each branch carries its own copy of the correction: a flag would become a select)
    } else {
        const double uo = s.dt * pc.invP;
        const double frac = uo - rint(uo);
        const float E1f = markley_starter_f32(frac, pc);
        E1 = (double)E1f;
        sincos_table(E1f, tab, s1, c1);
        f0 = fma(-pc.e, s1, fma(-frac, TWO_PI, E1));
        kepler_correct<INV_NR>(s, pc, E1, s1, c1, f0);
    }
    }
    st.sE = s.sE; st.cE = s.cE; st.invD = s.invD;
    return s;
}

}  // namespace octo
