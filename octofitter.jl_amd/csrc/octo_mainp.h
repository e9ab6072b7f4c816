// octo_mainp.h — the epoch-loop kernel for SEVERAL planets with ONE PLANET PER WAVE (round 5), and the finish that goes
// with it for more planets than the templated kernels are compiled for.
//
// k_main<P> gives every wave all P planets of its 64 walkers: P sets of orbit constants and P·PL_N running sums per lane — 231 VALU per
// row at 0.43 of the FP64 peak for two planets, but 0.36 for three and 0.26 for four (held to 256 VGPRs the four-planet kernel parks
// 2-17 doubles per row in scratch memory and runs two waves per SIMD). The model couples the planets of a row only through a 2-vector,
//     ra_model = Σ_p f_p · ra_p,   dec_model = Σ_p f_p · dec_p          (src/likelihoods/relative-astrometry.jl:117-138)
//     rv_model = offset + trend + Σ_p g_p · K_p · V_p                   (rv-absolute.jl:143-155, rv-relative.jl:130-160)
// so here a block is P waves on one tile of 64 walkers and one task of rows, wave p holds planet p's constants and sums ONLY, and the
// waves meet in LDS twice per chunk of MP_R rows:
//   phase 1   wave p solves planet p at the chunk's rows (the solutions stay in its registers) and publishes f_p·(ra_p, dec_p) / g_p K_p V_p;
//   barrier
//   phase 2a  the chunk's rows are dealt round-robin: the wave that owns row r adds the P contributions, forms the residual, the density
//             and ∂ll/∂(model) — ONCE per row, not once per planet — accumulates the observation's own sums (S, nuisance adjoints) for it
//             and publishes (r̄a, d̄ec) / r̄v;
//   barrier
//   phase 2b  wave p turns every row's adjoint into planet p's running sums with the solutions it kept.
// No buffer is written while another wave may still read it (contributions: written in phase 1, read in 2a, rewritten after the second
// barrier; adjoints: written in 2a, read in 2b, rewritten after the next chunk's first barrier), so two barriers per chunk and single
// buffers suffice. Per row and tile that is P solves + ONE density + P adjoint updates — the instruction count of k_main<P> — spread over P
// SIMDs at ~100 VGPRs per lane, no scratch, and P is a RUN-TIME block shape: the partials keep k_main's layout (Layout<2>'s offsets hold
// for every P >= 2), so k_finish<3>, k_finish<4> read them unchanged, and k_finishp below finishes any P up to OCTO_MAX_PLANETS.
// Kind sets: every row kind. An HGCA table reaches k_finishp through `extra` (k_hgcap, octo_hgca.h); the O'Neil prior's term (round 6, more than four planets only: four
// keep k_main<4> for it) is accumulated by the wave of the planet its table is attached to — it needs that planet's E alone — and closed in k_finishp. Marginalised RV (round 6, for systems of more than four planets — the usual RV
// likelihood of a many-planet RV fit): the owning wave of a row accumulates the three sums A = Σ 1/var, B = Σ −2 r/var, C = Σ r²/var of
// rv-absolute-margin.jl:171-180 with the observation's other sums; a gradient takes the two-pass flow of k_main (forward pre-pass over that table's tasks,
// k_marg for μ̂ = −B/2A and A per walker, then the gradient pass with r̄v = 2 (r − μ̂)/var).
#pragma once
#include "octo_kernels.h"

namespace octo {

// Two shapes of the kernel (MP_R = rows per chunk, compile-time: the kept solutions are statically indexed registers; WPE = waves per SIMD
// the register allocation is held to), chosen per planet count from the probes of tools/many_planet_steps.sh (profiles/r5_many_planet_steps.txt):
//   4-6 planets   MP_R = 4, 168 VGPRs (three waves per SIMD): four interleaved solves per phase;
//   7-8 planets   MP_R = 2, 128 VGPRs (four waves per SIMD): an 8-wave block puts two waves on every SIMD, so at three per SIMD a CU holds ONE
//                 block; at four it holds two (8 planets: 3.81 -> 2.56 ms per step of the probe). (MP_R = 4 would also pass the 48 KB default
//                 limit of a launch's dynamic LDS from seven planets on.)
// The planet counts in between: the hardware deals a block's waves to the SIMDs in order, so a block of 5 (6, 7) waves loads SIMD 0 (0-1, 0-2) twice as
// much as the rest and the lock-stepped phases run at the pace of the doubly loaded SIMD — 1.05e11 (5 planets) and 1.36e11 (6) Kepler solves per second
// against 2.1e11 for four planets and 1.6e11 for eight. Blocks of TWO tiles (TPB, a run-time block shape like P: launch_mainp_shape) spread ten or twelve
// waves over the four SIMDs: 1.55e11 and 2.11e11 (profiles/r5_tpb_ab.txt). Seven planets stay at one tile (fourteen waves were slower: 1.34e11 -> 1.15e11).
#ifndef OCTO_MP_NOBAR
#define OCTO_MP_NOBAR 0      // timing diagnostic only (wrong results): 1 drops both barriers of a chunk, 2 the second one
#endif
#define MP_BAR1() do { if (OCTO_MP_NOBAR != 1) __syncthreads(); } while (0)
#define MP_BAR2() do { if (OCTO_MP_NOBAR == 0) __syncthreads(); } while (0)
constexpr int mp_rows(int P) { return P > 6 ? 2 : 4; }
constexpr int mp_wpe(int P) { return P > 6 ? 4 : 3; }

template <bool GRAD, bool NUIS, int KM>
using LayoutP = Layout<2, GRAD, NUIS, KM>;      // OFF_* and PL_N of every P >= 2

// LDS (MP_R = rows per chunk): [sin/cos table][contributions: MP_R × P × 64 × 2][adjoints: MP_R × 64 × 2]; the prologue's exchange (a_p, m_p/M per planet) and the
// final combine of the observation sums (P × OFF_PL × 64) reuse the contribution area.
template <bool GRAD, bool NUIS, int KM, int MP_R>
__host__ __device__ constexpr size_t mainp_contrib_doubles(int P) {
    const size_t contrib = (size_t)MP_R * P * WAVE * 2, comb = (size_t)P * LayoutP<GRAD, NUIS, KM>::OFF_PL * WAVE;
    return contrib > comb ? contrib : comb;
}
// (TPB tiles per block: the table once, the exchange areas once per tile)
template <bool GRAD, bool NUIS, int KM, int MP_R>
__host__ __device__ constexpr size_t mainp_tile_doubles(int P) { return mainp_contrib_doubles<GRAD, NUIS, KM, MP_R>(P) + (size_t)MP_R * WAVE * 2; }
template <bool GRAD, bool NUIS, int KM, int MP_R>
__host__ __device__ constexpr size_t mainp_lds_bytes(int P, int TPB = 1) { return sizeof(double) * (2 * SCT_N + (size_t)TPB * mainp_tile_doubles<GRAD, NUIS, KM, MP_R>(P)); }

// one planet's share of a row, kept from phase 1 to phase 2b
struct MpKept { double sE, cE, invD; };      // (t − tp is re-derived in phase 2b from the row's epoch, which stays in SGPRs: one v_add instead of two registers per row)

template <bool GRAD, bool NUIS, int KM, int MP_R, int WPE>
__attribute__((amdgpu_waves_per_eu(WPE)))
static __global__ __launch_bounds__(64 * 4 * WPE) void k_mainp(EvalArgs a) {
    using L = LayoutP<GRAD, NUIS, KM>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int P = a.n_planets;                                             // = waves per tile
    // TPB tiles per block (blockDim.x = 64·P·TPB): the hardware deals a block's waves to the four SIMDs in order, so a block of 3, 5, 6 or 7
    // waves leaves SIMDs idle or doubly loaded in every lock-stepped phase; several tiles in one block (3 planets x 4, 5-7 planets x 2) fill
    // them evenly. The tiles share the sin/cos table and the barriers, nothing else.
    const int wvb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tib = (wvb >= P ? 1 : 0) + (wvb >= 2 * P ? 1 : 0) + (wvb >= 3 * P ? 1 : 0);      // tile in block (TPB <= 4)
    const int wv = wvb - tib * P;                                          // = the planet this wave owns
    const int TPB = a.n_rblocks;                                           // (k_small's field, free here: tiles per block, set by launch_mainp_shape)
    const int64_t w = ((int64_t)blockIdx.x * TPB + tib) * WAVE + lane;     // (a tile past the batch recomputes the last walker and stores nothing)
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int task = a.task0 + (int)blockIdx.y;
    const Task tk = a.tasks[task];
    const DevObs ob = a.obs[tk.obs];
    const int n_rows = tk.nrows;

    const SinCosTab tab = make_sincos_tab(reinterpret_cast<const double2*>(lds));
    double* const contrib = lds + 2 * SCT_N + (size_t)tib * mainp_tile_doubles<GRAD, NUIS, KM, MP_R>(P);
    double* const adj = contrib + mainp_contrib_doubles<GRAD, NUIS, KM, MP_R>(P);

    // ---- prologue: the table, this wave's planet, and what the coefficients need of the others (a_p, m_p/M)
    {
        const double2* __restrict__ g = reinterpret_cast<const double2*>(a.sctab);
        double2* t = reinterpret_cast<double2*>(lds);
        for (int i = threadIdx.x; i < SCT_N; i += (int)blockDim.x) t[i] = g[i];
    }
    PC pc;
    {
        double elv[OCTO_N_EL];
        const double* el = a.elems + (int64_t)wv * OCTO_N_EL * a.ld + wl;
#pragma unroll
        for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = el[(int64_t)k * a.ld];
        const SetupOut so = setup_planet_vals<true>(elv, a.c, a.orbit_kind[wv], a.has_mass[wv]);      // the routine k_finish derives its constants with
        const double* v = so.v;
        pc.invP = v[WC_INVP]; pc.tp = v[WC_TP]; pc.e = v[WC_E]; pc.beta = v[WC_BETA]; pc.eob = v[WC_EOB];
        pc.cB = v[WC_CB]; pc.cG = v[WC_CG]; pc.cA = v[WC_CA]; pc.cF = v[WC_CF]; pc.K = v[WC_K]; pc.cw = v[WC_COSW];
        pc.sw = v[WC_SINW]; pc.mu = v[WC_MU]; pc.a = v[WC_A]; pc.cGb = v[WC_CGB]; pc.cFb = v[WC_CFB]; pc.cBe = v[WC_CBE]; pc.cAe = v[WC_CAE];
        const float2 fa = *reinterpret_cast<const float2*>(&v[WC_F32A]);
        const float2 fb = *reinterpret_cast<const float2*>(&v[WC_F32B]);
        set_starter(pc, fa.x, fa.y, fb.x);
    }
    const bool oneil = L::HAS_ONEIL && (ob.kind == OCTO_ONEIL_RADEC || ob.kind == OCTO_ONEIL_SEPPA);      // prior-observable.jl:78-137: the wrapped table's rows + the prior's term
    const bool is_astrom = !L::HAS_RV || ob.kind == OCTO_ASTROM_RADEC || ob.kind == OCTO_ASTROM_SEPPA || oneil;
    const bool rel = (KM & KM_RVREL) && ob.kind == OCTO_RV_REL;
    const bool marg = (KM & KM_MARG) && ob.kind == OCTO_RV_ABS_MARG;
    // the semi-major axis of the planet the table is attached to, through LDS (relative-astrometry.jl:120-123, rv-relative.jl:148-152: strictly inner)
    contrib[wv * WAVE + lane] = pc.a;
    __syncthreads();                                                       // (also: table filled)
    const double a_this = (is_astrom || rel) ? contrib[ob.planet * WAVE + lane] : 0.0;
    __syncthreads();                                                       // the exchange area becomes the contribution buffer
    // coefficient of this planet in the model: astrometry 1 | m/M | 0; relative RV 1 | −m/M | 0; absolute RV −m/M
    const double coef = is_astrom ? ((wv == ob.planet) ? 1.0 : ((pc.a < a_this) ? pc.mu : 0.0))
                                  : (rel ? ((wv == ob.planet) ? 1.0 : ((pc.a < a_this) ? -pc.mu : 0.0)) : -pc.mu);
    const double gK = coef * pc.K;                                         // RV: contribution = gK · V
    // θ_obs of the table (every wave: each owns some rows of every chunk)
    double n0 = 0.0, n1 = is_astrom ? 1.0 : 0.0, n2 = 0.0;                 // jitter | offset, platescale | jitter, northangle | trend
    if constexpr (NUIS) {
        const double* nu = a.nuis + (int64_t)tk.obs * OCTO_N_NUIS * a.ld + wl;
        n0 = nu[0]; n1 = nu[(int64_t)a.ld]; n2 = nu[(int64_t)2 * a.ld];
    }
    double sn = 0.0, cn = 1.0;
    if constexpr (NUIS && (KM & (KM_RADEC | KM_SEPPA)) != 0) { if (is_astrom) sincos_reduced(n2, sn, cn); }
    const double jit = is_astrom ? n0 : n1, j2 = jit * jit;
    const bool seppa = (KM & KM_SEPPA) && (ob.kind == OCTO_ASTROM_SEPPA || (L::HAS_ONEIL && ob.kind == OCTO_ONEIL_SEPPA));
    const double ib2 = GRAD ? 1.0 / (pc.beta * pc.beta) : 0.0;
    double mu_hat = 0.0, iA = 0.0;                                         // marginalised RV, gradient pass: the pre-pass's μ̂ and 1/A of this walker (k_marg)
    if constexpr (GRAD && L::HAS_MARG) {
        if (marg && a.marg) { mu_hat = a.marg[((int64_t)tk.obs * 2 + 0) * a.ldw + wl]; iA = 1.0 / a.marg[((int64_t)tk.obs * 2 + 1) * a.ldw + wl]; }
    }
    int n_owned = 0;                                                       // rows this wave owned (marginalised RV with nuisances: Σ log 2π var)

    double ao[L::OFF_PL];                                                  // the observation's sums over the rows this wave owns
#pragma unroll
    for (int k = 0; k < L::OFF_PL; ++k) ao[k] = 0.0;
    double ap[L::PL_N > 0 ? L::PL_N : 1];                                  // this planet's sums
#pragma unroll
    for (int k = 0; k < L::PL_N; ++k) ap[k] = 0.0;
    LogProd lp;

    const crow_t rows = constant_rows((NUIS ? ob.raw : ob.pre) + (int64_t)tk.row0 * ROW_STRIDE);
    int base = 0;                                                          // j0 mod P, carried (no integer division in the loop)
    for (int j0 = 0; j0 < n_rows; j0 += MP_R) {
        MpKept kp[MP_R];
        // ---------------- phase 1: this planet at the chunk's rows. The kind branch sits OUTSIDE the unrolled row loop: MP_R independent
        // solves in one basic block, which the scheduler interleaves (a branch per row left each solve a dependent chain of its own).
        double tr[MP_R];
#pragma unroll
        for (int r = 0; r < MP_R; ++r) tr[r] = rows[(int64_t)(j0 + r < n_rows ? j0 + r : n_rows - 1) * ROW_STRIDE];      // (a short last chunk re-solves the last row)
        if (is_astrom) {
#pragma unroll
            for (int r = 0; r < MP_R; ++r) {
                const KSol s = kepler_solve<1, true>(tr[r], pc, tab);
                kp[r] = {s.sE, s.cE, s.invD};
                if constexpr (L::HAS_ONEIL) {
                    // the O'Neil prior's term of this row, by the wave of the planet the table is attached to (it needs that planet's E alone):
                    // t = 3M(e + cos E) + 2(−2 + e² + e cos E) sin E, Σ|t| and its adjoints — astrom_row's lines (prior-observable.jl:129-133)
                    if (oneil && wv == ob.planet && j0 + r < n_rows) {
                        const double ee = pc.e, sE = s.sE, cE = s.cE;
                        const double Mm = fma(-ee, sE, s.E);
                        const double c2 = fma(ee, ee + cE, -2.0);
                        const double tt = fma(3.0 * Mm, ee + cE, 2.0 * c2 * sE);
                        ao[L::OFF_ONEIL] += fabs(tt);
                        if constexpr (GRAD) {
                            const double sg = tt < 0.0 ? -1.0 : 1.0;
                            const double D = fma(-ee, cE, 1.0);
                            const double tM = 3.0 * (ee + cE);
                            const double tE = fma(-3.0 * Mm, sE, 2.0 * fma(c2, cE, -(ee * sE * sE))) + tM * D;
                            const double te = fma(2.0 * sE, 2.0 * ee + cE, 3.0 * Mm) - tM * sE;
                            const double Mb = sg * tE * s.invD;
                            ao[L::OFF_ONEIL + 1] += sg * te + Mb * sE;
                            ao[L::OFF_ONEIL + 2] += Mb;
                            ao[L::OFF_ONEIL + 3] = fma(Mb, tr[r] - pc.tp, ao[L::OFF_ONEIL + 3]);
                        }
                    }
                }
                const double c0 = coef * fma(pc.cB, s.cE, fma(pc.cGb, s.sE, -pc.cBe));
                const double c1 = coef * fma(pc.cA, s.cE, fma(pc.cFb, s.sE, -pc.cAe));
                *reinterpret_cast<double2*>(&contrib[((size_t)(r * P + wv) * WAVE + lane) * 2]) = make_double2(c0, c1);
            }
        } else if constexpr (L::HAS_RV) {
#pragma unroll
            for (int r = 0; r < MP_R; ++r) {
                const KSol s = kepler_solve<2, true>(tr[r], pc, tab);
                kp[r] = {s.sE, s.cE, s.invD};
                const double cnu = (s.cE - pc.e) * s.invD, snu = pc.beta * s.sE * s.invD;
                const double c0 = gK * fma(cnu + pc.e, pc.cw, -(snu * pc.sw));      // g K (cos(ν+ω) + e cos ω)
                *reinterpret_cast<double2*>(&contrib[((size_t)(r * P + wv) * WAVE + lane) * 2]) = make_double2(c0, 0.0);
            }
        }
        MP_BAR1();
        // ---------------- phase 2a: the rows this wave owns — model, residual, density, ∂ll/∂model
#pragma unroll
        for (int r = 0; r < MP_R; ++r) {
            const int j = j0 + r;
            int owner = base + r;
            while (owner >= P) owner -= P;
            if (owner != wv) continue;                                     // wave-uniform
            if (j >= n_rows) {                                             // a short last chunk: phase 2b runs unconditionally over MP_R rows
                if constexpr (GRAD) *reinterpret_cast<double2*>(&adj[((size_t)r * WAVE + lane) * 2]) = make_double2(0.0, 0.0);
                continue;
            }
            const crow_t rw = rows + (int64_t)j * ROW_STRIDE;
            double m0 = 0.0, m1 = 0.0;
            for (int p = 0; p < P; ++p) {
                const double2 c = *reinterpret_cast<const double2*>(&contrib[((size_t)(r * P + p) * WAVE + lane) * 2]);
                m0 += c.x; m1 += c.y;
            }
            double b0 = 0.0, b1 = 0.0;                                     // ∂ll/∂(ra_model, dec_model) | ∂ll/∂rv_model
            if (is_astrom) {
                const double y1 = rw[1], y2 = rw[2], c3 = rw[3], c4 = rw[4], c5 = rw[5];
                double r1, r2, irho = 1.0, u1 = 0.0, u2 = 0.0;
                if (seppa) {                                               // relative-astrometry.jl:192-202
                    const double rho2 = fma(m0, m0, m1 * m1);
                    irho = rsqrt_nr(rho2);
                    const double pa = atan2_fast(m0, m1);
                    double dpa = (y1 + n2) - pa + PI;
                    dpa = rem_2pi_trunc(dpa) - PI;
                    dpa = dpa < -PI ? dpa + TWO_PI : dpa;
                    r1 = dpa;
                    r2 = fma(-rho2, irho, y2 * n1);
                } else if constexpr (NUIS) {                               // :210-215
                    u1 = fma(y1, cn, y2 * sn); u2 = fma(y2, cn, -(y1 * sn));
                    r1 = fma(n1, u1, -m0); r2 = fma(n1, u2, -m1);
                } else { r1 = y1 - m0; r2 = y2 - m1; }
                double g1, g2;
                if constexpr (!NUIS) {
                    double a1, a2;
                    if constexpr (L::HAS_COR) { a1 = fma(c3, r1, c5 * r2); a2 = fma(c5, r1, c4 * r2); }
                    else { a1 = c3 * r1; a2 = c4 * r2; }
                    ao[L::OFF_S] = fma(r1, a1, fma(r2, a2, ao[L::OFF_S]));
                    g1 = -a1; g2 = -a2;
                } else {
                    const double v1 = fma(c3, c3, j2), v2 = fma(c4, c4, j2);
                    const double v12 = v1 * v2;
                    const double iv12 = rcp_nr<2>(v12);
                    const double iv1 = iv12 * v2, iv2 = iv12 * v1;
                    double a1, a2;
                    if (L::HAS_COR && ob.has_cor) {
                        const double cor = c5, omc = 1.0 - cor * cor, ic = rcp_nr<2>(omc), is = rsqrt(v12);
                        a1 = fma(r1, iv1, -(cor * r2 * is)) * ic;
                        a2 = fma(r2, iv2, -(cor * r1 * is)) * ic;
                        lp.mul(v12 * omc);
                    } else { a1 = r1 * iv1; a2 = r2 * iv2; lp.mul(v12); }
                    ao[L::OFF_S] = fma(r1, a1, fma(r2, a2, ao[L::OFF_S]));
                    g1 = -a1; g2 = -a2;
                    if constexpr (GRAD) {
                        ao[L::OFF_NU + OCTO_NU_JITTER] += jit * fma(fma(r1, a1, -1.0), iv1, fma(r2, a2, -1.0) * iv2);
                        if (seppa) {
                            ao[L::OFF_NU + OCTO_NU_PLATESCALE] = fma(g2, y2, ao[L::OFF_NU + OCTO_NU_PLATESCALE]);
                            ao[L::OFF_NU + OCTO_NU_NORTHANGLE] += g1;
                        } else {
                            ao[L::OFF_NU + OCTO_NU_PLATESCALE] += g1 * u1 + g2 * u2;
                            ao[L::OFF_NU + OCTO_NU_NORTHANGLE] += n1 * (g1 * u2 - g2 * u1);
                        }
                    }
                }
                if (seppa) {
                    const double pab = -g1, rhob = -g2;
                    b0 = (rhob * m0 + pab * m1 * irho) * irho;
                    b1 = (rhob * m1 - pab * m0 * irho) * irho;
                } else { b0 = -g1; b1 = -g2; }
            } else if constexpr (L::HAS_RV) {
                // rv-absolute.jl:143-204, rv-relative.jl:131-210
                const double rv = rw[1], c2 = rw[2], basis = NUIS ? rw[3] : 0.0;
                const double model = (NUIS ? fma(n2, basis, marg ? 0.0 : n0) : 0.0) + m0;      // (a marginalised table has no offset: rv-absolute-margin.jl:111)
                const double resid = rv - model;
                double iv;
                if constexpr (NUIS) { const double var = fma(c2, c2, j2); iv = rcp_nr<2>(var); lp.mul(var); } else { iv = c2; }
                n_owned += 1;
                if (L::HAS_MARG && marg) {
                    if constexpr (L::HAS_MARG) {      // rv-absolute-margin.jl:171-180
                        ao[L::OFF_MARG + 0] += iv;
                        ao[L::OFF_MARG + 1] = fma(-2.0 * resid, iv, ao[L::OFF_MARG + 1]);
                        ao[L::OFF_MARG + 2] = fma(resid * resid, iv, ao[L::OFF_MARG + 2]);
                    }
                    const double dmv = resid - mu_hat;
                    b0 = 2.0 * dmv * iv;
                    if constexpr (GRAD && NUIS) {
                        ao[L::OFF_NU + OCTO_NU_RV_JITTER] += 2.0 * jit * iv * (dmv * dmv * iv - 1.0 + iv * iA);
                        ao[L::OFF_NU + OCTO_NU_RV_TREND] = fma(b0, basis, ao[L::OFF_NU + OCTO_NU_RV_TREND]);
                    }
                } else {
                    ao[L::OFF_S] = fma(resid * resid, iv, ao[L::OFF_S]);
                    b0 = resid * iv;
                    if constexpr (GRAD && NUIS) {
                        ao[L::OFF_NU + OCTO_NU_RV_OFFSET] += b0;
                        ao[L::OFF_NU + OCTO_NU_RV_JITTER] += jit * iv * (resid * resid * iv - 1.0);
                        ao[L::OFF_NU + OCTO_NU_RV_TREND] = fma(b0, basis, ao[L::OFF_NU + OCTO_NU_RV_TREND]);
                    }
                }
            }
            if constexpr (GRAD) *reinterpret_cast<double2*>(&adj[((size_t)r * WAVE + lane) * 2]) = make_double2(b0, b1);
        }
        base += MP_R;
        while (base >= P) base -= P;
        MP_BAR2();
        // ---------------- phase 2b: every row's adjoint into this planet's sums (missing rows of a short last chunk carry zero adjoints)
        if constexpr (GRAD) {
            if (is_astrom) {
#pragma unroll
                for (int r = 0; r < MP_R; ++r) {
                    const double2 b = *reinterpret_cast<const double2*>(&adj[((size_t)r * WAVE + lane) * 2]);
                    const double sE = kp[r].sE, cE = kp[r].cE, invD = kp[r].invD, dt = tr[r] - pc.tp;
                    const double ra_f = coef * b.x, de_f = coef * b.y;
                    ap[L::U1] = fma(cE, ra_f, ap[L::U1]);
                    ap[L::U2] = fma(sE, ra_f, ap[L::U2]);
                    ap[L::U3] = fma(cE, de_f, ap[L::U3]);
                    ap[L::U4] = fma(sE, de_f, ap[L::U4]);
                    ap[L::U5] += ra_f;
                    ap[L::U6] += de_f;
                    // ∂/∂(m/M) of a reflex term: r̄a·ra_p + d̄ec·dec_p (the planet's own offsets, re-derived from the kept solution)
                    const double rap = fma(pc.cB, cE, fma(pc.cGb, sE, -pc.cBe)), dep = fma(pc.cA, cE, fma(pc.cFb, sE, -pc.cAe));
                    ap[L::GC] += (wv == ob.planet) ? 0.0 : ((coef != 0.0) ? fma(b.x, rap, b.y * dep) : 0.0);
                    const double dra = fma(pc.cGb, cE, -(pc.cB * sE)), dde = fma(pc.cFb, cE, -(pc.cA * sE));
                    const double Mb = fma(ra_f, dra, de_f * dde) * invD;
                    ap[L::GE] = fma(Mb, sE, ap[L::GE]);
                    ap[L::GM] += Mb;
                    ap[L::GT] = fma(Mb, dt, ap[L::GT]);
                }
            } else if constexpr (L::HAS_RV) {
#pragma unroll
                for (int r = 0; r < MP_R; ++r) {
                    const double2 b = *reinterpret_cast<const double2*>(&adj[((size_t)r * WAVE + lane) * 2]);
                    const double sE = kp[r].sE, cE = kp[r].cE, invD = kp[r].invD, dt = tr[r] - pc.tp;
                    const double rvb = b.x;
                    const double cnu = (cE - pc.e) * invD, snu = pc.beta * sE * invD;
                    const double V = fma(cnu + pc.e, pc.cw, -(snu * pc.sw));
                    ap[L::GK] = fma(coef * V, rvb, ap[L::GK]);
                    const bool via_mu = rel ? (wv != ob.planet && coef != 0.0) : true;
                    ap[L::GC] += via_mu ? -(pc.K * V * rvb) : 0.0;
                    const double Vb = gK * rvb;
                    const double S = fma(snu, pc.cw, cnu * pc.sw);
                    ap[L::GW] = fma(Vb, -fma(pc.e, pc.sw, S), ap[L::GW]);
                    const double VS = Vb * S;
                    const double Mb = -(VS * pc.beta) * (invD * invD);
                    double eb = Vb * pc.cw;
                    eb = fma(-(snu * ib2), VS, eb);
                    eb = fma(Mb, sE, eb);
                    ap[L::GE] += eb;
                    ap[L::GM] += Mb;
                    ap[L::GT] = fma(Mb, dt, ap[L::GT]);
                }
            }
        }
    }
    if constexpr (NUIS) {                                                   // Σ log|Σ_row| / Σ log var (marginalised RV: Σ log 2π var) over the rows this wave owned
        double lg = lp.log_value();
        if (L::HAS_MARG && marg) lg = fma((double)n_owned, LOG2PI, lg);
        ao[L::OFF_S] += lg;
    }
    // ---- one partial per (tile, task): the planets' sums straight from their waves, the observation's sums added in wave order
    __syncthreads();                                                        // the last chunk's contributions are dead
    if (wv > 0) {
#pragma unroll
        for (int k = 0; k < L::OFF_PL; ++k) contrib[((size_t)(wv - 1) * L::OFF_PL + k) * WAVE + lane] = ao[k];
    }
    __syncthreads();
    if (w < a.W) {
        double* out = a.partials + (int64_t)task * (L::OFF_PL + P * L::PL_N) * a.ldw + w;
        if (wv == 0) {
            for (int q = 1; q < P; ++q) {
#pragma unroll
                for (int k = 0; k < L::OFF_PL; ++k) ao[k] += contrib[((size_t)(q - 1) * L::OFF_PL + k) * WAVE + lane];
            }
#pragma unroll
            for (int k = 0; k < L::OFF_PL; ++k) out[(int64_t)k * a.ldw] = ao[k];
        }
#pragma unroll
        for (int k = 0; k < L::PL_N; ++k) out[(int64_t)(L::OFF_PL + wv * L::PL_N + k) * a.ldw] = ap[k];
    }
}

// ------------------------------------------------------------------------------------ k_finishp
// finish_tile_multi with the number of planets as a run-time block shape (1 + P waves: wave 0 the observations, wave 1 + p planet p), for
// systems of more planets than k_finish<P> is compiled for. Same sums in the same order as finish_tile_multi with one wave per planet.
template <bool GRAD, bool NUIS, int KM>
static __global__ __launch_bounds__(64 * (1 + OCTO_MAX_PLANETS)) void k_finishp(EvalArgs a) {
    using L = LayoutP<GRAD, NUIS, KM>;
    extern __shared__ __attribute__((aligned(16))) double lds[];            // (1 + P) rows of 64 validity flags [+ P x 6 rows: the O'Neil terms of each planet's adjoints]
    const int lane = threadIdx.x & (WAVE - 1);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = a.n_planets;
    constexpr int PLN = L::PL_N, NOB = L::OFF_PL;
    const int nacc = NOB + P * PLN;
    const int64_t w = (int64_t)blockIdx.x * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    double gp[PLN > 0 ? PLN : 1];
#pragma unroll
    for (int k = 0; k < PLN; ++k) gp[k] = 0.0;
    double ll = 0.0;
    FinPC fp = {};
    double elv[OCTO_N_EL];
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = 0.0;
    bool ok_mine = true;
    double* const og_lds = lds + (size_t)(1 + P) * WAVE;                    // [P][6][64]: ΔGE, ΔGM, ΔGT, Δā, ΔM̄tot, Δē per planet from its O'Neil tables
    constexpr int KMF = KM & ~KM_ONEIL;                                     // obs_finish's closed forms without the O'Neil block (its planet arrays are sized by the template's P = 2)
    if (grp == 0) {
        double dummy_sma[2] = {0.0, 0.0}, dummy_e[2] = {0.0, 0.0}, dummy_M[2] = {1.0, 1.0}, og[1] = {0.0};
        if constexpr (L::HAS_ONEIL && GRAD) {
            for (int k = 0; k < P * 6; ++k) og_lds[(size_t)k * WAVE + lane] = 0.0;
        }
        for (int o = 0; o < a.n_obs; ++o) {
            double vo[NOB];
#pragma unroll
            for (int k = 0; k < NOB; ++k) vo[k] = 0.0;
            const int t0 = a.obs_range[2 * o], t_end = a.obs_range[2 * o + 1];
#pragma clang loop unroll_count(4)
            for (int tt = t0; tt < t_end; ++tt) {
                const double* pt = a.partials + (int64_t)tt * nacc * a.ldw + wl;
                double to[NOB];
#pragma unroll
                for (int k = 0; k < NOB; ++k) to[k] = pt[(int64_t)k * a.ldw];
#pragma unroll
                for (int k = 0; k < NOB; ++k) vo[k] += to[k];
            }
            double v[NOBS_ACC];
#pragma unroll
            for (int k = 0; k < NOBS_ACC; ++k) v[k] = 0.0;
            v[0] = vo[L::OFF_S];
            if constexpr (L::HAS_MARG) { v[1] = vo[L::OFF_MARG + 0]; v[2] = vo[L::OFF_MARG + 1]; v[3] = vo[L::OFF_MARG + 2]; }
            if constexpr (L::N_NU > 0) { v[4] = vo[L::OFF_NU + 0]; v[5] = vo[L::OFF_NU + 1]; v[6] = vo[L::OFF_NU + 2]; }
            ll += obs_finish<2, GRAD, NUIS, KMF>(a.obs, a.ld, L::N_NU > 0 ? a.g_nuis + (int64_t)o * OCTO_N_NUIS * a.ld + w : nullptr, a.extra ? a.extra + wl : nullptr,
                                                 a.ldw, a.c.k_yr, o, v, a.obs_const[o], dummy_sma, dummy_e, dummy_M, w < a.W, og, P);
            if constexpr (L::HAS_ONEIL) {
                // ln_prior = 2 log(Σ|t_j| · ∛P / √(1−e²)), P = period/365.25 (prior-observable.jl:96,136-139) — obs_finish's block for a run-time planet index
                const int kind = a.obs[o].kind;
                if ((kind == OCTO_ONEIL_RADEC || kind == OCTO_ONEIL_SEPPA) && a.obs[o].n > 0) {
                    const int ip = a.obs[o].planet;
                    const double sma = setup_planet<true>(a, ip, wl).v[WC_A];
                    const double e = a.elems[((int64_t)ip * OCTO_N_EL + OCTO_EL_E) * a.ld + wl], Mt = a.elems[((int64_t)ip * OCTO_N_EL + OCTO_EL_M) * a.ld + wl];
                    const double s_abs = vo[L::OFF_ONEIL];
                    const double Pyr = a.c.k_yr * sqrt(sma * sma * sma / Mt) / 365.25;
                    ll += 2.0 * log(s_abs * cbrt(Pyr) / sqrt(1.0 - e * e));
                    if constexpr (GRAD) {
                        const double f = 2.0 / s_abs;
                        double* gq = og_lds + (size_t)ip * 6 * WAVE + lane;
                        gq[0 * WAVE] += f * vo[L::OFF_ONEIL + 1]; gq[1 * WAVE] += f * vo[L::OFF_ONEIL + 2]; gq[2 * WAVE] += f * vo[L::OFF_ONEIL + 3];
                        gq[3 * WAVE] += 1.0 / sma; gq[4 * WAVE] += -1.0 / (3.0 * Mt); gq[5 * WAVE] += 2.0 * e / (1.0 - e * e);
                    }
                }
            }
        }
        if (a.extra) ll += a.extra[wl];                                     // the proper-motion anomaly (k_hgcap ahead of this launch)
        ok_mine = isfinite(ll);
        if constexpr (!GRAD) {
            for (int p = 0; p < P; ++p) ok_mine = ok_mine && setup_planet<true>(a, p, wl).ok;
        }
        if (a.nuis)
            for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) ok_mine = ok_mine && isfinite(a.nuis[(int64_t)k * a.ld + wl]);
    } else if constexpr (GRAD) {
        const int p = grp - 1;
        const SetupOut so = setup_planet<true>(a, p, wl);
        fp.sma = so.v[WC_A]; fp.P_d = rcp_nr<2>(so.v[WC_INVP]); fp.beta = so.v[WC_BETA];
        fp.si = so.v[WC_SINI]; fp.ci = so.v[WC_COSI]; fp.sO = so.v[WC_SINO]; fp.cO = so.v[WC_COSO];
        fp.sw = so.v[WC_SINW]; fp.cw = so.v[WC_COSW];
#pragma unroll
        for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = so.el[k];
        ok_mine = so.ok;
        const double* pcol = a.partials + (int64_t)(NOB + p * PLN) * a.ldw + wl;
        const int64_t tstride = (int64_t)nacc * a.ldw;
#pragma clang loop unroll_count(4)
        for (int tt = 0; tt < a.n_tasks; ++tt) {
            const double* pt = pcol + (int64_t)tt * tstride;
            double tmp[PLN > 0 ? PLN : 1];
#pragma unroll
            for (int k = 0; k < PLN; ++k) tmp[k] = pt[(int64_t)k * a.ldw];
#pragma unroll
            for (int k = 0; k < PLN; ++k) gp[k] += tmp[k];
        }
    }
    bool ok = ok_mine;
    if constexpr (GRAD) {
        lds[grp * WAVE + lane] = ok_mine ? 1.0 : 0.0;
        __syncthreads();
        ok = true;
        for (int q = 0; q <= P; ++q) ok = ok && lds[q * WAVE + lane] != 0.0;
    }
    const bool live = w < a.W;
    if (live && grp == 0) {
        a.ll_out[w] = ok ? ll : -INFINITY;
        if constexpr (GRAD && L::N_NU > 0) {
            if (!ok)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) a.g_nuis[(int64_t)k * a.ld + w] = 0.0;
        }
    }
    if constexpr (GRAD) {
        if (grp > 0 && live) {
            const int p = grp - 1;
            double ogp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if constexpr (L::HAS_ONEIL) {
#pragma unroll
                for (int k = 0; k < 6; ++k) ogp[k] = og_lds[((size_t)p * 6 + k) * WAVE + lane];      // (written by wave 0 ahead of the barrier above)
            }
            planet_finish<2, GRAD, NUIS, KM, OCTO_FIN_FAST>(elv, a.g_elems + (int64_t)p * OCTO_N_EL * a.ld + w, a.ld, a.extra ? a.extra + w : nullptr, a.ldw, a.c,
                                                            a.orbit_kind[p], a.has_mass[p], p, gp, L::HAS_ONEIL ? ogp : nullptr, fp, ok);
        }
    }
    if (a.mt_lpp) model_tail_n(a, w, grp, GRAD ? 1 + P : 1, P);
}

}  // namespace octo
