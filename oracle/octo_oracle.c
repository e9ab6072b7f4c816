/*
 * octo_oracle.c — CPU restatement of Octofitter.jl's epoch-loop likelihood path.
 *
 * TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call anything in oracle/. The product path (the HIP library
 * behind include/octofitter_hip.h) never links or falls back to this file.
 *
 * PARITY UNPINNED. The reference is 100 % Julia; Julia is not installed here and cannot be
 * (no network), and the arithmetic of this path lives in third-party packages that are
 * Project.toml dependencies of the reference but are NOT vendored under /root/reference:
 *     PlanetOrbits.jl  compat "0.11.1"  (/root/reference/Project.toml:42,112; no Manifest)
 *     Distributions.jl compat "0.25"    (/root/reference/Project.toml:87)
 *     AstroLib.jl's Markley solver, copied into PlanetOrbits (/root/reference/docs/src/kepler.md:15-19)
 * The reference's tests hold no golden log-likelihood and no known-answer Kepler vector
 * (SURVEY.md §4, §8c). This file therefore restates (a) the reference's own code where it is in
 * the tree, citing file:line, and (b) the published algorithms of those packages, and it is
 * pinned against an INDEPENDENT 50-digit mpmath oracle (oracle/mp_oracle.py ->
 * the fixtures under tests/golden/), against the reference tests' self-consistency properties
 * (tests/test_oracle.py), and against the tutorial astrometry table that the
 * reference ships in test/integration-tests.jl:8-15 — not against outputs of the reference.
 *
 * Build: oracle/Makefile  ->  oracle/liboctooracle.so   (gcc -O3 -march=native -fopenmp, no fast-math)
 */
#define _GNU_SOURCE
#include <alloca.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/octofitter_hip.h"
#include "octo_oracle.h"

#define ORACLE_LOG2PI 1.8378770664093454835606594728112  /* log(2π), Distributions' log2π (IrrationalConstants) */

/* rem2pi(x, RoundNearest) -> [-π, π]   (Julia Base; exact to double-double 2π) */
static double oracle_rem2pi_nearest(double x) {
    const long double twopi = 6.283185307179586476925286766559005768L;
    long double k = nearbyintl((long double)x / twopi);
    return (double)((long double)x - k * twopi);
}
/* rem2pi(x, RoundDown) -> [0, 2π) */
static double oracle_rem2pi_down(double x) {
    const long double twopi = 6.283185307179586476925286766559005768L;
    long double k = floorl((long double)x / twopi);
    double r = (double)((long double)x - k * twopi);
    if (r < 0.0) r = 0.0;
    return r;
}

/* kepler_solver(MA, e, Markley())  [AL]/[PO]: Markley (1995) CeMDA 63, 101 — non-iterative:
 * cubic starter (eqs 5,9,10,14,15,20) + one fifth-order correction (eqs 21-29).
 * Call site in the tree: [ref] src/parameterizations.jl:340; provenance docs/src/kepler.md:15-19. */
double octo_oracle_kepler_markley(double MA, double e) {
    double M = oracle_rem2pi_nearest(MA);
    if (M == 0.0 || e == 0.0) return M;
    const double pi = M_PI;
    const double pi2 = pi * pi;
    double alpha = (3.0 * pi2 + 8.0 * (pi2 - pi * fabs(M)) / (5.0 * (1.0 + e))) / (pi2 - 6.0);   /* (20) */
    double d = 3.0 * (1.0 - e) + alpha * e;                                                       /* (5)  */
    double q = 2.0 * alpha * d * (1.0 - e) - M * M;                                               /* (9)  */
    double r = 3.0 * alpha * d * (d - 1.0 + e) * M + M * M * M;                                   /* (10) */
    double t = fabs(r) + sqrt(q * q * q + r * r);
    double w = cbrt(t * t);                                                                       /* (14) */
    double E1 = (2.0 * r * w / (w * w + w * q + q * q) + M) / d;                                  /* (15) */
    double f2 = e * sin(E1), f3 = e * cos(E1);                                                    /* (26),(27) */
    double f0 = E1 - f2 - M;                                                                      /* (21) */
    double f1 = 1.0 - f3;                                                                         /* (25) */
    double d3 = -f0 / (f1 - f0 * f2 / (2.0 * f1));                                                /* (22) */
    double d4 = -f0 / (f1 + f2 * d3 / 2.0 + d3 * d3 * f3 / 6.0);                                  /* (23) */
    double d5 = -f0 / (f1 + d4 * f2 / 2.0 + d4 * d4 * f3 / 6.0 - d4 * d4 * d4 * f2 / 24.0);       /* (24),(28) */
    return E1 + d5;                                                                               /* (29) */
}

#define NP 0
#include "octo_oracle_core.inc"
#undef NP
#define NP 8
#include "octo_oracle_core.inc"
#undef NP
#define NP 16
#include "octo_oracle_core.inc"
#undef NP
#define NP 32
#include "octo_oracle_core.inc"
#undef NP
#define NP 64
#include "octo_oracle_core.inc"
#undef NP

int32_t octo_oracle_max_partials(void) { return 64; }

int32_t octo_oracle_eval(const octo_consts* c,
                         const octo_obs_desc* obs, int32_t n_obs,
                         const octo_planet_desc* planets, int32_t n_planets,
                         const double* elems, const double* nuis, int64_t ld, int64_t W,
                         double* ll_out, double* g_elems, double* g_nuis,
                         const uint8_t* active_mask, int32_t n_threads) {
    if (!c || (!obs && n_obs > 0) || !planets || n_planets < 1 || !elems || !ll_out || W < 0 || ld < W) return OCTO_EINVAL;
    const int n_el = n_planets * OCTO_N_EL, n_nu = n_obs * OCTO_N_NUIS, n_in = n_el + n_nu;
    const int want_grad = g_elems != NULL;
    int* slot = (int*)malloc(sizeof(int) * (size_t)n_in);
    if (!slot) return OCTO_ENOMEM;
    int n_active = 0;
    for (int k = 0; k < n_in; ++k) {
        int on = want_grad && (active_mask ? active_mask[k] != 0 : 1);
        if (k >= n_el && (!nuis || !g_nuis)) on = 0;
        slot[k] = on ? n_active++ : -1;
    }
    if (n_active > 64) { free(slot); return OCTO_EINVAL; }
    const int np = !want_grad ? 0 : n_active <= 8 ? 8 : n_active <= 16 ? 16 : n_active <= 32 ? 32 : 64;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    if (want_grad) {
        for (int k = 0; k < n_el; ++k) memset(g_elems + (int64_t)k * ld, 0, sizeof(double) * (size_t)W);
        if (g_nuis) for (int k = 0; k < n_nu; ++k) memset(g_nuis + (int64_t)k * ld, 0, sizeof(double) * (size_t)W);
    }
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
    for (int64_t w = 0; w < W; ++w) {
        double in[n_in];
        double gs[64];
        for (int k = 0; k < n_el; ++k) in[k] = elems[(int64_t)k * ld + w];
        for (int k = 0; k < n_nu; ++k) in[n_el + k] = nuis ? nuis[(int64_t)k * ld + w] : 0.0;
        const int has_nuis = nuis != NULL;
        double ll;
        switch (np) {
            case 0:  ll = eval_walker_np0(c, obs, n_obs, planets, n_planets, in, slot, has_nuis, NULL); break;
            case 8:  ll = eval_walker_np8(c, obs, n_obs, planets, n_planets, in, slot, has_nuis, gs); break;
            case 16: ll = eval_walker_np16(c, obs, n_obs, planets, n_planets, in, slot, has_nuis, gs); break;
            case 32: ll = eval_walker_np32(c, obs, n_obs, planets, n_planets, in, slot, has_nuis, gs); break;
            default: ll = eval_walker_np64(c, obs, n_obs, planets, n_planets, in, slot, has_nuis, gs); break;
        }
        if (!isfinite(ll)) {   /* invalid walker: -Inf and zero gradient (C-ABI convention) */
            ll_out[w] = -INFINITY;
            continue;
        }
        ll_out[w] = ll;
        if (want_grad) {
            for (int k = 0; k < n_in; ++k) {
                if (slot[k] < 0) continue;
                if (k < n_el) g_elems[(int64_t)k * ld + w] = gs[slot[k]];
                else g_nuis[(int64_t)(k - n_el) * ld + w] = gs[slot[k]];
            }
        }
    }
    free(slot);
    return OCTO_OK;
}

/* Probe for the tests: solve ONE orbit at ONE epoch in reference order and return the
 * intermediates a PlanetOrbits solution exposes.
 * out = {MA, EA, ν, r, raoff, decoff, radvel, n [rad/yr], K, cart2angle} */
int32_t octo_oracle_orbitsolve(const octo_consts* c, int32_t orbit_kind, const double* el9, double t, double* out10) {
    if (!c || !el9 || !out10) return OCTO_EINVAL;
    orbit_np0 o; sol_np0 s;
    orbit_ctor_np0(&o, c, orbit_kind, dc_np0(el9[OCTO_EL_A]), dc_np0(el9[OCTO_EL_E]), dc_np0(el9[OCTO_EL_I]),
                   dc_np0(el9[OCTO_EL_W]), dc_np0(el9[OCTO_EL_O]), dc_np0(el9[OCTO_EL_TP]), dc_np0(el9[OCTO_EL_M]),
                   dc_np0(el9[OCTO_EL_PLX]));
    orbitsolve_np0(&s, &o, c, t);
    out10[0] = o.n.v * (t - o.tp.v) / c->year2day_julian;
    out10[1] = s.EA.v; out10[2] = s.nu.v; out10[3] = s.r.v;
    out10[4] = raoff_np0(&s, &o).v; out10[5] = decoff_np0(&s, &o).v; out10[6] = radvel_np0(&s, &o).v;
    out10[7] = o.n.v; out10[8] = o.K.v; out10[9] = o.cart2angle.v;
    return OCTO_OK;
}

/* ofti_linear_solve — restated from /root/reference/src/parameterizations.jl:318-405 (the function is IN the tree;
 * only kepler_solver is third-party): dense D (2N×4), block-diagonal W (2N×2N), DtW = D'W, Σ_post⁻¹ = DtW·D + Λ,
 * Σ_post = inv(·) and logdet(·) by LU with partial pivoting (what Julia's `inv`/`logdet` do), μ = (Σ_post·DtW)·d.
 * nl = {e, a, tp, M, plx}; out5 = {A, B, F, G, log_marginal_likelihood}. */
static int lu4(double A[4][4], int piv[4], double* sign) {
    *sign = 1.0;
    for (int k = 0; k < 4; ++k) {
        int p = k; double mx = fabs(A[k][k]);
        for (int i = k + 1; i < 4; ++i) if (fabs(A[i][k]) > mx) { mx = fabs(A[i][k]); p = i; }
        piv[k] = p;
        if (mx == 0.0) return -1;
        if (p != k) { for (int j = 0; j < 4; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; } *sign = -*sign; }
        for (int i = k + 1; i < 4; ++i) {
            A[i][k] /= A[k][k];
            for (int j = k + 1; j < 4; ++j) A[i][j] -= A[i][k] * A[k][j];
        }
    }
    return 0;
}

int32_t octo_oracle_ofti(const octo_consts* c, const double* epochs, const double* ra, const double* dec, const double* s_ra,
                         const double* s_dec, const double* cor, int64_t N, double sigma_abfg, const double* nl, double* out5) {
    if (!c || !nl || !out5 || N < 0) return OCTO_EINVAL;
    const double e = nl[0], a = nl[1], tp = nl[2], M = nl[3], plx = nl[4];
    for (int k = 0; k < 5; ++k) out5[k] = NAN;
    out5[4] = -INFINITY;
    if (!(isfinite(e) && isfinite(a) && isfinite(tp) && isfinite(M) && isfinite(plx)) || !(e >= 0.0 && e < 1.0) || !(a > 0.0) || !(M > 0.0)) return OCTO_OK;
    const double period_days = sqrt(a * a * a / M) * c->kepler_year_to_julian_day;       /* :322 */
    const double n = 2.0 * M_PI / (period_days / c->year2day_julian);                     /* :323 */
    const double sqrt1me2 = sqrt(1.0 - e * e);                                            /* :325 */
    const int64_t R = 2 * N;
    double* D = (double*)calloc((size_t)(R > 0 ? R : 1) * 4, sizeof(double));
    double* d = (double*)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double* Wm = (double*)calloc((size_t)(R > 0 ? R : 1) * 3, sizeof(double));    /* per epoch: W_rr, W_dd, W_rd */
    for (int64_t j = 0; j < N; ++j) {
        const double MA = n / c->year2day_julian * (epochs[j] - tp);                      /* :337 */
        const double EA = octo_oracle_kepler_markley(MA, e);                              /* :340 */
        const double sea = sin(EA), cea = cos(EA);
        const double x = cea - e, y = sea * sqrt1me2;                                     /* :344-345 */
        D[(2 * j) * 4 + 1] = x; D[(2 * j) * 4 + 3] = y;        /* ra row: B, G   :350-351 */
        D[(2 * j + 1) * 4 + 0] = x; D[(2 * j + 1) * 4 + 2] = y; /* dec row: A, F  :352-353 */
        d[2 * j] = ra[j]; d[2 * j + 1] = dec[j];
        const double sr = s_ra[j], sd = s_dec[j], rho = cor ? cor[j] : 0.0;
        const double det = sr * sr * sd * sd * (1.0 - rho * rho);                         /* :362 */
        Wm[3 * j + 0] = sd * sd / det; Wm[3 * j + 1] = sr * sr / det; Wm[3 * j + 2] = -rho * sr * sd / det;
    }
    /* DtW = D' * W  (4 × 2N) */
    double* DtW = (double*)calloc((size_t)(R > 0 ? R : 1) * 4, sizeof(double));
    for (int64_t j = 0; j < N; ++j)
        for (int k = 0; k < 4; ++k) {
            const double dr = D[(2 * j) * 4 + k], dd = D[(2 * j + 1) * 4 + k];
            DtW[k * R + 2 * j] = dr * Wm[3 * j + 0] + dd * Wm[3 * j + 2];
            DtW[k * R + 2 * j + 1] = dr * Wm[3 * j + 2] + dd * Wm[3 * j + 1];
        }
    const double lam = 1.0 / (sigma_abfg * sigma_abfg);
    double S[4][4], LU[4][4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            double acc = 0.0;
            for (int64_t r = 0; r < R; ++r) acc += DtW[i * R + r] * D[r * 4 + k];
            S[i][k] = acc + (i == k ? lam : 0.0);                                          /* :374 */
            LU[i][k] = S[i][k];
        }
    int piv[4]; double sign;
    if (lu4(LU, piv, &sign) != 0) { free(D); free(d); free(Wm); free(DtW); return OCTO_OK; }
    /* inv via LU: solve for each unit vector */
    double Sinv[4][4];
    for (int col = 0; col < 4; ++col) {
        double bb[4] = {0, 0, 0, 0}; bb[col] = 1.0;
        for (int k = 0; k < 4; ++k) { const double t = bb[k]; bb[k] = bb[piv[k]]; bb[piv[k]] = t; }
        for (int i = 0; i < 4; ++i) for (int k = 0; k < i; ++k) bb[i] -= LU[i][k] * bb[k];
        for (int i = 3; i >= 0; --i) { for (int k = i + 1; k < 4; ++k) bb[i] -= LU[i][k] * bb[k]; bb[i] /= LU[i][i]; }
        for (int i = 0; i < 4; ++i) Sinv[i][col] = bb[i];
    }
    /* μ_post = (Σ_post * DtW) * d     :378 */
    double mu[4];
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
        for (int64_t r = 0; r < R; ++r) {
            double t = 0.0;
            for (int k = 0; k < 4; ++k) t += Sinv[i][k] * DtW[k * R + r];
            acc += t * d[r];
        }
        mu[i] = acc;
    }
    double data_quad = 0.0, ldc = 0.0;
    for (int64_t j = 0; j < N; ++j) {                                                     /* :387, :394-400 */
        data_quad += d[2 * j] * (Wm[3 * j + 0] * d[2 * j] + Wm[3 * j + 2] * d[2 * j + 1]) + d[2 * j + 1] * (Wm[3 * j + 2] * d[2 * j] + Wm[3 * j + 1] * d[2 * j + 1]);
        ldc += log(s_ra[j] * s_ra[j] * s_dec[j] * s_dec[j] * (1.0 - (cor ? cor[j] * cor[j] : 0.0)));
    }
    double post_quad = 0.0;
    for (int i = 0; i < 4; ++i) { double t = 0.0; for (int k = 0; k < 4; ++k) t += S[i][k] * mu[k]; post_quad += mu[i] * t; }   /* :388 */
    double logdet = 0.0;
    for (int i = 0; i < 4; ++i) { logdet += log(fabs(LU[i][i])); if (LU[i][i] < 0) sign = -sign; }                            /* :390 */
    if (sign < 0) logdet = NAN;
    const double ldp = 4.0 * log(1.0 / (sigma_abfg * sigma_abfg));                        /* :391 */
    out5[0] = mu[0]; out5[1] = mu[1]; out5[2] = mu[2]; out5[3] = mu[3];
    out5[4] = -0.5 * (data_quad - post_quad + logdet - ldp + ldc) - (double)N * log(2.0 * M_PI);   /* :402 */
    if (!isfinite(out5[4])) { out5[4] = -INFINITY; for (int k = 0; k < 4; ++k) out5[k] = NAN; }
    free(D); free(d); free(Wm); free(DtW);
    return OCTO_OK;
}

int32_t octo_oracle_model_logpost(const octo_consts* c, const octo_obs_desc* obs, int32_t n_obs,
                                  const octo_planet_desc* planets, int32_t n_planets,
                                  const octo_prior* priors, int32_t D, const octo_source* elem_src, const octo_source* nuis_src,
                                  const double* theta_t, int64_t ld, int64_t W, double* lp_out, double* grad_out, int32_t n_threads) {
    if (!c || !planets || !priors || !elem_src || !theta_t || !lp_out || D < 1 || D > 64 || W < 0 || ld < W) return OCTO_EINVAL;
    const int np = !grad_out ? 0 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
    for (int64_t w = 0; w < W; ++w) {
        double th[64], g[64];
        for (int k = 0; k < D; ++k) th[k] = theta_t[(int64_t)k * ld + w];
        double lp;
        switch (np) {
            case 0:  lp = model_logpost_np0(c, obs, n_obs, planets, n_planets, priors, D, elem_src, nuis_src, th, NULL); break;
            case 8:  lp = model_logpost_np8(c, obs, n_obs, planets, n_planets, priors, D, elem_src, nuis_src, th, g); break;
            case 16: lp = model_logpost_np16(c, obs, n_obs, planets, n_planets, priors, D, elem_src, nuis_src, th, g); break;
            case 32: lp = model_logpost_np32(c, obs, n_obs, planets, n_planets, priors, D, elem_src, nuis_src, th, g); break;
            default: lp = model_logpost_np64(c, obs, n_obs, planets, n_planets, priors, D, elem_src, nuis_src, th, g); break;
        }
        lp_out[w] = lp;
        if (grad_out) for (int k = 0; k < D; ++k) grad_out[(int64_t)k * ld + w] = isfinite(lp) ? g[k] : 0.0;
    }
    return OCTO_OK;
}

int32_t octo_oracle_consts_default(octo_consts* out) {
    if (!out) return OCTO_EINVAL;
    /* [PO] PlanetOrbits.jl constants (recalled from the public source; the host passes the
     * live values through octo_consts_set, so these are only defaults):
     *   kepler_year_to_julian_day_conversion_factor = 2π√(au³/GM☉)/86400 with au = 1.495978707e11 m,
     *   GM☉ = 1.3271244e20 m³/s² -> 365.2568983840419 d (reproduced in SURVEY.md §8c). */
    out->kepler_year_to_julian_day = 365.2568983840419;
    out->year2day_julian = 365.25;
    out->au2m = 1.495978707e11;
    out->sec2year_julian = 3.168808781402895e-8;   /* 1/(365.25*86400) */
    out->pc2au = 206265.0;
    out->rad2as = 206265.0;
    out->mjup2msol = 0.0009545942339693249;        /* 1.2668653e17 / 1.3271244e20 (IAU 2015 nominal) */
    return OCTO_OK;
}
