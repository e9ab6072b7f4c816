// octo_hgca.h — HGCAInstantaneousObs on the device: the Hipparcos–Gaia proper-motion anomaly of the
// primary from the companions' reflex position and velocity at a handful of epochs (src/likelihoods/hgca.jl:155-400).
// This term has no long epoch loop (4·N_ave rows), so it does not go through k_main: one thread per (walker, input
// direction) carries the value and ONE partial of every quantity as a forward-mode dual — what ForwardDiff does in the
// reference — and writes ∂ll/∂input[dir] directly; no reduction is needed. k_finish adds the results to the epoch-loop sums.
#pragma once
#include "octo_model.h"

namespace octo {

// Reflex position [mas] and proper motion [mas/yr] of the primary due to one companion at time t, along RA or Dec.
// E-form of PlanetOrbits' raoff/decoff/pmra/pmdec(sol, mass): X = cos E − e, Y = √(1−e²) sin E, Thiele-Innes
// projection, Ė = n/(1 − e cos E); the reference's ν-form is restated in oracle/octo_oracle_core.inc.
struct HgcaPlanet {
    Dual<1> e, beta, tp, n_day, T, cA, cB, cF, cG, fac;
    PC pc;     // value-only constants for the Kepler solve
};

__device__ __forceinline__ void hgca_solve(const HgcaPlanet& h, double t, int axis, double yd, Dual<1>& pos, Dual<1>& vel) {
    using D = Dual<1>;
    const KSol s = kepler_solve<-1>(t, h.pc);
    // ∂E/∂M = 1/(1 − e cos E), ∂E/∂e = sin E/(1 − e cos E); M = n (t − tp)
    const double invD = 1.0 / (1.0 - h.e.v * s.cE);
    const D MA = h.n_day * (dconst<1>(t) - h.tp);
    D E; E.v = 0.0; E.d[0] = (MA.d[0] + s.sE * h.e.d[0]) * invD;
    const D sE = chain(E, s.sE, s.cE), cE = chain(E, s.cE, -s.sE);
    const D X = cE - h.e, Y = h.beta * sE;
    const D Edot = h.n_day / (dconst<1>(1.0) - h.e * cE);
    const D c1 = axis == OCTO_HGCA_RA ? h.cB : h.cA, c2 = axis == OCTO_HGCA_RA ? h.cG : h.cF;
    pos = h.fac * (h.T * (c1 * X + c2 * Y));
    vel = h.fac * (h.T * (c2 * (h.beta * cE) - c1 * sE) * Edot * yd);
}

// logpdf(MvNormal([σ1² ρσ1σ2; ρσ1σ2 σ2²]), [r1, r2])
__device__ __forceinline__ Dual<1> hgca_logpdf2(const Dual<1>& r1, const Dual<1>& r2, double s1, double s2, double rho) {
    const double omr = 1.0 - rho * rho;
    const Dual<1> z1 = r1 * (1.0 / s1), z2 = r2 * (1.0 / s2);
    const Dual<1> q = (z1 * z1 - (z1 * z2) * (2.0 * rho) + z2 * z2) * (1.0 / omr);
    return q * (-0.5) + (-LOG2PI - 0.5 * log(s1 * s1 * s2 * s2 * omr));
}

// One companion's constants with input direction `dirl` (0 … 8 = this planet's element rows, anything else: none) seeded — one partial, as ForwardDiff would carry it.
__device__ __forceinline__ void hgca_setup_one(const double (&elv)[OCTO_N_EL], int orbit_kind, int has_mass, const DevConsts& c, int dirl, HgcaPlanet& h, bool& visual) {
    using D = Dual<1>;
    visual = orbit_kind != OCTO_ORBIT_RADVEL && orbit_kind != OCTO_ORBIT_KEP;      // Visual{KepOrbit} or ThieleInnesOrbit (hgca.jl:255-262)
    const bool ti = orbit_kind == OCTO_ORBIT_THIELE_INNES;
    D el[OCTO_N_EL];
#pragma unroll
    for (int k = 0; k < OCTO_N_EL; ++k) el[k] = (dirl == k) ? dvar<1>(elv[k], 0) : dconst<1>(elv[k]);
    if (!has_mass) el[OCTO_EL_MASS] = dconst<1>(0.0);
    const D e = el[OCTO_EL_E], Mt = el[OCTO_EL_M];
    // The branch below fills LOCALS, and h is written once, outside of it: with `h.T = …` in both arms the optimizer sinks the
    // two stores into the join block with a phi of the two ADDRESSES (h is still indexed by the caller's loop variable at that point), and
    // after the loop is unrolled that phi keeps every such field in scratch memory (40 bytes per planet in rounds 2-3: the
    // `.private_segment_fixed_size` of k_hgca<P >= 2> and of every k_small<P >= 2, NUIS> variant).
    D sma, T, cA, cB, cF, cG;
    if (ti) {
        // constants in mas; a = α/plx   (src/parameterizations.jl:14-19)
        cA = el[OCTO_EL_TI_A]; cB = el[OCTO_EL_TI_B]; cF = el[OCTO_EL_TI_F]; cG = el[OCTO_EL_TI_G];
        const D pp = ((cA + cG) * (cA + cG) + (cB - cF) * (cB - cF)) * 0.5;      // u + v, u − v as sums of squares
        const D mm = ((cA - cG) * (cA - cG) + (cB + cF) * (cB + cF)) * 0.5;      // (see setup_planet_vals)
        sma = ((dsqrt(pp) + dsqrt(mm)) * 0.70710678118654752440) / el[OCTO_EL_PLX];
        T = dconst<1>(1.0);
    } else {
        D inc = el[OCTO_EL_I], Om = el[OCTO_EL_O];
        inc.v = inc.v - PI * floor(inc.v / PI);                // KepOrbit ctor invariants, as in k_setup
        Om.v = Om.v - TWO_PI * floor(Om.v / TWO_PI);
        sma = el[OCTO_EL_A];
        T = sma * el[OCTO_EL_PLX] * c.mas_per_au_per_plx;     // parameterizations.jl:215-216
        const D ci = dcos(inc), sw = dsin(el[OCTO_EL_W]), cw = dcos(el[OCTO_EL_W]), sO = dsin(Om), cO = dcos(Om);
        cA = cO * cw - sO * sw * ci; cB = sO * cw + cO * sw * ci;
        cF = -(cO * sw) - sO * cw * ci; cG = -(sO * sw) + cO * cw * ci;
    }
    h.T = T; h.cA = cA; h.cB = cB; h.cF = cF; h.cG = cG;
    const D P_d = dsqrt(sma * sma * sma / Mt) * c.k_yr;   // parameterizations.jl:62
    h.e = e; h.tp = el[OCTO_EL_TP];
    h.beta = dsqrt(dconst<1>(1.0) - e * e);
    h.n_day = dconst<1>(TWO_PI) / P_d;
    h.fac = -(el[OCTO_EL_MASS] * c.mjup2msol) / Mt;      // q(sol, M_planet) = −M_planet/M_tot · q(sol)
    h.pc = PC{};
    h.pc.invP = 1.0 / P_d.v; h.pc.tp = h.tp.v; h.pc.e = e.v;
    set_starter<false>(h.pc, (float)e.v, (float)(1.0 - e.v), (float)(MK_K1N / (1.0 + e.v)));
}

// The companions' constants at walker wl with input direction `dir` seeded.
template <int P>
__device__ __forceinline__ void hgca_setup(const double (&elv)[P][OCTO_N_EL], const int32_t (&orbit_kind)[MAXP], const int32_t (&has_mass)[MAXP], const DevConsts& c,
                                           int dir, HgcaPlanet (&hp)[P], bool (&visual)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) hgca_setup_one(elv[p], orbit_kind[p], has_mass[p], c, dir - p * OCTO_N_EL, hp[p], visual[p]);
}

// The observation's three 2-D Gaussians from the epoch-averaged positions and proper motions (hgca.jl:301-382).
__device__ __forceinline__ Dual<1> hgca_terms(Dual<1> (&pos)[2][2], Dual<1> (&pm)[2][2], double (&ep)[2][2], const int (&cnt)[2][2],
                                              const Dual<1> (&pm_sys)[2], double yd, const double* __restrict__ x /* the 15 catalogue numbers */) {
    using D = Dual<1>;
    D model[3][2];
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {                       // :301-308, 353-360
            const double ic = 1.0 / (double)cnt[m][ax];
            pos[m][ax] = pos[m][ax] * ic; pm[m][ax] = pm[m][ax] * ic + pm_sys[ax]; ep[m][ax] *= ic;
        }
        model[0][ax] = pm[0][ax];
        model[1][ax] = (pos[1][ax] - pos[0][ax]) * (yd / (ep[1][ax] - ep[0][ax])) + pm_sys[ax];   // :379-382
        model[2][ax] = pm[1][ax];
    }
    D ll = dconst<1>(0.0);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        ll = ll + hgca_logpdf2(model[k][0] + (-x[5 * k]), model[k][1] + (-x[5 * k + 1]), x[5 * k + 2], x[5 * k + 3], x[5 * k + 4]);
    return ll;
}

// grid = (walker tiles of 64, P·9 + n_obs·3 input directions), block = 64
template <int P>
static __global__ __launch_bounds__(64) void k_hgca(EvalArgs a) {
    using D = Dual<1>;
    const int64_t w = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int dir = blockIdx.y;
    HgcaPlanet hp[P];
    bool visual[P];
    {
        double elv[P][OCTO_N_EL];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < OCTO_N_EL; ++k) elv[p][k] = a.elems[((int64_t)p * OCTO_N_EL + k) * a.ld + wl * a.ws_in];      // ws_in: 1, or the walker stride of k_small's staging
        hgca_setup<P>(elv, a.orbit_kind, a.has_mass, a.c, dir, hp, visual);
    }
    D ll = dconst<1>(0.0);
    for (int o = 0; o < a.n_obs; ++o) {
        const DevObs ob = a.obs[o];
        if (ob.kind != OCTO_HGCA) continue;
        D pm_sys[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double v = a.nuis[((int64_t)o * OCTO_N_NUIS + k) * a.ld + wl * a.ws_in];
            pm_sys[k] = (dir == P * OCTO_N_EL + o * OCTO_N_NUIS + k) ? dvar<1>(v, 0) : dconst<1>(v);
        }
        D pos[2][2], pm[2][2];
        double ep[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        int cnt[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) { pos[m][ax] = dconst<1>(0.0); pm[m][ax] = dconst<1>(0.0); }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (!visual[p]) continue;                           // hgca.jl:255-262
            for (int64_t j = 0; j < ob.n; ++j) {
                const double* rw = ob.raw + j * ROW_STRIDE;     // wave-uniform
                const double t = rw[0];
                const int ax = (int)rw[1], m = (int)rw[2];
                D q, v;
                hgca_solve(hp[p], t, ax, a.c.yd, q, v);
                // the counters and epoch sums advance once per (planet, row), as in the reference (:276-278)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa)
                        if (mm == m && aa == ax) { cnt[mm][aa] += 1; ep[mm][aa] += t; pos[mm][aa] = pos[mm][aa] + q; pm[mm][aa] = pm[mm][aa] + v; }
            }
        }
        ll = ll + hgca_terms(pos, pm, ep, cnt, pm_sys, a.c.yd, ob.pre);
    }
    if (w < a.W) {
        if (dir == 0) a.extra[w] = ll.v;
        a.extra[(int64_t)(1 + dir) * a.ldw + w] = ll.d[0];
    }
}

// The same for a system of more planets than k_hgca<P> is compiled for (round 6: the planet-per-wave kernels' systems, 5 … OCTO_MAX_PLANETS): the number of
// planets is a run-time loop bound and ONE planet's constants are live at a time — same sums in the same order (planet, then row) as k_hgca<P>.
static __global__ __launch_bounds__(64) void k_hgcap(EvalArgs a) {
    using D = Dual<1>;
    const int64_t w = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int dir = blockIdx.y;
    const int P = a.n_planets;
    D ll = dconst<1>(0.0);
    for (int o = 0; o < a.n_obs; ++o) {
        const DevObs ob = a.obs[o];
        if (ob.kind != OCTO_HGCA) continue;
        D pm_sys[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double v = a.nuis[((int64_t)o * OCTO_N_NUIS + k) * a.ld + wl];
            pm_sys[k] = (dir == P * OCTO_N_EL + o * OCTO_N_NUIS + k) ? dvar<1>(v, 0) : dconst<1>(v);
        }
        D pos[2][2], pm[2][2];
        double ep[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        int cnt[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) { pos[m][ax] = dconst<1>(0.0); pm[m][ax] = dconst<1>(0.0); }
#pragma unroll 1
        for (int p = 0; p < P; ++p) {
            double elv[OCTO_N_EL];
#pragma unroll
            for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = a.elems[((int64_t)p * OCTO_N_EL + k) * a.ld + wl];
            HgcaPlanet hp;
            bool visual;
            hgca_setup_one(elv, a.orbit_kind[p], a.has_mass[p], a.c, dir - p * OCTO_N_EL, hp, visual);
            if (!visual) continue;                              // hgca.jl:255-262 (wave-uniform: a planet's orbit kind)
            for (int64_t j = 0; j < ob.n; ++j) {
                const double* rw = ob.raw + j * ROW_STRIDE;     // wave-uniform
                const double t = rw[0];
                const int ax = (int)rw[1], m = (int)rw[2];
                D q, v;
                hgca_solve(hp, t, ax, a.c.yd, q, v);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa)
                        if (mm == m && aa == ax) { cnt[mm][aa] += 1; ep[mm][aa] += t; pos[mm][aa] = pos[mm][aa] + q; pm[mm][aa] = pm[mm][aa] + v; }
            }
        }
        ll = ll + hgca_terms(pos, pm, ep, cnt, pm_sys, a.c.yd, ob.pre);
    }
    if (w < a.W) {
        if (dir == 0) a.extra[w] = ll.v;
        a.extra[(int64_t)(1 + dir) * a.ldw + w] = ll.d[0];
    }
}

}  // namespace octo
