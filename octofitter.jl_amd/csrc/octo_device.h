// octo_device.h — device-side math of the epoch-loop likelihood path (gfx950, FP64 VALU).
//
// Lane = walker. Everything that depends only on the walker lives in VGPRs for the whole row
// loop; everything that depends only on the observation row is wave-uniform and arrives through
// scalar loads (SGPRs), so a row costs no vector-memory traffic at all.
//
// The Kepler solve follows the reference's solver (PlanetOrbits.jl `kepler_solver(MA, e, Markley())`,
// call site src/parameterizations.jl:340; provenance docs/src/kepler.md:15-19): Markley's cubic
// starter + one fifth-order correction — non-iterative, hence divergence-free across the 64 lanes.
// The projection uses the Thiele-Innes form the reference itself uses in ofti_linear_solve
// (src/parameterizations.jl:343-353): X = cos E − e, Y = √(1−e²) sin E, ra = cB X + cG Y,
// dec = cA X + cF Y — algebraically identical to orbitsolve's 2·atan(ν_fact·tan(E/2)) route
// (no tan/atan/second sincos), parity-checked against oracle/ to ≤1e-12.
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
    WC_K1,        // MK_K1N / (1+e)
    WC_CB, WC_CG, WC_CA, WC_CF,   // mas per unit (X, Y):  ra = CB·X + CG·Y,  dec = CA·X + CF·Y
    WC_K,         // RV semi-amplitude     [m/s]
    WC_COSW, WC_SINW,
    WC_MU,        // m_planet / M_tot  (0 when the planet declares no mass)
    WC_A,         // semi-major axis       [AU]
    NWC
};

struct PC {   // one planet's constants for one walker, in registers
    double invP, tp, e, beta, k1, cB, cG, cA, cF, K, cw, sw, mu, a;
};

struct KSol { double X, Y, sE, cE, invD; };

__device__ __forceinline__ void load_pc(PC& pc, const double* __restrict__ wc, int64_t ldw, int p, int64_t w) {
    const double* b = wc + (int64_t)p * NWC * ldw + w;
    pc.invP = b[WC_INVP * ldw]; pc.tp = b[WC_TP * ldw]; pc.e = b[WC_E * ldw]; pc.beta = b[WC_BETA * ldw];
    pc.k1 = b[WC_K1 * ldw]; pc.cB = b[WC_CB * ldw]; pc.cG = b[WC_CG * ldw]; pc.cA = b[WC_CA * ldw];
    pc.cF = b[WC_CF * ldw]; pc.K = b[WC_K * ldw]; pc.cw = b[WC_COSW * ldw]; pc.sw = b[WC_SINW * ldw];
    pc.mu = b[WC_MU * ldw]; pc.a = b[WC_A * ldw];
}

// Eccentric anomaly and the quantities every projection needs.
__device__ __forceinline__ KSol kepler_solve(double t, const PC& pc) {
    // mean anomaly reduced to [-π, π]: work in orbits, subtract the nearest integer (exact), scale.
    const double u = (t - pc.tp) * pc.invP;
    const double M = (u - rint(u)) * TWO_PI;
    const double e = pc.e;
    const double ome = 1.0 - e;
    // ---- Markley (1995) starter, eqs (20),(5),(9),(10),(14),(15)
    const double alpha = fma(pc.k1, PI - fabs(M), MK_K0);
    const double d = fma(alpha, e, 3.0 * ome);
    const double ad = alpha * d;
    const double M2 = M * M;
    const double q = fma(2.0 * ad, ome, -M2);
    const double r = M * fma(3.0 * ad, d - ome, M2);
    const double q2 = q * q;
    const double x = fabs(r) + sqrt(fma(q2, q, r * r));
    const double w = cbrt(x * x);
    const double E1 = (2.0 * r * w / fma(w, w + q, q2) + M) / d;
    // ---- one fifth-order correction, eqs (21)-(29)
    double s1, c1;
    sincos(E1, &s1, &c1);
    const double f2 = e * s1, f3 = e * c1;
    const double f0 = E1 - f2 - M;
    const double f1 = 1.0 - f3;
    const double d3 = -f0 / (f1 - f0 * f2 / (2.0 * f1));
    const double d4 = -f0 / (f1 + f2 * d3 * 0.5 + d3 * d3 * f3 * (1.0 / 6.0));
    const double d42 = d4 * d4;
    double d5 = -f0 / (f1 + d4 * f2 * 0.5 + d42 * f3 * (1.0 / 6.0) - d42 * d4 * f2 * (1.0 / 24.0));
    // M == 0 or e == 0: the reference returns M itself (early exit); the formulas above give the
    // same value up to rounding, the select keeps the exact reference result.
    const bool trivial = (M == 0.0) | (e == 0.0);
    const double E = trivial ? M : E1 + d5;
    KSol s;
    sincos(E, &s.sE, &s.cE);
    s.X = s.cE - e;
    s.Y = pc.beta * s.sE;
    s.invD = 1.0 / (1.0 - e * s.cE);
    return s;
}

// Reverse sweep through X = cE − e, Y = β sE, D = 1 − e cE and the Kepler root E(M, e).
// Inputs: adjoints of X, Y, D and the direct e-adjoint accumulated so far.
// Adds into ge (Σ ē), gm (Σ M̄), gt (Σ M̄·(t−tp)).
__device__ __forceinline__ void kepler_adjoint(const KSol& s, const PC& pc, double t, double Xb, double Yb, double Db,
                                               double eb_direct, double& ge, double& gm, double& gt) {
    const double cEb = Xb - pc.e * Db;
    const double sEb = pc.beta * Yb;
    const double Eb = sEb * s.cE - cEb * s.sE;
    const double Mb = Eb * s.invD;                                  // ∂E/∂M = 1/(1 − e cos E)
    const double eb = eb_direct - Xb - s.cE * Db - (pc.e / pc.beta) * s.sE * Yb + Mb * s.sE;   // ∂E/∂e = sin E/(1 − e cos E)
    ge += eb;
    gm += Mb;
    gt = fma(Mb, t - pc.tp, gt);
}

}  // namespace octo
