// octo_model.h — standard parameterisation on the device: θ_t -> natural θ (Bijectors invlink,
// src/variables.jl:1449-1493), log-prior with Jacobian in declaration order (:1205-1369), UniformCircular angles and
// their UnitLengthPrior terms (:279-323), tp = θ_at_epoch_to_tperi (src/parameterizations.jl:6-69) -> the kernel's
// inputs and their Jacobian w.r.t. θ_t (what ForwardDiff does on the host in the reference, src/logdensitymodel.jl:169-177; this part
// has no epoch loop), then after the likelihood kernels ∇θ_t = Jᵀ ḡ + ∇(prior).
//   k_small<MODEL> (octo_small.h): one-partial fast-math duals, lane = ∂/∂θ_t[d] — the Dual type and its operations below.
//   k_model_fwd (throughput path): the Jacobian in its sparse form — two entries per input + tp's closed-form gradient (tperi_campbell).
#pragma once
#include "octo_kernels.h"

namespace octo {

// FAST (k_small<MODEL>, the one-θ-per-call latency path): reciprocal-multiply divisions, rsqrt-based roots, polynomial
// sincos / atan2 (octo_device.h, octo_kernels.h) instead of the IEEE / ocml routines — the chain θ_t -> elements is a few thousand
// SERIAL instructions for one wave, and at ~10 cycles per dependent FP64 instruction that is most of the call's latency.
template <int N, bool FAST = false>
struct Dual {
    double v;
    double d[N];
};

#define DFOR _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_)
#define DT template <int N, bool FAST> __device__ __forceinline__
#define DU Dual<N, FAST>
template <bool FAST> __device__ __forceinline__ double m_div(double x, double y) { return FAST ? x * rcp_nr<2>(y) : x / y; }
template <bool FAST> __device__ __forceinline__ double m_sqrt(double x) { return FAST ? sqrt_fast(x) : sqrt(x); }
template <bool FAST> __device__ __forceinline__ void m_sincos(double x, double& s, double& c) { if constexpr (FAST) sincos_reduced(x, s, c); else sincos(x, &s, &c); }
template <bool FAST> __device__ __forceinline__ double m_atan2(double y, double x) { return FAST ? atan2_fast(y, x) : atan2(y, x); }

template <int N, bool FAST = false> __device__ __forceinline__ Dual<N, FAST> dconst(double x) { Dual<N, FAST> r; r.v = x; DFOR r.d[k_] = 0.0; return r; }
template <int N, bool FAST = false> __device__ __forceinline__ Dual<N, FAST> dvar(double x, int slot) { Dual<N, FAST> r = dconst<N, FAST>(x); DFOR r.d[k_] = (k_ == slot) ? 1.0 : 0.0; return r; }
DT DU operator+(const DU& a, const DU& b) { DU r; r.v = a.v + b.v; DFOR r.d[k_] = a.d[k_] + b.d[k_]; return r; }
DT DU operator-(const DU& a, const DU& b) { DU r; r.v = a.v - b.v; DFOR r.d[k_] = a.d[k_] - b.d[k_]; return r; }
DT DU operator-(const DU& a) { DU r; r.v = -a.v; DFOR r.d[k_] = -a.d[k_]; return r; }
DT DU operator*(const DU& a, const DU& b) { DU r; r.v = a.v * b.v; DFOR r.d[k_] = a.d[k_] * b.v + a.v * b.d[k_]; return r; }
DT DU operator/(const DU& a, const DU& b) {
    DU r; const double ib = m_div<FAST>(1.0, b.v); r.v = FAST ? a.v * ib : a.v / b.v; DFOR r.d[k_] = (a.d[k_] - r.v * b.d[k_]) * ib; return r;
}
DT DU operator*(const DU& a, double s) { DU r; r.v = a.v * s; DFOR r.d[k_] = a.d[k_] * s; return r; }
DT DU operator+(const DU& a, double s) { DU r = a; r.v = a.v + s; return r; }
DT DU chain(const DU& a, double fv, double df) { DU r; r.v = fv; DFOR r.d[k_] = a.d[k_] * df; return r; }
DT DU dsqrt(const DU& a) { const double s = m_sqrt<FAST>(a.v); return chain(a, s, m_div<FAST>(0.5, s)); }
DT DU dlog(const DU& a) { return chain(a, log(a.v), m_div<FAST>(1.0, a.v)); }
DT DU dsin(const DU& a) { double s, c; m_sincos<FAST>(a.v, s, c); return chain(a, s, c); }
DT DU dcos(const DU& a) { double s, c; m_sincos<FAST>(a.v, s, c); return chain(a, c, -s); }
DT void dsincos(const DU& a, DU& s, DU& c) {
    double sv, cv; m_sincos<FAST>(a.v, sv, cv);
    s = chain(a, sv, cv); c = chain(a, cv, -sv);
}
DT DU datan2(const DU& y, const DU& x) {
    DU r; r.v = m_atan2<FAST>(y.v, x.v); const double ih = m_div<FAST>(1.0, x.v * x.v + y.v * y.v);
    DFOR r.d[k_] = (x.v * y.d[k_] - y.v * x.d[k_]) * ih; return r;
}

constexpr int MODEL_MAXCIRC = 12;   // UniformCircular pairs whose atan2 / UnitLengthPrior values are shared through LDS (more: computed in place)

struct ModelArgs {
    const octo_prior* priors;       // [D]
    const double* prior_logz;       // [D][PRIOR_NC] constants of each prior (prior_density_lanes)
    const octo_source* esrc;        // [n_el]
    const octo_source* nsrc;        // [n_nu] or null
    const DevObs* obs;
    int32_t D, n_el, n_nu, n_planets;
    int32_t n_circ, pad_nc;         // number of UniformCircular pairs precomputed through LDS
    const int32_t* circ_slot;       // [n_el + n_nu] LDS slot of a CIRCULAR / TPERI source's (atan2, UnitLength) values, or -1
    const int32_t* circ_pair;       // [n_circ][2] (i0, i1) of each slot
    const double* theta_t; int64_t ld, W, ldw;
    double* elems; double* nuis;    // [n_el][ldw], [n_nu][ldw]   (kernel inputs)
    double* Jc;                     // [2·(n_el+n_nu)][ldw]  ∂input k/∂θ_t[i0_k], ∂input k/∂θ_t[i1_k] (for tp: through its own θ pair only)
    double* gtp;                    // [n_el][ldw]           ∂tp/∂(element) of the element's planet (0 where tp is not derived)
    double* lpp; double* glp;       // [ldw], [D][ldw]            (prior + UnitLength terms and their θ_t-gradient)
    const double* ll; const double* g_el; const double* g_nu;     // from the likelihood kernels
    double* lp_out; double* grad_out;
    double k_yr, yd;
};

// Bijectors.invlink + logpdf_with_trans (TruncatedBijector; Distributions densities) — mirrors oracle/octo_oracle_core.inc.
// pc: four constants of the prior precomputed at model creation (octo_model_create) — {−log(Φ(hi) − Φ(lo)) of a truncated
// Normal, 1/(b − a), −log(b − a) | log(b/a) | −log σ, 1/σ} — so that the serial chain of a call holds no logarithm or division of constants.
constexpr int PRIOR_NC = 4;

// Written for the lane = parameter mapping of k_small<MODEL>: every lane applies ITS OWN prior, so the kinds differ across the wave and
// branches on the kind would run one after the other (five kinds: ~700 serial instructions, most of them the transcendentals of kinds a lane
// does not have). Branch-free instead — ONE exp, TWO logs and (only if some lane holds a Sine prior: a wave-uniform test) one sincos for all
// kinds, everything else selects on per-lane flags. Same formulas, same constants (`pc`, octo_model_create), one-partial fast-math duals:
//   link       both bounds: x = a + (b − a)·σ(y), dx = (b − a)σ(1 − σ);  lower only: x = a + e^y;  upper only: x = b − e^y;  none: x = y
//   log|J|     log((x − a)(b − x)/(b − a)) | log(x − a) | log(b − x) | 0                                  — the first log
//   logpdf     Uniform: −log(b − a) · LogUniform: −log(x·log(b/a)) · Normal: −(z² + log 2π)/2 − log σ (+ truncation constant) · Sine: log(sin x / 2)
//                                                                                                         — the second log
// Split in two because only the LINK is on the critical path of a call (x feeds the elements, the epoch loop and the finish); the
// density and log|J| are needed at the very end, so k_small<MODEL> lets another wave of the block compute them meanwhile.
struct PriorBounds {
    double a, b;
    bool fa, fb, both;
};
__device__ __forceinline__ PriorBounds prior_bounds(const octo_prior& pr) {
    const int kind = pr.kind;
    const bool is_u = kind == OCTO_PRIOR_UNIFORM, is_lu = kind == OCTO_PRIOR_LOGUNIFORM, is_tn = kind == OCTO_PRIOR_TRUNCNORMAL, is_s = kind == OCTO_PRIOR_SINE;
    PriorBounds B;
    B.a = (is_u || is_lu) ? pr.p0 : (is_tn ? pr.lo : (is_s ? 2.220446049250313e-16 : -INFINITY));
    B.b = (is_u || is_lu) ? pr.p1 : (is_tn ? pr.hi : (is_s ? PI - 2.220446049250313e-16 : INFINITY));
    B.fa = isfinite(B.a); B.fb = isfinite(B.b); B.both = B.fa && B.fb;
    return B;
}

// Bijectors.invlink of this lane's prior: x and dx/dθ_t
__device__ __forceinline__ void prior_link_lanes(const octo_prior& pr, double y, double& xv, double& xd) {
    const PriorBounds B = prior_bounds(pr);
    const double em = exp(-fabs(y));                                // in (0, 1]: never overflows, whatever θ_t a sampler tries
    const double r1 = rcp_nr<2>(1.0 + em);
    const double sg = y >= 0.0 ? r1 : em * r1;                       // σ(y) = 1/(1 + e^−y)
    const double ey = y >= 0.0 ? rcp_nr<2>(em) : em;                 // e^y for the one-sided links
    const double ba = B.b - B.a;
    xv = B.both ? fma(ba, sg, B.a) : (B.fa ? ey + B.a : (B.fb ? B.b - ey : y));
    xd = B.both ? ba * sg * (1.0 - sg) : (B.fa ? ey : (B.fb ? -ey : 1.0));
}

// logpdf_with_trans of this lane's prior at the linked x: value and d/dθ_t
__device__ __forceinline__ void prior_density_lanes(const octo_prior& pr, double xv, double xd, double& lpv, double& lpd, const double* __restrict__ pc) {
    const int kind = pr.kind;
    const bool is_u = kind == OCTO_PRIOR_UNIFORM, is_lu = kind == OCTO_PRIOR_LOGUNIFORM, is_tn = kind == OCTO_PRIOR_TRUNCNORMAL, is_s = kind == OCTO_PRIOR_SINE;
    const bool is_n = kind == OCTO_PRIOR_NORMAL || is_tn;
    const PriorBounds B = prior_bounds(pr);
    const double a = B.a, b = B.b;
    const bool fa = B.fa, fb = B.fb, both = B.both;
    // log|J| and its derivative d/dθ_t = (d arg/dx · dx/dy)/arg
    const double xa = xv - a, bx = b - xv;
    const double jarg = both ? xa * bx * pc[1] : (fa ? xa : (fb ? bx : 1.0));
    const double jdarg = both ? (bx - xa) * pc[1] : (fa ? 1.0 : (fb ? -1.0 : 0.0));
    const double ladj = (fa || fb) ? log(jarg) : 0.0;
    const double ladj_d = (fa || fb) ? jdarg * xd * rcp_nr<2>(jarg) : 0.0;
    // the density
    double sx = 0.0, cx = 1.0;
    if (__any(is_s)) sincos_reduced(xv, sx, cx);
    const double larg = is_lu ? xv * pc[2] : (is_s ? 0.5 * sx : 1.0);
    const double ldarg = is_lu ? pc[2] : (is_s ? 0.5 * cx : 0.0);
    const double lg = (is_lu || is_s) ? log(larg) : 0.0;
    const double lg_d = (is_lu || is_s) ? ldarg * xd * rcp_nr<2>(larg) : 0.0;
    const double z = is_n ? (xv - pr.p0) * pc[3] : 0.0;
    double pv = is_u ? pc[2] : (is_lu ? -lg : (is_s ? lg : fma(-0.5, fma(z, z, LOG2PI), pc[2]) + (is_tn ? pc[0] : 0.0)));
    double pd = is_lu ? -lg_d : (is_s ? lg_d : (is_n ? -z * pc[3] * xd : 0.0));
    const bool inside = (is_u || is_lu) ? (xv >= a && xv <= b) : (is_tn ? (xv >= pr.lo && xv <= pr.hi) : (is_s ? (xv > 0.0 && xv < PI) : true));
    pv = inside ? pv : -INFINITY;
    pd = inside ? pd : 0.0;
    lpv = pv + ladj;
    lpd = pd + ladj_d;
}

// logpdf(LogNormal(log(1.0), 0.1), sqrt(x² + y²))   src/variables.jl:309-323
DT DU unit_length(const DU& x, const DU& y) {
    const DU r = dsqrt(x * x + y * y);
    const DU z = dlog(r) * 10.0;
    return (-(z * z + LOG2PI)) * 0.5 - dlog(r * 0.1);
}

// θ_at_epoch_to_tperi   src/parameterizations.jl:34-67
// Thiele-Innes planets (ti): the arguments a, inc, w, O carry A, B, F, G [mas] and a = α/plx (:14-19).
// (The reference-order route through dual numbers: Thiele-Innes planets, and the oracle's twin. Campbell planets: tperi_campbell below.)
DT DU tperi(const DU& th, double theta_epoch, const DU& M, const DU& e, const DU& a_in,
            const DU& inc, const DU& w, const DU& O, double k_yr, double yd, bool ti = false, const DU* plx = nullptr) {
    DU A, B, F, G, a = a_in;
    if (ti) {
        A = a_in; B = inc; F = w; G = O;
        // a = α/plx, α = (√(u+v) + √(u−v))/√2 with u ± v as sums of squares (see setup_planet_vals)
        const DU pp = ((A + G) * (A + G) + (B - F) * (B - F)) * 0.5, mm = ((A - G) * (A - G) + (B + F) * (B + F)) * 0.5;
        a = ((dsqrt(pp) + dsqrt(mm)) * 0.70710678118654752440) / *plx;
    } else {
        DU cO, sO, cw, sw;
        dsincos(O, sO, cO); dsincos(w, sw, cw);
        const DU ci = dcos(inc);
        A = cO * cw - sO * sw * ci; B = sO * cw + cO * sw * ci;
        F = -(cO * sw) - sO * cw * ci; G = -(sO * sw) + cO * cw * ci;
    }
    DU ct, st;
    dsincos(th, st, ct);
    const DU det = A * G - F * B;
    const DU xr = (G * ct - F * st) / det, yr = (A * st - B * ct) / det;
    const DU s1 = dsqrt(dconst<N, FAST>(1.0) - e * e);
    DU sn, cn;
    if constexpr (FAST) {
        // The reference takes ν = atan(yr, xr) and then sin ν, cos ν (parameterizations.jl:52-56); ν itself is not used again, and
        // sin ν = yr/r, cos ν = xr/r with r = hypot(xr, yr) are the same numbers without an arctangent followed by a sincos (~90 serial
        // instructions of the one-θ callback's chain; the partials follow from dν = (xr·dyr − yr·dxr)/r², as datan2 forms them).
        const double ih = rsqrt_nr(fma(xr.v, xr.v, yr.v * yr.v));
        cn.v = xr.v * ih; sn.v = yr.v * ih;
        DFOR {
            const double dnu = (xr.v * yr.d[k_] - yr.v * xr.d[k_]) * (ih * ih);
            sn.d[k_] = cn.v * dnu; cn.d[k_] = -sn.v * dnu;
        }
    } else {
        const DU nu = datan2(yr, xr);
        dsincos(nu, sn, cn);
    }
    const DU MA = datan2(-(s1 * sn), -e - cn) + PI - (e * s1 * sn) / (e * cn + 1.0);
    const DU period_yrs = dsqrt(a * a * a / M) * (k_yr / yd);
    const DU n = dconst<N, FAST>(TWO_PI) / period_yrs;
    return dconst<N, FAST>(theta_epoch) - (MA / n) * yd;
}

// θ_at_epoch_to_tperi of a Campbell planet in closed form, WITH its gradient w.r.t. the seven quantities it reads (throughput path,
// k_model_fwd). The reference inverts the Thiele-Innes matrix T = R(Ω)·diag(1, cos i)·R(ω) (parameterizations.jl:29-47), so
// (x, y)/r = R(−ω)·diag(1, 1/cos i)·R(−Ω)·(cos θ, sin θ): with c, s = cos, sin(θ − Ω),
//     cos ν = σ·(c·ci·cω + s·sω)/r',  sin ν = σ·(−c·ci·sω + s·cω)/r',  r'² = c²ci² + s²,  σ = sign(cos i)
// — three sincos and one rsqrt instead of four sincos, A/B/F/G, a determinant and its division. MA(ν, e) as the reference writes it (:57).
//     ∂ν/∂θ = −∂ν/∂Ω = ci/r'²   ∂ν/∂ω = −1   ∂ν/∂i = s·c·sin i/r'²
//     ∂MA/∂ν = (1−e²)^{3/2}/(1+e cos ν)²   ∂MA/∂e = −√(1−e²)·sin ν·(2 + e cos ν)/(1+e cos ν)²
//     tp = θ_epoch − MA·P_d/2π,  P_d = √(a³/M)·k_yr:  ∂tp/∂a = −MA·P_d/2π·(3/2)/a,  ∂tp/∂M = +MA·P_d/2π·(1/2)/M
// (checked against the reference's expression and its central differences at 40 digits: oracle/ has the script's twin in tests/test_model.py).
// g[OCTO_N_EL]: ∂tp/∂(element slot) — zero for tp, plx, mass; g_theta: ∂tp/∂θ.
// pre (k_small<MODEL>, wave-uniform arguments): {sin, cos} of θ − Ω, i, ω from one lane-batched pass (sincos_lanes) by the caller.
template <bool GRAD>
__device__ __forceinline__ double tperi_campbell(double th, double theta_epoch, double M, double e, double a, double inc, double w, double O,
                                                 double k_yr, double (&g)[OCTO_N_EL], double& g_theta, const double (*pre)[2] = nullptr) {
    double s, c, si, ci, sw, cw;
    if (pre) { s = pre[0][0]; c = pre[0][1]; si = pre[1][0]; ci = pre[1][1]; sw = pre[2][0]; cw = pre[2][1]; }
    else {
        sincos_reduced(th - O, s, c);
        sincos_reduced(inc, si, ci);
        sincos_reduced(w, sw, cw);
    }
    const double cci = c * ci;
    const double Xp = fma(cci, cw, s * sw), Yp = fma(s, cw, -(cci * sw));
    const double r2 = fma(cci, cci, s * s);
    const double ir = rsqrt_nr(r2);
    const double sg = ci == 0.0 ? NAN : copysign(ir, ci);      // cos i = 0: T is singular, the reference divides by zero
    const double cn = Xp * sg, sn = Yp * sg;
    const double s1 = sqrt_fast(fma(-e, e, 1.0));
    const double Dn = fma(e, cn, 1.0), iD = rcp_nr<2>(Dn);
    const double s1sn = s1 * sn;
    const double MA = atan2_fast(-s1sn, -e - cn) + PI - e * s1sn * iD;
    const double iM = rcp_nr<2>(M);
    const double Pd = sqrt_fast(a * a * a * iM) * k_yr;
    const double K = Pd * (-1.0 / TWO_PI);
    if constexpr (GRAD) {
        const double iD2 = iD * iD;
        const double Kn = K * (s1 * s1 * s1) * iD2;                 // ∂tp/∂ν
        const double ir2 = ir * ir;
        const double gth = Kn * ci * ir2;
        g_theta = gth;
        g[OCTO_EL_O] = -gth; g[OCTO_EL_W] = -Kn; g[OCTO_EL_I] = Kn * (s * c) * si * ir2;
        g[OCTO_EL_E] = -K * s1sn * fma(e, cn, 2.0) * iD2;
        g[OCTO_EL_A] = K * MA * 1.5 * rcp_nr<2>(a);
        g[OCTO_EL_M] = -K * MA * 0.5 * iM;
        g[OCTO_EL_TP] = 0.0; g[OCTO_EL_PLX] = 0.0; g[OCTO_EL_MASS] = 0.0;
    }
    return fma(K, MA, theta_epoch);
}

#ifdef OCTO_API_TU      // the non-template kernels are launched from octo_api.hip only: one copy in the library, not one per translation unit
// k_model_fwd — θ_t -> kernel inputs, prior, and the COMPACT Jacobian of the inputs (round 4; rounds 1-3: dense forward-mode duals, 4 partials
// per thread, (9P + 3·n_obs)·D Jacobian rows through memory: 26 µs at 1e4 walkers, D = 11).
// The map θ_t -> inputs is sparse: an input is a constant, ONE natural parameter, the angle of ONE UniformCircular pair, or tp =
// θ_at_epoch_to_tperi(pair; the planet's own a, e, i, ω, Ω, M) — so its Jacobian is two numbers per input (∂/∂θ_t[i0], ∂/∂θ_t[i1]: Jc)
// plus tp's gradient w.r.t. the planet's elements (gtp), folded into the element adjoints by the tail (model_tail / k_model_bwd):
//     ∇θ_t[d] = ∂prior/∂θ_t[d] + Σ_k ([i0_k = d]·Jc[2k] + [i1_k = d]·Jc[2k+1]) · (ḡ[k] + ḡ[tp]·gtp[k]).
// block = 64 walkers × NW waves (blockDim.y), one block per tile; the work is a short dependent chain per walker, spread over the waves:
//   phase 1  wave k: x[k] = invlink(θ_t[k]) and dx/dθ_t                                                  -> LDS
//   phase 2  one task per wave: atan(y, x) of a UniformCircular pair and its two partials | the pair's UnitLengthPrior term and its
//            partials | logpdf_with_trans of prior k and its derivative (needed only by the sums of phase 3)           -> LDS
//   phase 3  wave p: planet p's nine inputs, tp in closed form (tperi_campbell) with its gradient; wave P: nuisance inputs and the prior
//            sum (healing rule, variables.jl:1229-1236); the other waves: the rows of ∂(prior + UnitLength terms)/∂θ_t.
// TI: the instantiation for models with a Thiele-Innes planet (its tp keeps the dual-number route of `tperi`, two passes of four partials).
template <bool GRAD, bool TI>
static __global__ __launch_bounds__(TI ? 512 : 1024) void k_model_fwd(ModelArgs a) {
    constexpr int N2 = GRAD ? 2 : 0;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // [4][D][64] x, dx, p, dp | [n_circ][6][64]
    const int lane = threadIdx.x;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y), NW = blockDim.y;
    const int64_t w = (int64_t)blockIdx.x * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int D = a.D, P = a.n_planets;
    double* Lx = lds; double* Ldx = lds + (int64_t)D * WAVE; double* Lp = lds + 2 * (int64_t)D * WAVE; double* Ldp = lds + 3 * (int64_t)D * WAVE;
    double* Lc = lds + 4 * (int64_t)D * WAVE;
    // θ_t is read exactly ONCE per (walker, parameter): a mid-size host-buffer call hands this kernel the mapped pinned staging buffer itself
    // (a PCIe read under the first phase instead of a copy kernel in front of the launch)
    __shared__ unsigned char sfin[16][WAVE];      // wave wy: every θ_t[k] it linked is finite (logdensitymodel.jl:120-124)
    {
        bool fin = true;
        for (int k = wy; k < D; k += NW) {
            const double y = a.theta_t[(int64_t)k * a.ld + wl];
            fin = fin && isfinite(y);
            double xv, xd;
            prior_link_lanes(a.priors[k], y, xv, xd);
            Lx[k * WAVE + lane] = xv; Ldx[k * WAVE + lane] = xd;
        }
        sfin[wy][lane] = fin ? 1 : 0;
    }
    __syncthreads();
    // the six numbers of a UniformCircular pair: angle, ∂angle/∂x, ∂angle/∂y, UnitLength term, its two partials (variables.jl:279-323)
    auto pair_angle = [&](int i0, int i1, double& v, double& vx, double& vy) {
        const Dual<N2, true> cx = dvar<N2, true>(Lx[i0 * WAVE + lane], 0), cy = dvar<N2, true>(Lx[i1 * WAVE + lane], 1);
        const Dual<N2, true> ang = datan2(cy, cx);
        v = ang.v;
        if constexpr (GRAD) { vx = ang.d[0]; vy = ang.d[1]; }
    };
    auto pair_unitlen = [&](int i0, int i1, double& v, double& vx, double& vy) {
        const Dual<N2, true> cx = dvar<N2, true>(Lx[i0 * WAVE + lane], 0), cy = dvar<N2, true>(Lx[i1 * WAVE + lane], 1);
        const Dual<N2, true> ul = unit_length(cx, cy);
        v = ul.v;
        if constexpr (GRAD) { vx = ul.d[0]; vy = ul.d[1]; }
    };
    const int n_circ = a.n_circ, n2 = 2 * n_circ + D;
    for (int t = wy; t < n2; t += NW) {
        if (t < 2 * n_circ) {
            const int slot = t < n_circ ? t : t - n_circ;
            const int i0 = a.circ_pair[2 * slot], i1 = a.circ_pair[2 * slot + 1];
            double* o = Lc + ((int64_t)slot * 6 + (t < n_circ ? 0 : 3)) * WAVE + lane;
            double v = 0.0, vx = 0.0, vy = 0.0;
            if (t < n_circ) pair_angle(i0, i1, v, vx, vy); else pair_unitlen(i0, i1, v, vx, vy);
            o[0] = v;
            if constexpr (GRAD) { o[WAVE] = vx; o[2 * WAVE] = vy; }
        } else {
            const int k = t - 2 * n_circ;
            double lpv, lpd;
            prior_density_lanes(a.priors[k], Lx[k * WAVE + lane], Ldx[k * WAVE + lane], lpv, lpd, a.prior_logz + PRIOR_NC * k);
            Lp[k * WAVE + lane] = lpv; Ldp[k * WAVE + lane] = lpd;
        }
    }
    __syncthreads();
    const bool live = w < a.W;
    // an input that is not tp: value and its two Jacobian entries
    auto resolve = [&](const octo_source& sc, int k, double& val, double& ja, double& jb, double (&c6)[6]) {
        ja = 0.0; jb = 0.0; val = sc.value;                                         // OCTO_SRC_CONST
        if (sc.kind == OCTO_SRC_THETA) { val = Lx[sc.i0 * WAVE + lane]; if constexpr (GRAD) ja = Ldx[sc.i0 * WAVE + lane]; }
        else if (sc.kind == OCTO_SRC_CIRCULAR || sc.kind == OCTO_SRC_TPERI) {
            const int slot = a.circ_slot[k];
            if (slot >= 0) {
                const double* c = Lc + (int64_t)slot * 6 * WAVE + lane;
#pragma unroll
                for (int q = 0; q < 6; ++q) c6[q] = (GRAD || q == 0 || q == 3) ? c[q * WAVE] : 0.0;
            } else {                                                                // beyond the LDS budget: in place
                pair_angle(sc.i0, sc.i1, c6[0], c6[1], c6[2]); pair_unitlen(sc.i0, sc.i1, c6[3], c6[4], c6[5]);
            }
            if (sc.kind == OCTO_SRC_CIRCULAR) {                                     // atan(y, x) / 2π · domain, variables.jl:284
                const double scl = sc.value * (1.0 / TWO_PI);
                val = c6[0] * scl;
                if constexpr (GRAD) { ja = scl * c6[1] * Ldx[sc.i0 * WAVE + lane]; jb = scl * c6[2] * Ldx[sc.i1 * WAVE + lane]; }
            }
        }
    };
    auto put_input = [&](int k, double val, double ja, double jb) {
        if (!live) return;
        double* dst = k < a.n_el ? a.elems + (int64_t)k * a.ldw + w : a.nuis + (int64_t)(k - a.n_el) * a.ldw + w;
        *dst = val;
        if constexpr (GRAD) { a.Jc[(int64_t)(2 * k) * a.ldw + w] = ja; a.Jc[(int64_t)(2 * k + 1) * a.ldw + w] = jb; }
    };
    for (int role = wy; role <= P; role += NW) {
        if (role < P) {
            const int p = role;
            double el[OCTO_N_EL], c6[6];
#pragma unroll
            for (int kk = 0; kk < OCTO_N_EL; ++kk) {
                const int k = p * OCTO_N_EL + kk;
                const octo_source sc = a.esrc[k];
                double ja, jb;
                resolve(sc, k, el[kk], ja, jb, c6);
                if (sc.kind != OCTO_SRC_TPERI) put_input(k, el[kk], ja, jb);
            }
            const int ktp = p * OCTO_N_EL + OCTO_EL_TP;
            const octo_source sc = a.esrc[ktp];
            double g[OCTO_N_EL];      // ∂tp/∂(element slot)
#pragma unroll
            for (int q = 0; q < OCTO_N_EL; ++q) g[q] = 0.0;
            if (sc.kind == OCTO_SRC_TPERI) {      // the one derived element of the standard parameterisation; it reads the planet's other elements
                double ja, jb, dummy, g_theta = 0.0, tp = 0.0;
                resolve(sc, ktp, dummy, ja, jb, c6);
                bool done = false;
                if constexpr (TI) {
                    if (sc.flags & OCTO_SRC_FLAG_TI) {
                        // locals: θ, M, e, A, B, F, G, plx — two passes of four partials through the reference-order routine
                        constexpr int slot_of[8] = {-1, OCTO_EL_M, OCTO_EL_E, OCTO_EL_A, OCTO_EL_I, OCTO_EL_W, OCTO_EL_O, OCTO_EL_PLX};
#pragma unroll
                        for (int pass = 0; pass < (GRAD ? 2 : 1); ++pass) {
                            constexpr int NT = GRAD ? 4 : 0;
                            auto lv = [&](double v, int idx) { Dual<NT, true> r = dconst<NT, true>(v); if constexpr (GRAD) { if (idx / 4 == pass) r.d[idx % 4] = 1.0; } return r; };
                            const Dual<NT, true> plx = lv(el[OCTO_EL_PLX], 7);
                            const Dual<NT, true> r = tperi(lv(c6[0], 0), sc.value, lv(el[OCTO_EL_M], 1), lv(el[OCTO_EL_E], 2), lv(el[OCTO_EL_A], 3), lv(el[OCTO_EL_I], 4),
                                                           lv(el[OCTO_EL_W], 5), lv(el[OCTO_EL_O], 6), a.k_yr, a.yd, true, &plx);
                            tp = r.v;
                            if constexpr (GRAD) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int idx = pass * 4 + j;
                                    if (idx == 0) g_theta = r.d[j]; else g[slot_of[idx]] = r.d[j];
                                }
                            }
                        }
                        done = true;
                    }
                }
                if (!done) tp = tperi_campbell<GRAD>(c6[0], sc.value, el[OCTO_EL_M], el[OCTO_EL_E], el[OCTO_EL_A], el[OCTO_EL_I], el[OCTO_EL_W], el[OCTO_EL_O], a.k_yr, g, g_theta);
                if constexpr (GRAD) { ja = g_theta * c6[1] * Ldx[sc.i0 * WAVE + lane]; jb = g_theta * c6[2] * Ldx[sc.i1 * WAVE + lane]; }
                put_input(ktp, tp, ja, jb);
            }
            if constexpr (GRAD) {
                if (live) {
#pragma unroll
                    for (int q = 0; q < OCTO_N_EL; ++q) a.gtp[(int64_t)(p * OCTO_N_EL + q) * a.ldw + w] = g[q];
                }
            }
        } else {
            // nuisance inputs, then the prior sum: logdensitymodel.jl:120-133, variables.jl:1205-1236, 309-323
            double c6[6];
            for (int k = 0; k < a.n_nu; ++k) {
                octo_source sc;
                if (a.nsrc) sc = a.nsrc[k];
                else {
                    const int r = k % OCTO_N_NUIS; const int kind = a.obs[k / OCTO_N_NUIS].kind;
                    sc.kind = OCTO_SRC_CONST; sc.i0 = sc.i1 = sc.flags = 0;
                    sc.value = ((kind <= OCTO_ASTROM_SEPPA || kind == OCTO_ONEIL_RADEC || kind == OCTO_ONEIL_SEPPA) && r == OCTO_NU_PLATESCALE) ? 1.0 : 0.0;
                }
                double val, ja, jb;
                resolve(sc, a.n_el + k, val, ja, jb, c6);
                put_input(a.n_el + k, val, ja, jb);
            }
            bool finite_in = true, healed = false;
            double lp = 0.0;
            for (int q = 0; q < NW; ++q) finite_in = finite_in && sfin[q][lane] != 0;
            for (int k = 0; k < D; ++k) {
                const double pv = Lp[k * WAVE + lane];
                healed = healed || !isfinite(pv);
                lp += pv;
            }
            lp = healed ? -1.7976931348623157e308 : lp;
            // Σ UnitLengthPrior terms: likelihood terms of the reference, so they and their gradient survive a healed prior
            double ulp = 0.0;
            for (int k = 0; k < a.n_el + (a.nsrc ? a.n_nu : 0); ++k) {
                const octo_source sc = k < a.n_el ? a.esrc[k] : a.nsrc[k - a.n_el];
                if ((sc.kind != OCTO_SRC_CIRCULAR && sc.kind != OCTO_SRC_TPERI) || !(sc.flags & OCTO_SRC_FLAG_UNITLEN)) continue;
                const int slot = a.circ_slot[k];
                if (slot >= 0) ulp += Lc[((int64_t)slot * 6 + 3) * WAVE + lane];
                else { pair_unitlen(sc.i0, sc.i1, c6[3], c6[4], c6[5]); ulp += c6[3]; }
            }
            if (live) a.lpp[w] = finite_in ? lp + ulp : -INFINITY;
        }
    }
    if constexpr (GRAD) {
        // rows of ∂(prior + UnitLength terms)/∂θ_t: by the waves without a role when there are any
        const int first = NW > P + 1 ? P + 1 : 0, nh = NW > P + 1 ? NW - (P + 1) : NW;
        if (wy < first) return;
        bool healed = false;
        for (int k = 0; k < D; ++k) healed = healed || !isfinite(Lp[k * WAVE + lane]);
        for (int d = wy - first; d < D; d += nh) {
            double g = healed ? 0.0 : Ldp[d * WAVE + lane];
            const double dxd = Ldx[d * WAVE + lane];
            for (int k = 0; k < a.n_el + (a.nsrc ? a.n_nu : 0); ++k) {
                const octo_source sc = k < a.n_el ? a.esrc[k] : a.nsrc[k - a.n_el];
                if ((sc.kind != OCTO_SRC_CIRCULAR && sc.kind != OCTO_SRC_TPERI) || !(sc.flags & OCTO_SRC_FLAG_UNITLEN)) continue;
                if (sc.i0 != d && sc.i1 != d) continue;
                const int slot = a.circ_slot[k];
                double c6[6];
                if (slot >= 0) { c6[4] = Lc[((int64_t)slot * 6 + 4) * WAVE + lane]; c6[5] = Lc[((int64_t)slot * 6 + 5) * WAVE + lane]; }
                else pair_unitlen(sc.i0, sc.i1, c6[3], c6[4], c6[5]);
                if (sc.i0 == d) g = fma(c6[4], dxd, g);
                if (sc.i1 == d) g = fma(c6[5], dxd, g);
            }
            if (live) a.glp[(int64_t)d * a.ldw + w] = g;
        }
    }
}

// grid = (walker tiles of 256, D): thread (w, d) produces grad[d][w]; only behind a batch that k_small evaluated (no k_finish to carry the tail).
static __global__ __launch_bounds__(256) void k_model_bwd(ModelArgs a) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (w >= a.W) return;
    const double lpp = a.lpp[w], ll = a.ll[w];
    // ℓπcallback: non-finite θ_t or prior -> return it without the likelihood (logdensitymodel.jl:120-133)
    double lp = isfinite(lpp) ? lpp + ll : lpp;
    if (isnan(lp)) lp = -INFINITY;
    if (d == 0) a.lp_out[w] = lp;
    if (!a.grad_out) return;
    const double g = model_grad_row(d, w, a.n_planets, a.n_nu, a.esrc, a.nsrc, a.Jc, a.gtp, a.glp, a.ldw, a.g_el, a.g_nu, a.ldw);
    a.grad_out[(int64_t)d * a.ld + w] = isfinite(lp) ? g : 0.0;
}
#endif      // OCTO_API_TU
#undef DFOR
#undef DT
#undef DU

}  // namespace octo
