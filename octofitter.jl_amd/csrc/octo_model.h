// octo_model.h — standard parameterisation on the device: θ_t -> natural θ (Bijectors invlink,
// src/variables.jl:1449-1493), log-prior with Jacobian in declaration order (:1205-1369), UniformCircular angles and
// their UnitLengthPrior terms (:279-323), tp = θ_at_epoch_to_tperi (src/parameterizations.jl:6-69) -> the kernel's
// inputs, their Jacobian w.r.t. θ_t by forward-mode duals (exactly what ForwardDiff does on the host in the reference,
// src/logdensitymodel.jl:169-177 — this part has no epoch loop, it is O(W·D²)), then after the likelihood kernels
// ∇θ_t = Jᵀ ḡ + ∇(prior). One thread per walker; cost is negligible next to k_main.
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
constexpr int MODEL_NPART = 4;   // k_model_fwd: partials per thread. The value path (sincos, atan2, …) is the expensive part of a dual;
                                 // 4 partials per thread repeat it D/4 times per walker instead of D times

struct ModelArgs {
    const octo_prior* priors;       // [D]
    const double* prior_logz;       // [D][PRIOR_NC] constants of each prior (prior_apply)
    const octo_source* esrc;        // [n_el]
    const octo_source* nsrc;        // [n_nu] or null
    const DevObs* obs;
    int32_t D, n_el, n_nu, n_planets;
    int32_t src_waves, n_circ;      // k_model_fwd: waves per block that resolve sources (the block may hold more, for the priors);
                                    // number of UniformCircular pairs precomputed through LDS
    int32_t write_values, pad_wv;   // k_model_fwd<N > 0>: also store the kernel inputs and the prior sum (0: a k_model_fwd<0> launch does — the
                                    // Jacobian launch then runs beside the likelihood kernels, which read those values)
    const int32_t* circ_slot;       // [n_el + n_nu] LDS slot of a CIRCULAR / TPERI source's (atan2, UnitLength) values, or -1
    const double* theta_t; int64_t ld, W, ldw;
    double* elems; double* nuis;    // [n_el][ldw], [n_nu][ldw]   (kernel inputs)
    double* J;                      // [(n_el+n_nu)*D][ldw]
    double* lpp; double* glp;       // [ldw], [D][ldw]            (prior + UnitLength terms and their θ_t-gradient)
    const double* ll; const double* g_el; const double* g_nu;     // from the likelihood kernels
    double* lp_out; double* grad_out;
    double k_yr, yd;
};

// Bijectors.invlink + logpdf_with_trans (TruncatedBijector; Distributions densities) — mirrors oracle/octo_oracle_core.inc
// pc: four constants of the prior precomputed at model creation (octo_model_create: prior_consts) — {−log(Φ(hi) − Φ(lo)) of a truncated
// Normal, 1/(b − a), −log(b − a) | log(b/a) | −log σ, 1/σ} — so that the serial chain of a call holds no logarithm or division of constants.
constexpr int PRIOR_NC = 4;
DT void prior_apply(const octo_prior& pr, const DU& y, DU& x, DU& lp, const double* __restrict__ pc = nullptr) {
    double a = -INFINITY, b = INFINITY;
    if (pr.kind == OCTO_PRIOR_UNIFORM || pr.kind == OCTO_PRIOR_LOGUNIFORM) { a = pr.p0; b = pr.p1; }
    else if (pr.kind == OCTO_PRIOR_TRUNCNORMAL) { a = pr.lo; b = pr.hi; }
    else if (pr.kind == OCTO_PRIOR_SINE) { a = 0.0 + 2.220446049250313e-16; b = PI - 2.220446049250313e-16; }
    DU ladj;
    if (isfinite(a) && isfinite(b)) {
        const double sg = m_div<FAST>(1.0, 1.0 + exp(-y.v));
        x = chain(y, (b - a) * sg + a, (b - a) * sg * (1.0 - sg));
        ladj = dlog(((x + (-a)) * (dconst<N, FAST>(b) - x)) * (pc ? pc[1] : 1.0 / (b - a)));
    } else if (isfinite(a)) {
        const double ey = exp(y.v);
        x = chain(y, ey + a, ey);
        ladj = dlog(x + (-a));
    } else if (isfinite(b)) {
        const double ey = exp(y.v);
        x = chain(y, b - ey, -ey);
        ladj = dlog(dconst<N, FAST>(b) - x);
    } else {
        x = y; ladj = dconst<N, FAST>(0.0);
    }
    switch (pr.kind) {
        case OCTO_PRIOR_UNIFORM: lp = dconst<N, FAST>((x.v >= a && x.v <= b) ? (pc ? pc[2] : -log(b - a)) : -INFINITY); break;
        case OCTO_PRIOR_LOGUNIFORM: lp = (x.v >= a && x.v <= b) ? dlog(dconst<N, FAST>(1.0) / (x * (pc ? pc[2] : log(b / a)))) : dconst<N, FAST>(-INFINITY); break;
        case OCTO_PRIOR_NORMAL: case OCTO_PRIOR_TRUNCNORMAL: {
            const DU z = (x + (-pr.p0)) * (pc ? pc[3] : 1.0 / pr.p1);
            lp = (-(z * z + LOG2PI)) * 0.5 + (pc ? pc[2] : -log(pr.p1));
            if (pr.kind == OCTO_PRIOR_TRUNCNORMAL) {
                double nlz = pc ? pc[0] : NAN;
                if (isnan(nlz)) {
                    const double lo = isfinite(pr.lo) ? 0.5 * erfc(-((pr.lo - pr.p0) / pr.p1) * 0.70710678118654752440) : 0.0;
                    const double hi = isfinite(pr.hi) ? 0.5 * erfc(-((pr.hi - pr.p0) / pr.p1) * 0.70710678118654752440) : 1.0;
                    nlz = -log(hi - lo);
                }
                lp = lp + nlz;
                if (!(x.v >= pr.lo && x.v <= pr.hi)) lp = dconst<N, FAST>(-INFINITY);
            }
            break;
        }
        case OCTO_PRIOR_SINE: lp = (x.v > 0.0 && x.v < PI) ? dlog(dsin(x) * 0.5) : dconst<N, FAST>(-INFINITY); break;
        default: lp = dconst<N, FAST>(NAN);
    }
    lp = lp + ladj;
}

// The same for the lane = parameter mapping of k_small<MODEL>: every lane applies ITS OWN prior, so the kinds differ across the wave and the
// branches above run one after the other (five kinds: ~700 serial instructions, most of them the transcendentals of kinds a lane does
// not have). Branch-free instead — ONE exp, TWO logs and (only if some lane holds a Sine prior: a wave-uniform test) one sincos for all
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
// pre (k_small<MODEL>, wave-uniform arguments): {sin, cos} of Ω, ω, i and θ computed in one lane-batched pass (sincos_lanes) by the caller.
DT DU tperi(const DU& th, double theta_epoch, const DU& M, const DU& e, const DU& a_in,
            const DU& inc, const DU& w, const DU& O, double k_yr, double yd, bool ti = false, const DU* plx = nullptr,
            const double (*pre)[2] = nullptr) {
    DU A, B, F, G, a = a_in;
    if (ti) {
        A = a_in; B = inc; F = w; G = O;
        // a = α/plx, α = (√(u+v) + √(u−v))/√2 with u ± v as sums of squares (see setup_planet_vals)
        const DU pp = ((A + G) * (A + G) + (B - F) * (B - F)) * 0.5, mm = ((A - G) * (A - G) + (B + F) * (B + F)) * 0.5;
        a = ((dsqrt(pp) + dsqrt(mm)) * 0.70710678118654752440) / *plx;
    } else {
        DU cO, sO, cw, sw, ci;
        if (pre) {
            sO = chain(O, pre[0][0], pre[0][1]); cO = chain(O, pre[0][1], -pre[0][0]);
            sw = chain(w, pre[1][0], pre[1][1]); cw = chain(w, pre[1][1], -pre[1][0]);
            ci = chain(inc, pre[2][1], -pre[2][0]);
        } else {
            dsincos(O, sO, cO); dsincos(w, sw, cw);
            ci = dcos(inc);
        }
        A = cO * cw - sO * sw * ci; B = sO * cw + cO * sw * ci;
        F = -(cO * sw) - sO * cw * ci; G = -(sO * sw) + cO * cw * ci;
    }
    DU ct, st;
    if (pre) { st = chain(th, pre[3][0], pre[3][1]); ct = chain(th, pre[3][1], -pre[3][0]); }
    else dsincos(th, st, ct);
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

#ifdef OCTO_API_TU      // the non-template kernels are launched from octo_api.hip only: one copy in the library, not one per translation unit
// block = 64 walkers × DB partials (DB = min(D, 8) waves), grid = (walker tiles, ⌈D/DB⌉). Thread (w, d) carries the
// value and ONE partial (∂/∂θ_t[d]) of every quantity, so a wave is 64 walkers × one partial: uniform control flow,
// coalesced Jacobian rows, D× the parallelism of a thread-per-walker layout. The diagonal part — invlink and
// logpdf_with_trans of every prior — is computed once per walker by the block's waves (prior k by wave k mod DB) and
// shared through LDS: x[k], dx/dθ_t[k], p[k], dp/dθ_t[k]. Fast-math duals as in k_small<MODEL> (polynomial sincos / atan2,
// reciprocal-multiply divisions): the chain is a few thousand serial instructions per wave, 30 µs at 1e4 walkers with the ocml routines.
// N = MODEL_NPART: values and Jacobian; N = 0: the VALUES alone (kernel inputs, prior sum) — all a forward-only callback needs, and what the
// likelihood kernels of a gradient callback wait for (the Jacobian launch runs beside them on a second stream, octo_model_logpost_device).
template <int N>
static __global__ __launch_bounds__(512) void k_model_fwd(ModelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];      // [4][D][64]
    const int lane = threadIdx.x;
    const int wy = threadIdx.y, DB = blockDim.y;
    const int64_t w = (int64_t)blockIdx.x * WAVE + lane;
    const int64_t wl = w < a.W ? w : a.W - 1;
    const int DBs = min(DB, a.src_waves);           // waves that go on to resolve the sources; all DB waves share the priors
    const int d0 = (blockIdx.y * DBs + wy) * N;     // this thread's partials: ∂/∂θ_t[d0 .. d0+N)
    const int D = a.D;
    double* Lx = lds; double* Ldx = lds + (int64_t)D * WAVE; double* Lp = lds + 2 * (int64_t)D * WAVE; double* Ldp = lds + 3 * (int64_t)D * WAVE;
    for (int k = wy; k < D; k += DB) {
        Dual<1, true> xk, p;
        prior_apply(a.priors[k], dvar<1, true>(a.theta_t[(int64_t)k * a.ld + wl], 0), xk, p, a.prior_logz + PRIOR_NC * k);
        Lx[k * WAVE + lane] = xk.v; Ldx[k * WAVE + lane] = xk.d[0]; Lp[k * WAVE + lane] = p.v; Ldp[k * WAVE + lane] = p.d[0];
    }
    __syncthreads();
    // UniformCircular pairs (variables.jl:279-323): angle atan(y, x), its two partials, and the UnitLengthPrior term with its two
    // partials depend on two natural parameters only. One wave computes them per pair, for all the threads of the walker — in
    // place they were a ~350-instruction serial chain of atan2, sqrt and two logs, repeated by every thread.
    double* Lc = lds + 4 * (int64_t)D * WAVE;           // [n_circ][6][64]
    for (int k = wy; k < a.n_el + a.n_nu; k += DB) {
        const int slot = a.circ_slot[k];
        if (slot < 0) continue;
        const octo_source sc = k < a.n_el ? a.esrc[k] : a.nsrc[k - a.n_el];
        const Dual<2, true> cx = dvar<2, true>(Lx[sc.i0 * WAVE + lane], 0), cy = dvar<2, true>(Lx[sc.i1 * WAVE + lane], 1);
        const Dual<2, true> ang = datan2(cy, cx), ul = unit_length(cx, cy);
        double* o = Lc + (int64_t)slot * 6 * WAVE + lane;
        o[0] = ang.v; o[WAVE] = ang.d[0]; o[2 * WAVE] = ang.d[1]; o[3 * WAVE] = ul.v; o[4 * WAVE] = ul.d[0]; o[5 * WAVE] = ul.d[1];
    }
    __syncthreads();
    if (w >= a.W || wy >= DBs || d0 >= D) return;
    bool finite_in = true;
    for (int k = 0; k < D; ++k) finite_in = finite_in && isfinite(a.theta_t[(int64_t)k * a.ld + w]);   // logdensitymodel.jl:120-124
    Dual<N, true> lp = dconst<N, true>(0.0);
    Dual<N, true> ulp = dconst<N, true>(0.0);      // Σ UnitLengthPrior terms: likelihood terms of the reference (variables.jl:309-323), so they and
                                       // their gradient survive a healed prior (the healed value is a constant, :1229-1236)
    bool healed = false;
    for (int k = 0; k < D; ++k) {
        const double pv = Lp[k * WAVE + lane];
        if (!healed) {
            if (!isfinite(pv)) { lp = dconst<N, true>(-1.7976931348623157e308); healed = true; }     // variables.jl:1229-1236
            else {
                lp.v += pv;
#pragma unroll
                for (int j = 0; j < N; ++j) if (k == d0 + j) lp.d[j] += Ldp[k * WAVE + lane];
            }
        }
    }
    auto nat = [&](int k) {      // natural-domain θ[k] with this thread's partial
        Dual<N, true> xk; xk.v = Lx[k * WAVE + lane];
        const double dx = Ldx[k * WAVE + lane];
#pragma unroll
        for (int j = 0; j < N; ++j) xk.d[j] = (k == d0 + j) ? dx : 0.0;
        return xk;
    };
    // Kernel inputs from the natural θ (arr2nt + Derived variables). The nine element rows of a planet are resolved with
    // compile-time positions so that they live in registers (a dynamically indexed array lands in scratch memory and made
    // this kernel 3x slower); θ_at_epoch_to_tperi comes last within a planet because it reads the planet's other elements.
    // atan(θy, θx) of a UniformCircular pair as a dual, adding its UnitLengthPrior term to lp when this source carries it
    auto circ_angle = [&](const octo_source& sc, int k) {
        const int slot = a.circ_slot[k];
        if (slot < 0) {                                               // beyond the LDS budget: in place
            const Dual<N, true> cx = nat(sc.i0), cy = nat(sc.i1);
            if (sc.flags & OCTO_SRC_FLAG_UNITLEN) ulp = ulp + unit_length(cx, cy);
            return datan2(cy, cx);
        }
        const double* c = Lc + (int64_t)slot * 6 * WAVE + lane;
        const double dx = Ldx[sc.i0 * WAVE + lane], dy = Ldx[sc.i1 * WAVE + lane];    // ∂x/∂θ_t, ∂y/∂θ_t (diagonal)
        Dual<N, true> ang; ang.v = c[0];
        const bool ul = (sc.flags & OCTO_SRC_FLAG_UNITLEN) != 0;
        if (ul) ulp.v += c[3 * WAVE];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double sx = (sc.i0 == d0 + j) ? dx : 0.0, sy = (sc.i1 == d0 + j) ? dy : 0.0;
            ang.d[j] = c[WAVE] * sx + c[2 * WAVE] * sy;
            if (ul) ulp.d[j] += c[4 * WAVE] * sx + c[5 * WAVE] * sy;
        }
        return ang;
    };
    auto plain = [&](const octo_source& sc, int k) {      // OCTO_SRC_CONST / _THETA / _CIRCULAR
        if (sc.kind == OCTO_SRC_CONST) return dconst<N, true>(sc.value);
        if (sc.kind == OCTO_SRC_THETA) return nat(sc.i0);
        return circ_angle(sc, k) * (sc.value / TWO_PI);               // atan(y, x) / 2π * domain, variables.jl:284
    };
    const bool store_values = (N == 0) || a.write_values != 0;
    auto emit = [&](int k, const Dual<N, true>& val) {
        if (d0 == 0 && store_values) {
            double* dst = k < a.n_el ? a.elems + (int64_t)k * a.ldw + w : a.nuis + (int64_t)(k - a.n_el) * a.ldw + w;
            *dst = val.v;
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (d0 + j < D) a.J[((int64_t)k * D + d0 + j) * a.ldw + w] = val.d[j];
    };
    for (int p = 0; p < a.n_planets; ++p) {
        Dual<N, true> el[OCTO_N_EL];
#pragma unroll
        for (int j = 0; j < OCTO_N_EL; ++j) el[j] = dconst<N, true>(0.0);
#pragma unroll 1
        for (int kk = 0; kk < OCTO_N_EL; ++kk) {          // one copy of the code; the store is a select chain, not an indexed write
            const octo_source sc = a.esrc[p * OCTO_N_EL + kk];
            if (sc.kind == OCTO_SRC_TPERI) continue;
            const Dual<N, true> val = plain(sc, p * OCTO_N_EL + kk);
#pragma unroll
            for (int j = 0; j < OCTO_N_EL; ++j) {
                el[j].v = (j == kk) ? val.v : el[j].v;
#pragma unroll
                for (int q = 0; q < N; ++q) el[j].d[q] = (j == kk) ? val.d[q] : el[j].d[q];
            }
        }
        {   // tp = θ_at_epoch_to_tperi(...) is the one derived element of the standard parameterisation
            const octo_source sc = a.esrc[p * OCTO_N_EL + OCTO_EL_TP];
            if (sc.kind == OCTO_SRC_TPERI) {
                el[OCTO_EL_TP] = tperi(circ_angle(sc, p * OCTO_N_EL + OCTO_EL_TP), sc.value, el[OCTO_EL_M], el[OCTO_EL_E], el[OCTO_EL_A], el[OCTO_EL_I], el[OCTO_EL_W],
                                       el[OCTO_EL_O], a.k_yr, a.yd, (sc.flags & OCTO_SRC_FLAG_TI) != 0, &el[OCTO_EL_PLX]);
            }
        }
#pragma unroll
        for (int kk = 0; kk < OCTO_N_EL; ++kk) emit(p * OCTO_N_EL + kk, el[kk]);
    }
    for (int k = 0; k < a.n_nu; ++k) {
        octo_source sc;
        if (a.nsrc) sc = a.nsrc[k];
        else {
            const int r = k % OCTO_N_NUIS; const int kind = a.obs[k / OCTO_N_NUIS].kind;
            sc.kind = OCTO_SRC_CONST; sc.i0 = sc.i1 = sc.flags = 0;
            sc.value = ((kind <= OCTO_ASTROM_SEPPA || kind == OCTO_ONEIL_RADEC || kind == OCTO_ONEIL_SEPPA) && r == OCTO_NU_PLATESCALE) ? 1.0 : 0.0;
        }
        emit(a.n_el + k, plain(sc, a.n_el + k));
    }
    if (d0 == 0 && store_values) a.lpp[w] = finite_in ? lp.v + ulp.v : -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (d0 + j < D) a.glp[(int64_t)(d0 + j) * a.ldw + w] = (healed ? 0.0 : lp.d[j]) + ulp.d[j];
}

// grid = (walker tiles of 256, D): thread (w, d) produces grad[d][w] = ∂(prior)/∂θ_t[d] + Σ_k J[k][d]·ḡ[k].
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
    const bool ok = isfinite(lp);
    const int n_in = a.n_el + a.n_nu;
    double g = a.glp[(int64_t)d * a.ldw + w];
    for (int k = 0; k < n_in; ++k) {
        const double gk = k < a.n_el ? a.g_el[(int64_t)k * a.ldw + w] : (a.g_nu ? a.g_nu[(int64_t)(k - a.n_el) * a.ldw + w] : 0.0);
        g = fma(a.J[((int64_t)k * a.D + d) * a.ldw + w], gk, g);
    }
    a.grad_out[(int64_t)d * a.ld + w] = ok ? g : 0.0;
}
#endif      // OCTO_API_TU
#undef DFOR
#undef DT
#undef DU

}  // namespace octo
