// octo_small.h — k_small: the SMALL-BATCH mapping of the path (W <= OCTO_SMALL_BATCH_MAX parameter sets per call — what NUTS, the
// slice sampler and Pigeons' explorers do: one θ per ℓπcallback / ∇ℓπcallback, src/logdensitymodel.jl:110-177).
// The whole evaluation is ONE launch, mapped the other way round from k_main — grid = (row tasks, walkers), lane = EPOCH:
//   * every block derives its walker's orbit constants itself (setup_planet_vals: wave-uniform; no k_setup launch, no `wc` round trip);
//   * its 256 lanes stride over the task's rows with per-lane row records, through the same row bodies as k_main;
//   * the running sums are reduced across the 64 lanes with DPP row shifts / row broadcasts (no LDS traffic), across the block's
//     four waves through LDS, in a fixed order — bit-reproducible run to run;
//   * with more than one task per walker, blocks publish their partial with device-scope (write-through) atomic stores and
//     bump a per-walker counter; the block that sees the last count sums the partials IN TASK ORDER (so the result does not
//     depend on which block that is) and runs the finish — no k_finish launch;
//   * an HGCA table (hgca.jl:155-400: no epoch loop, forward-mode partials) is evaluated by n_hblocks extra blocks per walker, one
//     input direction per WAVE with its lanes over the table's rows, published and counted like the row partials;
//   * inputs and outputs may live in mapped pinned host memory and completion is signalled through per-walker flags there,
//     so a host-buffer call is one launch and no copy engine, no stream synchronisation.
// MODEL = true fuses the standard parameterisation (octo_model.h) into the same launch: θ_t in, log-posterior and
// ∇θ_t out. There the 64 lanes of a wave carry the D <= 64 PARTIALS: lane d applies prior d (invlink, logpdf_with_trans) and then
// pushes ∂/∂θ_t[d] through the derived variables as a one-partial dual, so the whole Jacobian of (elements, nuisances) w.r.t.
// θ_t is one SIMD evaluation of the chain; the finishing wave forms ∇θ_t[d] = ∂prior/∂θ_t[d] + Σ_k J[k][d]·ḡ[k] in lane d.
#pragma once
#include "octo_model.h"
#include "octo_hgca.h"

namespace octo {

constexpr int SMALL_W = OCTO_SMALL_BATCH_MAX;
#ifndef OCTO_SMALL_WAVE_SETUP
#define OCTO_SMALL_WAVE_SETUP 1      // several planets: wave p derives planet p's orbit constants (0: every wave derives all of them, rounds 2-3)
#endif
constexpr int SMALL_TPB = 256;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of a wave, returned in every lane (wave-uniform). Inclusive scan within each row of 16 lanes by
// row_shr:1,2,4,8 (lanes without a source add 0), then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3:
// lane 63 holds the total; a fixed tree, so the rounding is the same every run.
__device__ __forceinline__ double wave_sum(double x) {
    x = dpp_add<0x111, 0xf>(x);
    x = dpp_add<0x112, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    x = dpp_add<0x142, 0xa>(x);
    x = dpp_add<0x143, 0xc>(x);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// ---- all running sums of a wave at once ---------------------------------------------------------------------------------------
// wave_sum reduces ONE value over the 64 lanes in 6 steps of 3 instructions; a gradient launch has 10-50 running sums, i.e. 200-1000
// instructions of pure reduction on a call whose whole budget is a few thousand. Reducing them TOGETHER costs about one step per value:
// at every level the lanes of a pair split the remaining values between them — the lane whose selector bit is 0 ends with the pair's
// sum of the first value of two, the other lane with the second — so the number of live values halves per level (recursive halving,
// "transpose while you reduce"), and once one value per lane is left the remaining levels are plain butterflies.
//   level 0  lane ^ 32   v_permlane32_swap (gfx950): exchanges the upper half of one register with the lower half of another —
//   level 1  lane ^ 16   v_permlane16_swap            one instruction per 32-bit half IS the transpose step, no selects
//   level 2  lane ^ 15   DPP row_mirror            } keep = sel ? b : a, send = sel ? a : b, sum = keep + dpp(send)
//   level 3  lane ^ 7    DPP row_half_mirror       }
//   level 4  lane ^ 2    DPP quad_perm [2,3,0,1]   }
//   level 5  lane ^ 1    DPP quad_perm [1,0,3,2]   }
// (the six masks are linearly independent over GF(2), so every lane's final value sums all 64 lanes exactly once). Every value goes
// through the SAME tree of additions, whatever else is reduced next to it and wherever its partial sums live: the value a forward-only
// launch returns is bit-identical to the one a gradient launch returns, and results are reproducible run to run. 10 sums: 49 instructions
// instead of 200; 40 sums: ~170 instead of 800.
typedef unsigned red_v2u __attribute__((ext_vector_type(2)));
template <int LEVEL>
__device__ __forceinline__ double red_dpp(double x) {
    constexpr int CTRL = LEVEL == 2 ? 0x140 : (LEVEL == 3 ? 0x141 : (LEVEL == 4 ? 0x4E : 0xB1));
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int LEVEL>
__device__ __forceinline__ double red_pair(double a, double b, bool sel) {
    if constexpr (LEVEL <= 1) {
        const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
        red_v2u lo, hi;
        if constexpr (LEVEL == 0) { lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false); hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false); }
        else { lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false); hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false); }
        return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    } else {
        const double keep = sel ? b : a, send = sel ? a : b;
        return keep + red_dpp<LEVEL>(send);
    }
}
template <int LEVEL>
__device__ __forceinline__ double red_single(double x) {
    if constexpr (LEVEL <= 1) return red_pair<LEVEL>(x, x, false);
    else return x + red_dpp<LEVEL>(x);
}
template <int LEVEL, int n, int NMAX>
__device__ __forceinline__ void red_levels(double (&v)[NMAX], int (&id)[NMAX], int lane) {
    if constexpr (LEVEL < 6) {
        const bool sel = (lane >> (5 - LEVEL)) & 1;      // the selector bit of this level: 5, 4, 3, 2, 1, 0
        constexpr int pairs = n / 2;
#pragma unroll
        for (int i = 0; i < pairs; ++i) {
            v[i] = red_pair<LEVEL>(v[2 * i], v[2 * i + 1], sel);
            id[i] = id[2 * i] + (int)sel * (id[2 * i + 1] - id[2 * i]);      // (written as arithmetic: `sel ? id[2i+1] : id[2i]` becomes a dynamically
                                                                             // indexed array, i.e. scratch memory)
        }
        if constexpr (n % 2 == 1) { v[pairs] = red_single<LEVEL>(v[n - 1]); id[pairs] = id[n - 1]; }
        red_levels<LEVEL + 1, pairs + n % 2, NMAX>(v, id, lane);
    }
}
// On return every lane holds the wave's total of ONE of the N values: `total` of value `index` (each value in at least one lane).
template <int N>
__device__ __forceinline__ void wave_sum_multi(const double (&x)[N], int lane, double& total, int& index) {
    static_assert(N >= 1 && N <= WAVE, "wave_sum_multi: one value per lane at most");
    double v[N];
    int id[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { v[k] = x[k]; id[k] = k; }
    red_levels<0, N, N>(v, id, lane);
    total = v[0]; index = id[0];
}

__device__ __forceinline__ void pc_from_setup(PC& pc, const double (&v)[NWC]) {
    pc.invP = v[WC_INVP]; pc.tp = v[WC_TP]; pc.e = v[WC_E]; pc.beta = v[WC_BETA]; pc.eob = v[WC_EOB];
    pc.cB = v[WC_CB]; pc.cG = v[WC_CG]; pc.cA = v[WC_CA]; pc.cF = v[WC_CF]; pc.K = v[WC_K]; pc.cw = v[WC_COSW]; pc.sw = v[WC_SINW];
    pc.mu = v[WC_MU]; pc.a = v[WC_A]; pc.cGb = v[WC_CGB]; pc.cFb = v[WC_CFB]; pc.cBe = v[WC_CBE]; pc.cAe = v[WC_CAE];
    const float2 fa = *reinterpret_cast<const float2*>(&v[WC_F32A]);
    const float2 fb = *reinterpret_cast<const float2*>(&v[WC_F32B]);
    set_starter<false>(pc, fa.x, fa.y, fb.x);      // wave-uniform here: the constants live in SGPRs
}

// ---- standard parameterisation, lane = partial -------------------------------------------------------------------------
using D1 = Dual<1, true>;      // one partial per lane, fast math (octo_model.h)
using D2 = Dual<2, true>;      // a UniformCircular pair's angle / UnitLengthPrior term with both partials
// This lane's own prior: natural value x[lane], dx/dθ_t[lane]. Other θ are fetched from their lanes.
struct LaneTheta {
    double xv, xd;
    int lane;
};

__device__ __forceinline__ D1 nat_theta(const LaneTheta& T, int k) {      // natural θ[k] with this lane's partial (k wave-uniform)
    D1 r;
    r.v = lane_value(T.xv, k);
    r.d[0] = (k == T.lane) ? T.xd : 0.0;
    return r;
}

// UniformCircular pairs (variables.jl:279-323), precomputed ONE PAIR PER LANE: lane j evaluates atan(y, x) and the UnitLengthPrior
// term of pair j with both partials — one pass of atan2 / sqrt / log for all pairs of the model instead of one per source.
struct CircTable {
    double ang, ang_x, ang_y, ul, ul_x, ul_y;      // this lane's pair: values and ∂/∂x, ∂/∂y
    int n;
};

// OCTO_SRC_CONST / _THETA / _CIRCULAR. A UniformCircular source that carries its UnitLengthPrior term (OCTO_SRC_FLAG_UNITLEN,
// variables.jl:309-323) adds it to `ul` when `count_ul`. `slot`: the source's entry in the pair table, or -1 (computed in place).
__device__ __forceinline__ D1 src_angle(const octo_source& sc, int slot, const CircTable& C, const LaneTheta& T, D1& ul, bool count_ul) {
    // (every CIRCULAR / TPERI source of a model that takes this launch has a slot: octo_model_logpost_device checks it. Computing
    // a pair in place here would put a copy of atan2 + sqrt + two logs at each of the ~15 inlined call sites — 100 KB of code
    // that a cold one-block launch has to fetch.)
    const bool want_ul = count_ul && (sc.flags & OCTO_SRC_FLAG_UNITLEN);
    // this lane's partial of (x, y): dx/dθ_t[lane] is non-zero only in the lanes that own x or y
    const double sx = (sc.i0 == T.lane) ? T.xd : 0.0, sy = (sc.i1 == T.lane) ? T.xd : 0.0;
    D1 ang;
    ang.v = lane_value(C.ang, slot);
    ang.d[0] = lane_value(C.ang_x, slot) * sx + lane_value(C.ang_y, slot) * sy;
    if (want_ul) {
        ul.v += lane_value(C.ul, slot);
        ul.d[0] += lane_value(C.ul_x, slot) * sx + lane_value(C.ul_y, slot) * sy;
    }
    return ang;
}

__device__ __forceinline__ D1 src_plain(const octo_source& sc, int slot, const CircTable& C, const LaneTheta& T, D1& ul, bool count_ul) {
    if (sc.kind == OCTO_SRC_CONST) return dconst<1, true>(sc.value);
    if (sc.kind == OCTO_SRC_THETA) return nat_theta(T, sc.i0);
    return src_angle(sc, slot, C, T, ul, count_ul) * (sc.value / TWO_PI);               // atan(y, x) / 2π * domain, variables.jl:284
}

// default nuisance source of row k of an observation when the model gives none (jitter 0, platescale 1, northangle 0 / offset 0)
__device__ __forceinline__ octo_source default_nuis_source(int obs_kind, int r) {
    octo_source sc;
    sc.kind = OCTO_SRC_CONST; sc.i0 = sc.i1 = sc.flags = 0;
    sc.value = ((obs_kind <= OCTO_ASTROM_SEPPA || obs_kind == OCTO_ONEIL_RADEC || obs_kind == OCTO_ONEIL_SEPPA) && r == OCTO_NU_PLATESCALE) ? 1.0 : 0.0;
    return sc;
}

// One parameter set per call (what NUTS does): its inputs travel INSIDE the kernel arguments — the runtime keeps those in device
// memory, so the block starts from a scalar load instead of a PCIe read of the mapped staging buffer (~1.3 µs of a 15 µs call).
// Layout = one walker's staging row: [elems (P·9) | nuis (n_obs·3)], or θ_t[D] for the fused model launch.
constexpr int SMALL_INL = 64;
struct SmallInline {
    int32_t n, pad;           // n = 0: inputs in memory as usual; pad = 1: a nuisance value is not finite (checked by the host)
    double v[SMALL_INL];
};

// The model's descriptors travel as ONE block of device memory (`blob`, built by octo_model_create: priors | prior constants | element
// sources | nuisance sources | pair slots | pairs) that every block copies into LDS with one load per thread: one memory round trip
// at the start of the call instead of a chain of a dozen dependent scalar loads through six separate allocations.
constexpr int SMALL_BLOB_MAX = 1024;      // doubles (8 KB of LDS); a model that needs more takes the throughput path
constexpr int SMALL_MX_NU = 192;          // nuisance values shared through LDS (64 observations x 3; only launches compiled with nuisances use them)
struct SmallModel {       // the part of the model k_small<MODEL> reads
    const double* blob;            // [blob_n] doubles; the priors (octo_prior[D]) sit at byte 0
    int32_t blob_n;
    int32_t off_logz;              // byte offsets inside the blob: [D][PRIOR_NC] constants of each prior (prior_density_lanes)
    int32_t off_esrc;              //   octo_source[n_el]
    int32_t off_nsrc;              //   octo_source[n_nu], or -1 = defaults
    int32_t off_cslot;             //   int32[n_el + n_nu] pair-table slot of each CIRCULAR / TPERI source, or -1
    int32_t off_cpair;             //   int32[n_circ][2] (i0, i1) of each slot
    int32_t n_circ, n_el;
    const double* theta_t;         // [W][D] walker-major (ld = 1) or [D][ld]
    int64_t ld_t, ws_t;            // θ_t[d * ld_t + w * ws_t]
    double* lp_out; double* grad_out; int64_t ld_o, ws_o;      // lp_out[w * ws_o], grad_out[d * ld_o + w * ws_o]
    int32_t D, n_nu;
    double k_yr, yd;
};

// The kernel-argument block as the ABI lays it out: by-value structs in declaration order at their natural alignment, i.e. exactly this
// C struct (checked against the `.offset` entries of the code object's metadata; the fixtures would fail on a mismatch). Keep it in step
// with k_small's parameter list. Lane d of the fused model launch loads θ_t[d] straight from it — one vector load next to the blob's,
// instead of 64 scalar values and a 63-step select chain.
struct SmallKernargs { EvalArgs a; SmallModel sm; int32_t* counters; uint64_t* done_flags; uint64_t seq; SmallInline inl; };

template <int P, bool GRAD, bool NUIS, int KM, bool MODEL>
static __global__ __launch_bounds__(SMALL_TPB) void k_small(EvalArgs a, SmallModel sm, int32_t* __restrict__ counters, uint64_t* done_flags, uint64_t seq,
                                                            SmallInline inl) {
    using L = Layout<P, GRAD, NUIS, KM>;
    constexpr int NACC = L::NACC;
    constexpr int NW = SMALL_TPB / WAVE;
    __shared__ double red[NW][NACC];
    __shared__ double tot[NACC];
    __shared__ int last_flag;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef OCTO_SMALL_TRACE
    unsigned long long tr[12]; int ntr = 0;
#define TRACE_POINT() tr[ntr++] = __builtin_readcyclecounter()
#else
#define TRACE_POINT()
#endif
    TRACE_POINT();
    const int64_t w = blockIdx.y;                       // this block's walker (wave-uniform)
    const int64_t wi = w * a.ws_in, wo = w * a.ws_out;  // its offset in the input and the output rows
    const int task = blockIdx.x;
    const int n_base = a.n_rblocks;      // blocks 0 .. n_base−1 take the row tasks (block b: tasks b, b + n_base, …); n_base .. : the HGCA term

    // ---- (MODEL) θ_t -> natural θ, priors, elements as one-partial duals: lane d carries ∂/∂θ_t[d].
    // What a one-θ call pays for is the LENGTH of one wave's instruction chain (~5 ns per serial FP64 instruction), and only part of this
    // section feeds the rest of the call: x = invlink(θ_t) -> the UniformCircular angles -> θ_at_epoch_to_tperi -> the elements. The
    // densities of the priors (two logarithms, a sincos) and the UnitLengthPrior terms (a root, two logarithms per pair) are needed at the
    // very end only. So the block's waves split the work — they run on different SIMDs, truly in parallel:
    //   every wave   the model's descriptors into LDS (one load per thread), θ_t[lane], the link x[lane], dx/dθ_t[lane]
    //   wave 0       atan(y, x) of every pair (one pair per lane), the elements, tp; publishes the element VALUES (and the nuisances')
    //   wave 1       logpdf_with_trans of every prior, their sum in declaration order with the healing rule, ∂/∂θ_t[lane]
    //   wave 2       the UnitLengthPrior term of every pair and their sum over the sources that carry one, ∂/∂θ_t[lane]
    // and meet at one barrier; the finishing wave picks the sums up from LDS at the end.
    LaneTheta T{0.0, 0.0, lane};
    D1 ul = dconst<1, true>(0.0);                       // Σ UnitLengthPrior terms (value and this lane's partial): from wave 2, through LDS — read at the finish
    CircTable CT{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0};
    bool finite_in = true;
    // What only the FINISH needs again is parked in LDS across the row loop instead of being held in registers (the 3- and 4-planet model
    // launches spilled up to 262 VGPRs in round 3): ∂(element)/∂θ_t[lane] of every element (wave 0 of a block computes them; the finishing
    // wave is a wave 0), and per planet the element values + the nine constants planet_finish takes (wave-uniform: one word each).
    constexpr int PARK_N = OCTO_N_EL + 9;
    __shared__ double park_fin[P][PARK_N];
    __shared__ double park_eld[(MODEL && GRAD) ? P * OCTO_N_EL * WAVE : 1];
    __shared__ double mblob[MODEL ? SMALL_BLOB_MAX : 1];
    __shared__ double mx_glp[MODEL ? WAVE : 1], mx_uld[MODEL ? WAVE : 1], mx_s[2];      // wave 1 -> finisher, wave 2 -> finisher, {lpp, Σ ul}
    __shared__ double mx_el[MODEL ? P * OCTO_N_EL : 1], mx_nu[MODEL ? SMALL_MX_NU : 1];   // wave 0 -> every wave: element and nuisance values
    const char* const mb = reinterpret_cast<const char*>(mblob);
    // a source record of the model, wave-uniform (every lane reads the same LDS words; the integers go to SGPRs)
    auto blob_src = [&](int off, int k) {
        octo_source sc = *reinterpret_cast<const octo_source*>(mb + off + k * (int)sizeof(octo_source));
        sc.kind = __builtin_amdgcn_readfirstlane(sc.kind); sc.i0 = __builtin_amdgcn_readfirstlane(sc.i0);
        sc.i1 = __builtin_amdgcn_readfirstlane(sc.i1); sc.flags = __builtin_amdgcn_readfirstlane(sc.flags);
        return sc;
    };
    auto blob_slot = [&](int k) { return __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int32_t*>(mb + sm.off_cslot + 4 * k)); };
    auto nuis_src = [&](int o, int r, int obs_kind) {
        return sm.off_nsrc >= 0 ? blob_src(sm.off_nsrc, o * OCTO_N_NUIS + r) : default_nuis_source(obs_kind, r);
    };
    if constexpr (MODEL) {
        const int D = sm.D;
        const int dl = lane < D ? lane : D - 1;
        for (int i = threadIdx.x; i < sm.blob_n; i += SMALL_TPB) mblob[i] = sm.blob[i];
        double y;
        if (inl.n > 0) {      // θ_t from the kernel arguments: lane d loads entry d from the argument block itself
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
            y = reinterpret_cast<const double*>(ka + offsetof(SmallKernargs, inl) + offsetof(SmallInline, v))[dl];
        } else {
            y = sm.theta_t[(int64_t)dl * sm.ld_t + w * sm.ws_t];
        }
        __syncthreads();      // the descriptors are in LDS
        finite_in = __all(isfinite(y));                                              // logdensitymodel.jl:120-124
        const octo_prior pr = *reinterpret_cast<const octo_prior*>(mb + dl * (int)sizeof(octo_prior));
        prior_link_lanes(pr, y, T.xv, T.xd);
        TRACE_POINT();      // (MODEL) priors linked
        const int nC = sm.n_circ < WAVE ? sm.n_circ : WAVE;
        const int jc = lane < nC ? lane : 0;
        const int32_t* cpair = reinterpret_cast<const int32_t*>(mb + sm.off_cpair);
        const int ci0 = nC > 0 ? cpair[2 * jc] : 0, ci1 = nC > 0 ? cpair[2 * jc + 1] : 0;
        if (wv == 1) {
            // logpdf_with_trans of prior `lane`, then the sum in declaration order, healing as the reference does (variables.jl:1229-1236)
            D1 pk;
            prior_density_lanes(pr, T.xv, T.xd, pk.v, pk.d[0], reinterpret_cast<const double*>(mb + sm.off_logz) + PRIOR_NC * dl);
            bool healed = false;
            double sum = 0.0;
            for (int k = 0; k < D; ++k) {
                const double pv = lane_value(pk.v, k);
                if (!healed) {
                    if (!isfinite(pv)) { sum = -1.7976931348623157e308; healed = true; }
                    else sum += pv;
                }
            }
            mx_glp[lane] = (healed || lane >= D) ? 0.0 : pk.d[0];
            if (lane == 0) mx_s[0] = sum;
        } else if (wv == 2) {
            // UnitLengthPrior term of pair `lane` (variables.jl:309-323) with both partials, then Σ over the sources that carry one, in the
            // order the elements are resolved in (a planet's plain sources, then its tp), then the nuisances'
            D1 acc = dconst<1, true>(0.0);
            if (nC > 0) {
                const D2 cx = dvar<2, true>(__shfl(T.xv, ci0), 0), cy = dvar<2, true>(__shfl(T.xv, ci1), 1);
                const D2 u2 = unit_length(cx, cy);
                auto add_ul = [&](const octo_source& sc, int k) {
                    if ((sc.kind == OCTO_SRC_CIRCULAR || sc.kind == OCTO_SRC_TPERI) && (sc.flags & OCTO_SRC_FLAG_UNITLEN)) {
                        const int slot = blob_slot(k);
                        const double sx = (sc.i0 == T.lane) ? T.xd : 0.0, sy = (sc.i1 == T.lane) ? T.xd : 0.0;
                        acc.v += lane_value(u2.v, slot);
                        acc.d[0] += lane_value(u2.d[0], slot) * sx + lane_value(u2.d[1], slot) * sy;
                    }
                };
                for (int p = 0; p < P; ++p) {
                    for (int k = 0; k < OCTO_N_EL; ++k) {
                        const octo_source sc = blob_src(sm.off_esrc, p * OCTO_N_EL + k);
                        if (sc.kind != OCTO_SRC_TPERI) add_ul(sc, p * OCTO_N_EL + k);
                    }
                    const octo_source sc = blob_src(sm.off_esrc, p * OCTO_N_EL + OCTO_EL_TP);
                    if (sc.kind == OCTO_SRC_TPERI) add_ul(sc, p * OCTO_N_EL + OCTO_EL_TP);
                }
                if (sm.off_nsrc >= 0)
                    for (int k = 0; k < sm.n_nu; ++k) add_ul(blob_src(sm.off_nsrc, k), sm.n_el + k);
            }
            mx_uld[lane] = acc.d[0];
            if (lane == 0) mx_s[1] = acc.v;
        } else if (wv == 0) {
            // atan(y, x) of the model's UniformCircular pairs, one per lane (variables.jl:279-299)
            CT.n = nC;
            if (nC > 0) {
                const D2 cx = dvar<2, true>(__shfl(T.xv, ci0), 0), cy = dvar<2, true>(__shfl(T.xv, ci1), 1);
                const D2 ang = datan2(cy, cx);
                CT.ang = ang.v; CT.ang_x = ang.d[0]; CT.ang_y = ang.d[1];
            }
            TRACE_POINT();      // (MODEL) UniformCircular angles
#pragma unroll
            for (int p = 0; p < P; ++p) {
                D1 elD[OCTO_N_EL];
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) {
                    const octo_source sc = blob_src(sm.off_esrc, p * OCTO_N_EL + k);
                    elD[k] = (sc.kind == OCTO_SRC_TPERI) ? dconst<1, true>(0.0) : src_plain(sc, blob_slot(p * OCTO_N_EL + k), CT, T, ul, false);
                }
                const octo_source sc = blob_src(sm.off_esrc, p * OCTO_N_EL + OCTO_EL_TP);
                if (sc.kind == OCTO_SRC_TPERI) {    // tp = θ_at_epoch_to_tperi(θ, epoch; M, e, a, i, ω, Ω | plx, A, B, F, G), parameterizations.jl:6-69
                    const D1 th = src_angle(sc, blob_slot(p * OCTO_N_EL + OCTO_EL_TP), CT, T, ul, false);
                    if (sc.flags & OCTO_SRC_FLAG_TI) {
                        elD[OCTO_EL_TP] = tperi(th, sc.value, elD[OCTO_EL_M], elD[OCTO_EL_E], elD[OCTO_EL_A], elD[OCTO_EL_I],
                                                elD[OCTO_EL_W], elD[OCTO_EL_O], sm.k_yr, sm.yd, true, &elD[OCTO_EL_PLX]);
                    } else {
                        // Campbell basis: tp in closed form with its gradient w.r.t. the planet's elements (octo_model.h: tperi_campbell) — the VALUES
                        // are wave-uniform, lane d only chains its own partials through the seven numbers. sin/cos of θ − Ω, i, ω in one pass,
                        // lane j taking angle j.
                        const double xs[3] = {th.v - elD[OCTO_EL_O].v, elD[OCTO_EL_I].v, elD[OCTO_EL_W].v};
                        double ss[3], cs[3];
                        sincos_lanes<3>(xs, ss, cs);
                        const double pre[3][2] = {{ss[0], cs[0]}, {ss[1], cs[1]}, {ss[2], cs[2]}};
                        double g[OCTO_N_EL], g_theta;
                        elD[OCTO_EL_TP].v = tperi_campbell<true>(th.v, sc.value, elD[OCTO_EL_M].v, elD[OCTO_EL_E].v, elD[OCTO_EL_A].v, elD[OCTO_EL_I].v,
                                                                 elD[OCTO_EL_W].v, elD[OCTO_EL_O].v, sm.k_yr, g, g_theta, pre);
                        double dtp = g_theta * th.d[0];
                        dtp = fma(g[OCTO_EL_A], elD[OCTO_EL_A].d[0], dtp); dtp = fma(g[OCTO_EL_E], elD[OCTO_EL_E].d[0], dtp);
                        dtp = fma(g[OCTO_EL_I], elD[OCTO_EL_I].d[0], dtp); dtp = fma(g[OCTO_EL_W], elD[OCTO_EL_W].d[0], dtp);
                        dtp = fma(g[OCTO_EL_O], elD[OCTO_EL_O].d[0], dtp); dtp = fma(g[OCTO_EL_M], elD[OCTO_EL_M].d[0], dtp);
                        elD[OCTO_EL_TP].d[0] = dtp;
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) mx_el[p * OCTO_N_EL + k] = elD[k].v;
                }
                if constexpr (GRAD) {      // this lane's partial of every element: parked for the finish (which a wave 0 runs)
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) park_eld[(p * OCTO_N_EL + k) * WAVE + lane] = elD[k].d[0];
                }
            }
            if constexpr (NUIS) {      // the nuisance VALUES, for every wave's row tasks (their partials: the finishing wave, below)
                for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) {
                    const octo_source sc = nuis_src(k / OCTO_N_NUIS, k % OCTO_N_NUIS, a.obs[k / OCTO_N_NUIS].kind);
                    const D1 v = src_plain(sc, sm.off_nsrc >= 0 ? blob_slot(sm.n_el + k) : -1, CT, T, ul, false);
                    if (lane == 0) mx_nu[k] = v.v;
                }
            }
        }
        __syncthreads();      // elements, nuisances, prior and UnitLength sums are in LDS (the sums are picked up at the finish)
    }

    if constexpr (MODEL) { TRACE_POINT(); }      // (MODEL) elements resolved (θ_at_epoch_to_tperi included)
    // ---- orbit constants of this walker (what k_setup would have written to `wc`)
    // One planet: every wave derives them itself (no exchange). Several planets: WAVE p derives planet p — the waves run on different
    // SIMDs, so the P orbit constructors (~300 dependent instructions each) run side by side instead of one after the other in every
    // wave — and lane 0 publishes the planet's NWC constants, validity flag and the finish's parked values through LDS; after ONE barrier
    // every wave reads all P sets back (wave-uniform LDS reads). Everything a wave reads of another planet comes out of LDS after that
    // barrier — nothing per-planet stays in a register array that only one wave has filled (the round-3 attempt).
    PC pc[P];
    bool ok = true;
    constexpr bool WAVE_SETUP = (P > 1) && OCTO_SMALL_WAVE_SETUP;
    __shared__ double park_pc[WAVE_SETUP ? P : 1][NWC];
    __shared__ int park_ok[WAVE_SETUP ? P : 1];
    static_assert(P <= SMALL_TPB / WAVE, "k_small: one wave per planet for the orbit constants");
    auto load_elv = [&](int p, double (&elv)[OCTO_N_EL]) {      // p: compile-time constant or wave-uniform
        if constexpr (MODEL) {
#pragma unroll
            for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = mx_el[p * OCTO_N_EL + k];      // wave 0's values, through LDS
        } else {
            if (inl.n > 0) {
                if constexpr (WAVE_SETUP) {      // (p is this wave's index: the values come from the argument block with a run-time offset)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
                    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
                    const double* iv = reinterpret_cast<const double*>(ka + offsetof(SmallKernargs, inl) + offsetof(SmallInline, v)) + p * OCTO_N_EL;
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = iv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = inl.v[p * OCTO_N_EL + k];
                }
            } else {
                const double* el = a.elems + (int64_t)p * OCTO_N_EL * a.ld + wi;
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = el[(int64_t)k * a.ld];
            }
        }
    };
    auto park_planet = [&](int p, const double (&elv)[OCTO_N_EL], const SetupOut& so) {
        // for the finish (and the HGCA block): [elements | sma, P_d, β, sin i, cos i, sin Ω, cos Ω, sin ω, cos ω]
#pragma unroll
        for (int k = 0; k < OCTO_N_EL; ++k) park_fin[p][k] = elv[k];
        double* q = &park_fin[p][OCTO_N_EL];
        q[0] = so.v[WC_A]; q[1] = 1.0 / so.v[WC_INVP]; q[2] = so.v[WC_BETA]; q[3] = so.v[WC_SINI]; q[4] = so.v[WC_COSI];
        q[5] = so.v[WC_SINO]; q[6] = so.v[WC_COSO]; q[7] = so.v[WC_SINW]; q[8] = so.v[WC_COSW];
    };
    if constexpr (WAVE_SETUP) {
        if (wv < P) {
            double elv[OCTO_N_EL];
            load_elv(wv, elv);
            int okind = a.orbit_kind[0], hmass = a.has_mass[0];      // select chains on the wave index: no run-time index into the argument arrays
#pragma unroll
            for (int p = 1; p < P; ++p) { okind = (wv == p) ? a.orbit_kind[p] : okind; hmass = (wv == p) ? a.has_mass[p] : hmass; }
            const SetupOut so = setup_planet_vals<true, true>(elv, a.c, okind, hmass);      // wave-uniform elements: lane-batched sincos
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < NWC; ++k) park_pc[wv][k] = so.v[k];
                park_ok[wv] = so.ok ? 1 : 0;
                park_planet(wv, elv, so);
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double v[NWC];
#pragma unroll
            for (int k = 0; k < NWC; ++k) v[k] = park_pc[p][k];
            pc_from_setup(pc[p], v);
            ok = ok && park_ok[p] != 0;
        }
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double elv[OCTO_N_EL];
            load_elv(p, elv);
            const SetupOut so = setup_planet_vals<true, true>(elv, a.c, a.orbit_kind[p], a.has_mass[p]);      // wave-uniform elements: lane-batched sincos
            pc_from_setup(pc[p], so.v);
            if (threadIdx.x == 0) park_planet(p, elv, so);
            ok = ok && so.ok;
        }
    }
    __syncthreads();      // the parked values are visible to every wave (the HGCA block's waves read them)
    auto parked_fp = [&](int p) {
        FinPC f;
        const double* q = &park_fin[p][OCTO_N_EL];
        f.sma = q[0]; f.P_d = q[1]; f.beta = q[2]; f.si = q[3]; f.ci = q[4]; f.sO = q[5]; f.cO = q[6]; f.sw = q[7]; f.cw = q[8];
        return f;
    };
    // nuisance rows of observation o: from memory, or (MODEL) from their sources
    auto nuis_of = [&](int o, int obs_kind, double (&nu)[OCTO_N_NUIS], D1* nuD, bool count_ul) {
        if constexpr (MODEL) {
            // values: resolved once by wave 0 of this block (above); with partials (nuD: the finishing wave, which is a wave 0): again
            // from the sources. The UnitLengthPrior terms of circular nuisance sources are in wave 2's sum already.
            (void)count_ul;
#pragma unroll
            for (int r = 0; r < OCTO_N_NUIS; ++r) {
                if (nuD) {      // (also the only route of a launch compiled without nuisances: wave 0 fills mx_nu only `if constexpr (NUIS)`)
                    nuD[r] = src_plain(nuis_src(o, r, obs_kind), sm.off_nsrc >= 0 ? blob_slot(sm.n_el + o * OCTO_N_NUIS + r) : -1, CT, T, ul, false);
                    nu[r] = nuD[r].v;
                } else {
                    nu[r] = mx_nu[o * OCTO_N_NUIS + r];
                }
            }
        } else {
            if (inl.n > 0) {
#pragma unroll
                for (int r = 0; r < OCTO_N_NUIS; ++r) nu[r] = inl.v[P * OCTO_N_EL + o * OCTO_N_NUIS + r];
            } else {
                const double* p = a.nuis + (int64_t)o * OCTO_N_NUIS * a.ld + wi;
#pragma unroll
                for (int r = 0; r < OCTO_N_NUIS; ++r) nu[r] = p[(int64_t)r * a.ld];
            }
        }
    };

    double acc[NACC];
    const int n_tasks = a.n_tasks;
    const int n_blocks = n_base + a.n_hblocks;      // blocks of this walker
    const bool multi = n_blocks > 1;                // more than one block works on this walker: counter + last-block finish
    const bool via_mem = multi || n_tasks > 1;      // the task sums reach the finish through `partials`
    TRACE_POINT();      // setup done
    for (int tq = task; tq < (task < n_base ? n_tasks : 0); tq += n_base) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
        const Task tk = a.tasks[tq];
        const DevObs ob = a.obs[tk.obs];
        LogProd lp;
        int my_rows = 0;
        const SinCosTab notab{nullptr, 0.0};
        const bool is_astrom = ob.kind == OCTO_ASTROM_RADEC || ob.kind == OCTO_ASTROM_SEPPA || ob.kind == OCTO_ONEIL_RADEC ||
                               ob.kind == OCTO_ONEIL_SEPPA;
        const double* __restrict__ rows = (NUIS ? ob.raw : ob.pre) + (int64_t)tk.row0 * ROW_STRIDE;
        double nu[OCTO_N_NUIS] = {0.0, 0.0, 0.0};
        if constexpr (NUIS) nuis_of(tk.obs, ob.kind, nu, nullptr, false);
        if (L::HAS_ASTROM && (!L::HAS_RV || is_astrom)) {
            const AstromCoef<P> co = astrom_coef_vals<P, GRAD, NUIS, KM>(nu[OCTO_NU_JITTER], nu[OCTO_NU_PLATESCALE], nu[OCTO_NU_NORTHANGLE], ob.kind, ob.planet, ob.has_cor, pc);
            for (int r = threadIdx.x; r < tk.nrows; r += SMALL_TPB) {
                const double* __restrict__ rw = rows + (int64_t)r * ROW_STRIDE;
                astrom_row<P, GRAD, NUIS, KM, false>(acc, lp, pc, co, rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], notab);
                ++my_rows;
            }
        }
        if (L::HAS_RV && !is_astrom) {
            RvCoef<P> co = rv_coef_vals<P, GRAD, NUIS, KM>(nu[OCTO_NU_RV_OFFSET], nu[OCTO_NU_RV_JITTER], nu[OCTO_NU_RV_TREND], nullptr, a.ldw, ob.kind, ob.planet, tk.obs, pc, 0);
            if constexpr (GRAD && L::HAS_MARG) {
                if (co.marg) {      // block-uniform
                    // Marginalised RV (rv-absolute-margin.jl:171-181): the adjoint of a row needs μ̂ = −B/(2A) of the WHOLE table. The
                    // planner gives such a table to one block (launch_small), which first runs the forward row body over it for
                    // A and B — what the throughput path does with a forward launch + k_marg — then the gradient pass below.
                    using LF = Layout<P, false, NUIS, KM>;
                    double accF[LF::NACC];
#pragma unroll
                    for (int k = 0; k < LF::NACC; ++k) accF[k] = 0.0;
                    LogProd lpF;
                    const RvCoef<P> coF = rv_coef_vals<P, false, NUIS, KM>(nu[OCTO_NU_RV_OFFSET], nu[OCTO_NU_RV_JITTER], nu[OCTO_NU_RV_TREND], nullptr, a.ldw, ob.kind, ob.planet, tk.obs, pc, 0);
                    for (int r = threadIdx.x; r < tk.nrows; r += SMALL_TPB) {
                        const double* __restrict__ rw = rows + (int64_t)r * ROW_STRIDE;
                        rv_row<P, false, NUIS, KM, false>(accF, lpF, pc, coF, rw[0], rw[1], rw[2], NUIS ? rw[3] : 0.0, notab);
                    }
                    const double wA = wave_sum(accF[LF::OFF_MARG + 0]), wB = wave_sum(accF[LF::OFF_MARG + 1]);
                    if (lane == 0) { red[wv][0] = wA; red[wv][1] = wB; }
                    __syncthreads();
                    double A = red[0][0], B = red[0][1];
#pragma unroll
                    for (int q = 1; q < NW; ++q) { A += red[q][0]; B += red[q][1]; }
                    __syncthreads();      // `red` is reused by the block reduction below
                    co.mu_hat = -B / (2.0 * A);
                    co.iA = 1.0 / A;
                }
            }
            for (int r = threadIdx.x; r < tk.nrows; r += SMALL_TPB) {
                const double* __restrict__ rw = rows + (int64_t)r * ROW_STRIDE;
                rv_row<P, GRAD, NUIS, KM, false>(acc, lp, pc, co, rw[0], rw[1], rw[2], NUIS ? rw[3] : 0.0, notab);
                ++my_rows;
            }
        }
        if constexpr (NUIS) {
            double lg = lp.log_value();      // a lane without rows: log(1) = 0
            if ((KM & KM_MARG) && ob.kind == OCTO_RV_ABS_MARG) lg = fma((double)my_rows, LOG2PI, lg);
            acc[L::OFF_S] += lg;
        }
        // ---- lanes -> wave (DPP), waves -> block (LDS), fixed order; one partial per task
        {
            double rsum; int rid;
            wave_sum_multi<NACC>(acc, lane, rsum, rid);
            red[wv][rid] = rsum;      // lanes that hold the same sum store the same number
        }
        __syncthreads();
        if (threadIdx.x < NACC) {
            double x = red[0][threadIdx.x];
#pragma unroll
            for (int q = 1; q < NW; ++q) x += red[q][threadIdx.x];
            tot[threadIdx.x] = x;
            // device-scope atomic stores are written through, so no L2 write-back fence is needed
            if (via_mem) __hip_atomic_store(a.partials + ((int64_t)w * n_tasks + tq) * NACC + threadIdx.x, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if constexpr (NUIS && (KM & KM_HGCA) != 0) {
        if (task >= n_base) {
            // ---- HGCAInstantaneousObs (octo_hgca.h): wave wv of this block carries input direction d as a one-partial dual (what
            // k_hgca does with one thread per (walker, direction)); its lanes take the table's rows. The forward-only launch needs
            // the value alone: one wave, no partial.
            using DH = Dual<1>;
            const int n_dir = GRAD ? P * OCTO_N_EL + a.n_obs * OCTO_N_NUIS : 1;
            const int d = (task - n_base) * NW + wv;
            if (d < n_dir) {
                const int dir = GRAD ? d : -1;
                HgcaPlanet hp[P];
                bool visual[P];
                double elv[P][OCTO_N_EL];      // (thread 0 parked them; the barrier below the setup made them visible)
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) elv[p][k] = park_fin[p][k];
                hgca_setup<P>(elv, a.orbit_kind, a.has_mass, a.c, dir, hp, visual);
                DH llh = dconst<1>(0.0);
                for (int o = 0; o < a.n_obs; ++o) {
                    const DevObs ob = a.obs[o];
                    if (ob.kind != OCTO_HGCA) continue;
                    double nu[OCTO_N_NUIS];
                    nuis_of(o, ob.kind, nu, nullptr, false);
                    DH pm_sys[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) pm_sys[k] = (dir == P * OCTO_N_EL + o * OCTO_N_NUIS + k) ? dvar<1>(nu[k], 0) : dconst<1>(nu[k]);
                    DH pos[2][2], pm[2][2];
                    double ep[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, cn[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ax = 0; ax < 2; ++ax) { pos[m][ax] = dconst<1>(0.0); pm[m][ax] = dconst<1>(0.0); }
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        if (!visual[p]) continue;                           // hgca.jl:255-262
                        for (int64_t j = lane; j < ob.n; j += WAVE) {
                            const double* rw = ob.raw + j * ROW_STRIDE;
                            const double t = rw[0];
                            const int ax = (int)rw[1], m = (int)rw[2];
                            DH q, v;
                            hgca_solve(hp[p], t, ax, a.c.yd, q, v);
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                                for (int aa = 0; aa < 2; ++aa)
                                    if (mm == m && aa == ax) { cn[mm][aa] += 1.0; ep[mm][aa] += t; pos[mm][aa] = pos[mm][aa] + q; pm[mm][aa] = pm[mm][aa] + v; }
                        }
                    }
                    int cnt[2][2];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ax = 0; ax < 2; ++ax) {
                            pos[m][ax].v = wave_sum(pos[m][ax].v); pos[m][ax].d[0] = wave_sum(pos[m][ax].d[0]);
                            pm[m][ax].v = wave_sum(pm[m][ax].v); pm[m][ax].d[0] = wave_sum(pm[m][ax].d[0]);
                            ep[m][ax] = wave_sum(ep[m][ax]);
                            cnt[m][ax] = (int)wave_sum(cn[m][ax]);      // small integers: exact
                        }
                    llh = llh + hgca_terms(pos, pm, ep, cnt, pm_sys, a.c.yd, ob.pre);
                }
                if (lane == 0) {      // write-through stores, released with the block's counter increment below
                    if (d == 0) __hip_atomic_store(a.extra + w, llh.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (GRAD) __hip_atomic_store(a.extra + (int64_t)(1 + d) * a.ldw + w, llh.d[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    TRACE_POINT();      // rows done, block reductions done
    if (via_mem) {
        // Every storing thread waits for its stores to be acknowledged BEFORE the barrier (the __threadfence() pattern): the partial /
        // HGCA stores above and the counter atomic below go to different L2 channels, and a workgroup-scope release does not wait
        // for vmcnt on gfx950 — without this the block that sees the last count could read another block's partials of the
        // PREVIOUS call (`partials` is reused). ISA: s_waitcnt vmcnt(0) ahead of s_barrier and global_atomic_add
        // (OCTO_SMALL_FULL_FENCE: the formal agent-scope release, buffer_wbl2 sc1 + the same wait).
#ifdef OCTO_SMALL_FULL_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
        // The published stores of this protocol — ALL of them must stay agent-scope atomic (write-through) stores for the shortcut below:
        //   (1) a.partials[(w·n_tasks + task)·NACC + k]   this block's task sums           (__hip_atomic_store above, after the block reduction)
        //   (2) a.extra[w], a.extra[(1 + d)·ldw + w]       the HGCA term and its partials   (__hip_atomic_store in the HGCA block)
        //   (3) counters[w]                                 the finished-block count        (__hip_atomic_fetch_add below)
        // A PLAIN store added to this set would need the full agent-scope release (OCTO_SMALL_FULL_FENCE) again.
        // Everything this protocol publishes is an agent-scope ATOMIC store (sc1: written through, never left dirty in L2), so the
        // L2 write-back half of the fence has nothing to do — and costs 1-6 µs per call on the multi-block latency path (same-box
        // A/B, profiles/r3_small_fence_ab.txt). What the release needs from the hardware is the wait for those stores' acknowledgements.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (multi) {
            if (threadIdx.x == 0) {
                const int old = __hip_atomic_fetch_add(&counters[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_flag = (old == n_blocks - 1) ? 1 : 0;
            }
            __syncthreads();
            if (!last_flag) return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- finish (the single / last block). All 256 threads gather the task partials — thread (slot, k) sums tasks slot,
    // slot + NTS, … of running sum k, plain loads (the acquire fence above made them coherent), several in flight — then wave 0
    // combines the slots in order: lane k owns running sum k. The order is fixed, whichever block got here.
    TRACE_POINT();      // last block known
    constexpr int NTS = SMALL_TPB / WAVE;      // independent of NACC: the forward-only and the gradient launch sum in the same order
    __shared__ double psum[NTS * NACC];
    const int slot = lane < NACC ? wv : NTS, kcol = lane < NACC ? lane : 0;
    if constexpr (NUIS && !MODEL) {      // every nuisance finite (k_setup's check)
        bool fin = true;
        if (inl.n > 0) fin = inl.pad == 0;      // the host looked at the handful of values it put into the arguments
        else for (int k = lane; k < a.n_obs * OCTO_N_NUIS; k += WAVE) fin = fin && isfinite(a.nuis[(int64_t)k * a.ld + wi]);
        ok = ok && __all(fin);
    }
    double gp_acc = 0.0, ll = 0.0;
    double gth = 0.0;                          // MODEL: Σ_k J[k][lane]·ḡ[k], this lane's share of ∇θ_t
    double oneil_g[oneil_slots<P, GRAD, NUIS, KM>()];
#pragma unroll
    for (int k = 0; k < oneil_slots<P, GRAD, NUIS, KM>(); ++k) oneil_g[k] = 0.0;
    double sma_p[P], e_p[P], M_p[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { sma_p[p] = park_fin[p][OCTO_N_EL]; e_p[p] = park_fin[p][OCTO_EL_E]; M_p[p] = park_fin[p][OCTO_EL_M]; }
    for (int o = 0; o < a.n_obs; ++o) {
        const int t0 = a.obs_range[2 * o], t1 = a.obs_range[2 * o + 1];
        if (slot < NTS) {
            double s = 0.0;
            if (via_mem) {
                const double* __restrict__ pp = a.partials + (int64_t)w * n_tasks * NACC + kcol;      // [walker][task][NACC]
#pragma unroll 8
                for (int t = t0 + slot; t < t1; t += NTS) s += pp[(int64_t)t * NACC];
            } else if (slot == 0 && t1 > t0) {
                s = tot[kcol];
            }
            psum[slot * NACC + kcol] = s;
        }
        __syncthreads();
        if (wv == 0) {
            double sum = 0.0;
            if (lane < NACC) {
#pragma unroll
                for (int q = 0; q < NTS; ++q) sum += psum[q * NACC + lane];
            }
            gp_acc += sum;
            double v[NOBS_ACC];
#pragma unroll
            for (int k = 0; k < NOBS_ACC; ++k) v[k] = 0.0;
            v[0] = lane_value(sum, L::OFF_S);
            if constexpr (L::HAS_MARG) { v[1] = lane_value(sum, L::OFF_MARG); v[2] = lane_value(sum, L::OFF_MARG + 1); v[3] = lane_value(sum, L::OFF_MARG + 2); }
            if constexpr (L::N_NU > 0) { v[4] = lane_value(sum, L::OFF_NU); v[5] = lane_value(sum, L::OFF_NU + 1); v[6] = lane_value(sum, L::OFF_NU + 2); }
            if constexpr (L::HAS_ONEIL) {
                v[7] = lane_value(sum, L::OFF_ONEIL);
                if constexpr (GRAD) { v[8] = lane_value(sum, L::OFF_ONEIL + 1); v[9] = lane_value(sum, L::OFF_ONEIL + 2); v[10] = lane_value(sum, L::OFF_ONEIL + 3); }
            }
            if constexpr (MODEL) {
                // the observation's nuisances once more, now with their partials (and their UnitLengthPrior terms, counted here
                // exactly once per walker), and ḡ_nuis kept in registers for this lane's chain-rule sum
                double nu[OCTO_N_NUIS];
                D1 nuD[OCTO_N_NUIS];
                const int kind = a.obs[o].kind;
                nuis_of(o, kind, nu, nuD, true);
                bool fin = true;
#pragma unroll
                for (int r = 0; r < OCTO_N_NUIS; ++r) fin = fin && isfinite(nu[r]);
                ok = ok && fin;
                double gn3[OCTO_N_NUIS] = {0.0, 0.0, 0.0};
                ll += obs_finish<P, GRAD, NUIS, KM>(a.obs, 1, gn3, a.extra ? a.extra + w : nullptr, a.ldw, a.c.k_yr, o, v, a.obs_const[o], sma_p, e_p, M_p, true, oneil_g);
                if constexpr (L::N_NU > 0) {
#pragma unroll
                    for (int r = 0; r < OCTO_N_NUIS; ++r) gth = fma(nuD[r].d[0], gn3[r], gth);
                }
            } else {
                ll += obs_finish<P, GRAD, NUIS, KM>(a.obs, a.ld, L::N_NU > 0 ? a.g_nuis + (int64_t)o * OCTO_N_NUIS * a.ld + wo : nullptr, a.extra ? a.extra + w : nullptr,
                                                    a.ldw, a.c.k_yr, o, v, a.obs_const[o], sma_p, e_p, M_p, lane == 0, oneil_g);
            }
        }
        __syncthreads();
    }
    TRACE_POINT();      // observations finished
    // ---- element adjoints. One planet: wave 0 alone. Several planets (WAVE_FIN): wave p maps planet p's running sums to its nine element
    // adjoints — planet_finish is ~250 dependent instructions, and P of them one after the other in wave 0 were most of the finish; the
    // sums, the O'Neil corrections and the validity verdict reach the waves through LDS (one barrier), and for the fused model launch
    // each wave's share of Σ_k J[k][lane]·ḡ[k] goes back to wave 0 the same way (summed in planet order: deterministic).
    constexpr bool WAVE_FIN = GRAD && WAVE_SETUP;
    constexpr int FX_OK = NACC, FX_ONEIL = NACC + 1, FX_N = NACC + 1 + oneil_slots<P, GRAD, NUIS, KM>();
    __shared__ double fin_x[WAVE_FIN ? FX_N : 1];
    __shared__ double fin_gth[(WAVE_FIN && MODEL) ? P * WAVE : 1];
    if (wv == 0) {
        if (multi && lane == 0) __hip_atomic_store(&counters[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        if (a.extra) ll += a.extra[w];      // the HGCA term (this walker's extra blocks)
        ok = ok && isfinite(ll);
        if constexpr (WAVE_FIN) {
            if (lane < NACC) fin_x[lane] = gp_acc;
            if (lane == 0) {
                fin_x[FX_OK] = ok ? 1.0 : 0.0;
                if constexpr (L::HAS_ONEIL) {
#pragma unroll
                    for (int k = 0; k < P * 6; ++k) fin_x[FX_ONEIL + k] = oneil_g[k];
                }
            }
        }
    }
    if constexpr (WAVE_FIN) {
        __syncthreads();
        if (wv < P) {
            const int pm = wv;      // this wave's planet (wave-uniform, run time)
            const bool okf = fin_x[FX_OK] != 0.0;
            double gpl[L::PL_N], ogl[6], elv[OCTO_N_EL];
#pragma unroll
            for (int k = 0; k < L::PL_N; ++k) gpl[k] = fin_x[L::OFF_PL + pm * L::PL_N + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) ogl[k] = L::HAS_ONEIL ? fin_x[FX_ONEIL + pm * 6 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = park_fin[pm][k];
            int okind = a.orbit_kind[0], hmass = a.has_mass[0];
#pragma unroll
            for (int p = 1; p < P; ++p) { okind = (pm == p) ? a.orbit_kind[p] : okind; hmass = (pm == p) ? a.has_mass[p] : hmass; }
            if constexpr (MODEL) {
                double gel[OCTO_N_EL];
                planet_finish<P, GRAD, NUIS, KM, true>(elv, gel, 1, a.extra ? a.extra + w : nullptr, a.ldw, a.c, okind, hmass, pm, gpl, L::HAS_ONEIL ? ogl : nullptr, parked_fp(pm), okf);
                double g = 0.0;
#pragma unroll
                for (int k = 0; k < OCTO_N_EL; ++k) g = fma(park_eld[(pm * OCTO_N_EL + k) * WAVE + lane], gel[k], g);
                fin_gth[pm * WAVE + lane] = g;
            } else {
                if (lane == 0)
                    planet_finish<P, GRAD, NUIS, KM, true>(elv, a.g_elems + (int64_t)pm * OCTO_N_EL * a.ld + wo, a.ld, a.extra ? a.extra + w : nullptr, a.ldw, a.c, okind, hmass,
                                                           pm, gpl, L::HAS_ONEIL ? ogl : nullptr, parked_fp(pm), okf);
                // the adjoints of planets 1.. are stored by waves 1..: acknowledged before the barrier that precedes wave 0's completion flag
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
    }
    if (wv != 0) return;
    if constexpr (MODEL) {
        // ℓπcallback: non-finite θ_t -> -Inf; non-finite prior -> returned without the likelihood (logdensitymodel.jl:120-133);
        // the UnitLengthPrior terms are likelihood terms of the reference (variables.jl:309-323): part of ll, never healed
        const double lpp = mx_s[0], glp = mx_glp[lane];      // wave 1's and wave 2's sums of THIS block (the prologue's barrier published them)
        ul.v = mx_s[1]; ul.d[0] = mx_uld[lane];
        const double llk = ok ? ll + ul.v : -INFINITY;
        const double lpr = finite_in ? lpp : -INFINITY;
        double lp = isfinite(lpr) ? lpr + llk : lpr;
        if (isnan(lp)) lp = -INFINITY;
        const bool fin = isfinite(lp);
        if constexpr (GRAD) {
            if constexpr (WAVE_FIN) {
#pragma unroll
                for (int p = 0; p < P; ++p) gth += fin_gth[p * WAVE + lane];
            } else {
                double gp[P * L::PL_N > 0 ? P * L::PL_N : 1];
#pragma unroll
                for (int k = 0; k < P * L::PL_N; ++k) gp[k] = lane_value(gp_acc, L::OFF_PL + k);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    double gel[OCTO_N_EL], elv[OCTO_N_EL];
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = park_fin[p][k];
                    planet_finish<P, GRAD, NUIS, KM, true>(elv, gel, 1, a.extra ? a.extra + w : nullptr, a.ldw, a.c, a.orbit_kind[p], a.has_mass[p], p, &gp[p * L::PL_N],
                                                           L::HAS_ONEIL ? &oneil_g[p * 6] : nullptr, parked_fp(p), ok);
#pragma unroll
                    for (int k = 0; k < OCTO_N_EL; ++k) gth = fma(park_eld[(p * OCTO_N_EL + k) * WAVE + lane], gel[k], gth);
                }
            }
            if (lane < sm.D) sm.grad_out[(int64_t)lane * sm.ld_o + w * sm.ws_o] = fin ? glp + ul.d[0] + gth : 0.0;
        }
        if (lane == 0) sm.lp_out[w * sm.ws_o] = lp;
    } else {
        if (lane == 0) {
            a.ll_out[wo] = ok ? ll : -INFINITY;
            if constexpr (GRAD) {
                if (!ok && L::N_NU > 0) {
                    for (int k = 0; k < a.n_obs * OCTO_N_NUIS; ++k) a.g_nuis[(int64_t)k * a.ld + wo] = 0.0;
                }
                if constexpr (!WAVE_FIN) {
                    double gp[P * L::PL_N > 0 ? P * L::PL_N : 1];
#pragma unroll
                    for (int k = 0; k < P * L::PL_N; ++k) gp[k] = lane_value(gp_acc, L::OFF_PL + k);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        double elv[OCTO_N_EL];
#pragma unroll
                        for (int k = 0; k < OCTO_N_EL; ++k) elv[k] = park_fin[p][k];
                        planet_finish<P, GRAD, NUIS, KM, true>(elv, a.g_elems + (int64_t)p * OCTO_N_EL * a.ld + wo, a.ld, a.extra ? a.extra + w : nullptr, a.ldw, a.c, a.orbit_kind[p], a.has_mass[p],
                                                               p, &gp[p * L::PL_N], L::HAS_ONEIL ? &oneil_g[p * 6] : nullptr, parked_fp(p), ok);
                    }
                }
            }
        }
    }
    TRACE_POINT();      // outputs stored
    // host-buffer calls: this walker's results are in (mapped, coherent) host memory — release them to the host, which spins
    // on the flag instead of paying a stream synchronisation. (All the wave's stores precede lane 0's release in program order.)
    if (done_flags && lane == 0) __hip_atomic_store(done_flags + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef OCTO_SMALL_TRACE
    TRACE_POINT();
    if (done_flags && w == 0 && lane == 0) for (int k = 0; k < ntr; ++k) done_flags[SMALL_W + k] = tr[k];      // h_flags has room behind the SMALL_W flags
#endif
}

}  // namespace octo
