// The planet-per-wave kernels (octo_mainp.h): their only instantiations, the launchers the per-planet-count translation units call for
// four planets, and the whole dispatch for datasets of more planets than the templated kernels are compiled for (see octo_host.h).
#include "octo_host.h"
#include "octo_mainp.h"

namespace octo {

// The grain: a block is P waves that ALL walk the task's rows, so a task is what ONE wave of k_main walks — two rounds of the resident
// capacity while that leaves a block >= 64 rows, one otherwise (get_tasks keeps >= 32 rows per task).
int64_t plan_key_mainp(const octo_ctx* ctx, int64_t W, int64_t n_rows, int blocks_per_cu) {
    if (ctx->env_chunk > 0) return -ctx->env_chunk;
    const int64_t cols = (W + WAVE - 1) / WAVE;
    const int64_t capacity = std::max<int64_t>((int64_t)blocks_per_cu * ctx->n_cus, 64);
    int64_t rounds = 2;
    while (rounds > 1 && n_rows * cols < 64 * rounds * capacity) --rounds;
    if (ctx->env_rounds > 0) rounds = ctx->env_rounds;
    return std::max<int64_t>(1, rounds * capacity / cols);
}

namespace {
constexpr int KA = KM_RADEC | KM_SEPPA | KM_COR, KB = KM_ALL & ~KM_MARG & ~KM_ONEIL, KC = KM_ALL;      // KC (round 6): + marginalised RV and the O'Neil prior, more than four planets only

// Tiles per block (octo_mainp.h), from the probes of tools/r5_tpb_ab.sh (profiles/r5_tpb_ab.txt): two for 5 and 6 planets — ten waves (3, 3, 2, 2 per SIMD)
// and twelve (3 each): 1.70 -> 1.15 ms and 1.80 -> 1.16 ms per step of the probe; one for 4 and 8 planets (whole multiples of four waves already) and for
// 7 (fourteen waves at four per SIMD: 2.40 -> 2.80 ms). Three planets (experiment builds with -DOCTO_MAINP_MINP=3 only): four tiles, still behind k_main<3>.
// OCTO_MAINP_TPB: experiments.
int mainp_tpb(const octo_ctx* ctx, int P) {
    int tpb = (P == 3) ? 4 : ((P == 5 || P == 6) ? 2 : 1);
    if (ctx->env_mainp_tpb > 0) tpb = (int)ctx->env_mainp_tpb;
    const int max_waves = 4 * mp_wpe(P);
    while (tpb > 1 && (tpb > 4 || tpb * P > max_waves)) --tpb;
    return tpb;
}

template <bool GRAD, bool NUIS, int KM, int MP_R, int WPE>
int launch_mainp_shape(octo_ctx* ctx, int64_t cols, const EvalArgs& a, hipStream_t st) {
    const int P = a.n_planets, tpb = mainp_tpb(ctx, P);
    const size_t lds = mainp_lds_bytes<GRAD, NUIS, KM, MP_R>(P, tpb);
    // (the raised limit is remembered per CONTEXT — per device and owner thread — not in a function-local static: the attribute belongs to the current
    // device's copy of the function, and contexts are not synchronised with each other)
    { int rcl = raise_dynamic_lds(ctx, (const void*)k_mainp<GRAD, NUIS, KM, MP_R, WPE>, lds, "k_mainp"); if (rcl) return rcl; }
    EvalArgs b = a;
    b.n_rblocks = tpb;
    hipLaunchKernelGGL((k_mainp<GRAD, NUIS, KM, MP_R, WPE>), dim3((unsigned)((cols + tpb - 1) / tpb), (unsigned)a.n_tasks), dim3((unsigned)(WAVE * P * tpb)), lds, st, b);
    return OCTO_OK;
}
template <bool GRAD, bool NUIS, int KM>
int launch_mainp_t(octo_ctx* ctx, int64_t cols, const EvalArgs& a, hipStream_t st) {
    if (a.n_planets > 6) return launch_mainp_shape<GRAD, NUIS, KM, mp_rows(8), mp_wpe(8)>(ctx, cols, a, st);
    return launch_mainp_shape<GRAD, NUIS, KM, mp_rows(4), mp_wpe(4)>(ctx, cols, a, st);
}
template <bool GRAD, bool NUIS, int KM>
int launch_finishp_t(octo_ctx* ctx, int64_t cols, const EvalArgs& a, hipStream_t st) {
    const int waves = GRAD ? 1 + a.n_planets : 1;
    const size_t rows = (size_t)(1 + a.n_planets) + ((GRAD && (KM & KM_ONEIL)) ? 6 * (size_t)a.n_planets : 0);      // validity flags [+ the O'Neil terms of each planet's adjoints]
    hipLaunchKernelGGL((k_finishp<GRAD, NUIS, KM>), dim3((unsigned)cols), dim3((unsigned)(WAVE * waves)), sizeof(double) * WAVE * rows, st, a);
    (void)ctx;
    return OCTO_OK;
}
template <bool NUIS, int KM>
int occupancy_t(octo_ctx* ctx, int P) {
    // in TILES per CU. (The API counts registers and LDS; what the hardware really places is bounded by its fixed wave -> SIMD order as well: octo_mainp.h)
    const int tpb = mainp_tpb(ctx, P);
    int nb = 0;
    // the two-tile blocks of 5 / 6 planets ask for more than the default 48 KB: opt in BEFORE the query
    if (P > 6) (void)raise_dynamic_lds(ctx, (const void*)k_mainp<true, NUIS, KM, mp_rows(8), mp_wpe(8)>, mainp_lds_bytes<true, NUIS, KM, mp_rows(8)>(P, tpb), "k_mainp");
    else (void)raise_dynamic_lds(ctx, (const void*)k_mainp<true, NUIS, KM, mp_rows(4), mp_wpe(4)>, mainp_lds_bytes<true, NUIS, KM, mp_rows(4)>(P, tpb), "k_mainp");
    hipError_t e = P > 6 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mainp<true, NUIS, KM, mp_rows(8), mp_wpe(8)>, WAVE * P * tpb, (mainp_lds_bytes<true, NUIS, KM, mp_rows(8)>(P, tpb)))
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mainp<true, NUIS, KM, mp_rows(4), mp_wpe(4)>, WAVE * P * tpb, (mainp_lds_bytes<true, NUIS, KM, mp_rows(4)>(P, tpb)));
    if (e != hipSuccess || nb < 1) nb = 1;
    const int waves = P * tpb, per_simd = (waves + 3) / 4;               // the most loaded SIMD's share of one block
    nb = std::max(1, std::min(nb, mp_wpe(P) / per_simd));
    return nb * tpb;
}
}  // namespace

#define OCTO_PN_DISPATCH_K(FN, K, ...)                                                                              \
    (grad ? (nuis ? FN<true, true, K>(__VA_ARGS__) : FN<true, false, K>(__VA_ARGS__))                                \
          : (nuis ? FN<false, true, K>(__VA_ARGS__) : FN<false, false, K>(__VA_ARGS__)))
#define OCTO_PN_DISPATCH(FN, ...)                                                                                   \
    (km_p == KA ? OCTO_PN_DISPATCH_K(FN, KA, __VA_ARGS__) : (km_p == KB ? OCTO_PN_DISPATCH_K(FN, KB, __VA_ARGS__) : OCTO_PN_DISPATCH_K(FN, KC, __VA_ARGS__)))

int mainp_occupancy(octo_ctx* ctx, bool nuis, int km_p, int P) {
    if (km_p == KA) return nuis ? occupancy_t<true, KA>(ctx, P) : occupancy_t<false, KA>(ctx, P);
    if (km_p == KB) return nuis ? occupancy_t<true, KB>(ctx, P) : occupancy_t<false, KB>(ctx, P);
    return nuis ? occupancy_t<true, KC>(ctx, P) : occupancy_t<false, KC>(ctx, P);
}
// k_marg on the forward partials of the planet-per-wave kernels (their layout is Layout<2, false, ·, ·> for every P: no planet sums without a gradient)
int launch_margp(octo_ctx* ctx, bool nuis, int km_p, const EvalArgs& a, hipStream_t st) {
    if (km_p != KC) return fail(ctx, OCTO_EINVAL, "internal: k_marg requested for a kind set without marginalised RV");
    const dim3 g((unsigned)((a.W + 255) / 256));
    if (nuis) hipLaunchKernelGGL((k_marg<2, true, KC>), g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_marg<2, false, KC>), g, dim3(256), 0, st, a);
    return OCTO_OK;
}
int launch_mainp(octo_ctx* ctx, bool grad, bool nuis, int km_p, int64_t cols, const EvalArgs& a, hipStream_t st) {
    return OCTO_PN_DISPATCH(launch_mainp_t, ctx, cols, a, st);
}
int launch_finishp(octo_ctx* ctx, bool grad, bool nuis, int km_p, int64_t cols, const EvalArgs& a, hipStream_t st) {
    return OCTO_PN_DISPATCH(launch_finishp_t, ctx, cols, a, st);
}

// A dataset of MAXP_T < P <= MAXP planets: always the throughput kernels (no k_small<P> there), k_mainp -> k_finishp on one stream.
int dispatch_many(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, bool grad, bool nuis, const SmallModel* sm, hipStream_t st) {
    if (sm) return fail(ctx, OCTO_EINVAL, "internal: fused model launch requested for a dataset of more than four planets");
    const int P = ds->n_planets;
    const int km_p = mainp_kind_set(ds->kind_mask);
    const int64_t cols = (a.W + WAVE - 1) / WAVE;
    int& blocks_per_cu = ctx->occupancy[(uint32_t)((P << 16) | ((nuis ? 1 : 0) << 15) | km_p)];
    if (blocks_per_cu == 0) blocks_per_cu = mainp_occupancy(ctx, nuis, km_p, P);
    TaskTable* tt = nullptr;
    int rc = get_tasks(ctx, ds, plan_key_mainp(ctx, a.W, ds->n_rows, blocks_per_cu), &tt, nuis, 1);
    if (rc) return rc;
    a.tasks = tt->d_tasks; a.task_const = nuis ? tt->d_const_raw : tt->d_const_pre;
    a.obs_range = tt->d_obs_range; a.obs_const = nuis ? tt->d_obs_const_raw : tt->d_obs_const_pre;
    a.n_tasks = tt->n_tasks;
    // partials: OFF_PL + P·PL_N rows per task — bounded by the widest layout (11 observation sums + 12 per planet)
    const int64_t nacc_max = NOBS_ACC + (int64_t)P * 12;
    rc = grow(ctx, ctx->d_partials, ctx->cap_part, (int64_t)std::max(a.n_tasks, 1) * nacc_max * a.ldw);
    if (rc) return rc;
    a.partials = ctx->d_partials;
    a.extra = nullptr; a.marg = nullptr; a.marg_out = nullptr;
    if (ds->n_hgca > 0) {      // the proper-motion anomaly (no epoch loop): k_hgcap ahead of k_finishp, which adds it (launch_all's hgca_term for these systems)
        if (!nuis) return fail(ctx, OCTO_EINVAL, "octo_eval: a dataset with an OCTO_HGCA table needs `nuis` (pmra, pmdec)");
        const int n_dir = grad ? P * OCTO_N_EL + a.n_obs * OCTO_N_NUIS : 1;
        rc = grow(ctx, ctx->d_extra, ctx->cap_extra, (int64_t)(1 + n_dir) * a.ldw);
        if (rc) return rc;
        a.extra = ctx->d_extra;
        hipLaunchKernelGGL(k_hgcap, dim3((unsigned)cols, (unsigned)n_dir), dim3(WAVE), 0, st, a);
    }
    if (a.n_tasks > 0) {
        hipEvent_t e1 = nullptr;
        if (ctx->timing_every > 0 && (ctx->timing_seq++ % ctx->timing_every) == 0) {      // HIP events around the epoch-loop kernel, as in launch_all
            if (ctx->ev_used == ctx->ev_pool.size()) {
                hipEvent_t x, y;
                HIPCHK(ctx, hipEventCreate(&x)); HIPCHK(ctx, hipEventCreate(&y));
                ctx->ev_pool.emplace_back(x, y);
            }
            hipEvent_t e0 = ctx->ev_pool[ctx->ev_used].first; e1 = ctx->ev_pool[ctx->ev_used].second; ctx->ev_used++;
            HIPCHK(ctx, hipEventRecord(e0, st));
        }
        if (grad && (ds->kind_mask & KM_MARG)) {
            // marginalised RV with a gradient (rv-absolute-margin.jl:140-185): r̄v = 2 (r − μ̂)/var needs μ̂ = −B/2A of the WHOLE table first — a forward
            // pre-pass over that table's tasks, k_marg, then the gradient pass (k_main's flow for these tables; the forward partials fit in the gradient buffer)
            rc = grow(ctx, ctx->d_marg, ctx->cap_marg, (int64_t)a.n_obs * 2 * a.ldw);
            if (rc) return rc;
            a.marg = nullptr; a.marg_out = ctx->d_marg;
            const Task* tks = tt->h_tasks.data();
            for (int t0 = 0; t0 < tt->n_tasks;) {
                const int o = tks[t0].obs;
                int t1 = t0;
                while (t1 < tt->n_tasks && tks[t1].obs == o) ++t1;
                if (ds->h_obs[o].kind == OCTO_RV_ABS_MARG) {
                    EvalArgs b = a;
                    b.task0 = t0; b.n_tasks = t1 - t0;
                    rc = launch_mainp(ctx, false, nuis, km_p, cols, b, st);
                    if (rc) return rc;
                }
                t0 = t1;
            }
            rc = launch_margp(ctx, nuis, km_p, a, st);
            if (rc) return rc;
            a.marg = ctx->d_marg;
        }
        rc = launch_mainp(ctx, grad, nuis, km_p, cols, a, st);
        if (rc) return rc;
        if (e1) HIPCHK(ctx, hipEventRecord(e1, st));
    }
    rc = launch_finishp(ctx, grad, nuis, km_p, cols, a, st);
    if (rc) return rc;
    HIPCHK(ctx, hipGetLastError());
    ctx->mt_applied = a.mt_lpp != nullptr;
    return OCTO_OK;
}

}  // namespace octo
