// octo_launch.h — the launch templates of one evaluation: planner, k_setup / k_main / k_finish (+ k_marg, k_hgca) for big batches,
// k_small for small ones, and the choice of the compiled kind set. Included only by octo_inst_p{1..4}.hip.
#pragma once
#include "octo_host.h"
#ifndef OCTO_MAINP
#define OCTO_MAINP 1      // 0: experiments — four planets on k_main<4> as before round 5
#endif
#ifndef OCTO_MAINP_MINP
#define OCTO_MAINP_MINP 4 // fewest planets that take the planet-per-wave kernel (3: experiments)
#endif

namespace octo {

template <int P, bool GRAD, bool NUIS, int KM, bool MODEL>
int launch_small(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, const SmallModel* smp, hipStream_t st) {
    using L = Layout<P, GRAD, NUIS, KM>;
    static_assert(L::NACC <= WAVE, "k_small: lane k of the finishing wave owns running sum k");
    // rows per block: enough blocks to spread one walker's epochs over the chip (about two blocks per CU in total), never
    // less than one row per lane; the same partition for forward and gradient launches (bit-identical values).
    // With the inputs in host memory every block starts with a PCIe read of its walker's elements (one or two 64-byte
    // requests); a few hundred of those in flight is what the link sustains without queueing (measured: 448 blocks x 9
    // scattered reads made W = 32 twice as slow as W = 1), so the block count is capped there.
    int64_t total_blocks = ctx->stage_ws_in > 0 ? 160 : 2 * (int64_t)ctx->n_cus;
    if (ctx->env_small_blocks > 0) total_blocks = ctx->env_small_blocks;      // OCTO_SMALL_BLOCKS: experiments
    const int64_t target_tasks = std::max<int64_t>(1, total_blocks / a.W);
    int64_t span = (ds->n_rows + target_tasks - 1) / target_tasks;
    int64_t min_span = a.W < 64 ? SMALL_TPB : 4 * SMALL_TPB;      // many walkers fill the chip by themselves: split a walker's rows only when each part is worth a block
    if (ctx->env_small_min_span > 0) min_span = ctx->env_small_min_span;      // OCTO_SMALL_MIN_SPAN: experiments
    span = std::max<int64_t>(min_span, (span + SMALL_TPB - 1) / SMALL_TPB * SMALL_TPB);
    TaskTable* tt = nullptr;
    int rc = get_tasks(ctx, ds, -(span / WPB) - SMALL_KEY, &tt);
    if (rc) return rc;
    a.tasks = tt->d_tasks; a.task_const = NUIS ? tt->d_const_raw : tt->d_const_pre;
    a.obs_range = tt->d_obs_range; a.obs_const = NUIS ? tt->d_obs_const_raw : tt->d_obs_const_pre;
    a.n_tasks = tt->n_tasks;
    rc = grow(ctx, ctx->d_partials, ctx->cap_part, (int64_t)std::max(a.n_tasks, 1) * L::NACC * a.ldw);
    if (rc) return rc;
    a.partials = ctx->d_partials;
    a.marg = nullptr; a.marg_out = nullptr; a.extra = nullptr;
    // row blocks per walker: every task its own block while the walkers are few; with many walkers the blocks fill the chip by
    // themselves and one block walks several tasks (tables), which saves their setup and, at one block per walker, the counter protocol
    a.n_rblocks = (int32_t)std::min<int64_t>(std::max(a.n_tasks, 1), target_tasks);
    a.n_hblocks = 0;
    if (ds->n_hgca > 0) {
        // The proper-motion anomaly has no epoch loop. A handful of walkers: extra blocks of the same launch, one input direction
        // per wave, lanes over the table's rows (26 µs per call at W = 1 instead of 30 with a launch of its own). More walkers:
        // k_hgca (lane = walker) ahead of k_small — 4-5 more blocks per walker would each fetch the walker's inputs again.
        if constexpr (NUIS && (KM & KM_HGCA) != 0) {
            const int n_dir = P * OCTO_N_EL + a.n_obs * OCTO_N_NUIS;
            rc = grow(ctx, ctx->d_extra, ctx->cap_extra, (int64_t)(1 + n_dir) * a.ldw);
            if (rc) return rc;
            a.extra = ctx->d_extra;
            if (hgca_in_small(a.W)) a.n_hblocks = GRAD ? (n_dir + SMALL_TPB / WAVE - 1) / (SMALL_TPB / WAVE) : 1;
            else if constexpr (!MODEL) hipLaunchKernelGGL((k_hgca<P>), dim3((unsigned)((a.W + WAVE - 1) / WAVE), (unsigned)n_dir), dim3(WAVE), 0, st, a);
            else return fail(ctx, OCTO_EINVAL, "internal: fused model launch requested for an HGCA dataset beyond hgca_in_small");
        } else {
            return fail(ctx, OCTO_EINVAL, NUIS ? "internal: HGCA dataset dispatched to a kind set without KM_HGCA"
                                               : "octo_eval: a dataset with an OCTO_HGCA table needs `nuis` (pmra, pmdec)");
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = ctx->timing_every > 0 && (ctx->timing_seq++ % ctx->timing_every) == 0;
    if (timed) {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            hipEvent_t x, y;
            HIPCHK(ctx, hipEventCreate(&x)); HIPCHK(ctx, hipEventCreate(&y));
            ctx->ev_pool.emplace_back(x, y);
        }
        e0 = ctx->ev_pool[ctx->ev_used].first; e1 = ctx->ev_pool[ctx->ev_used].second; ctx->ev_used++;
        HIPCHK(ctx, hipEventRecord(e0, st));
    }
    uint64_t* flags = nullptr;
    if (ctx->flag_request) {      // a host-buffer call is waiting for these results: let it spin on per-walker flags
        flags = ctx->h_flags; ctx->flag_seq += 1; ctx->flag_armed = true;
    }
    SmallModel sm;
    std::memset(&sm, 0, sizeof(sm));
    if (MODEL) sm = *smp;
    hipLaunchKernelGGL((k_small<P, GRAD, NUIS, KM, MODEL>), dim3((unsigned)(a.n_rblocks + a.n_hblocks), (unsigned)a.W), dim3(SMALL_TPB), 0, st, a, sm,
                       ctx->d_counters, flags, ctx->flag_seq, ctx->inl);
    if (timed) HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipGetLastError());
    return OCTO_OK;
}

// The kind set k_small is instantiated for (a subset of the epoch-loop kernels' sets: rows are per LANE there and the branch on the
// table's kind is block-uniform, so a wider set costs a one-θ call nothing it can measure — profiles/r4_small_kindsets_ab.txt — while
// every set costs ~1 MB of code over the four planet counts): RA/Dec alone; + sep/PA and cor; + absolute / relative RV; everything
// (O'Neil priors and marginalised RV).
constexpr int small_kind_set(int km_rows, int n_planets = 1) {
    // (three and four planets: RA/Dec alone rides the sep/PA + cor set — 0.8 MB of code for a branch a one-θ call cannot measure; round 6)
    if (km_rows == KM_RADEC) return n_planets >= 3 ? (KM_RADEC | KM_SEPPA | KM_COR) : KM_RADEC;
    if ((km_rows & ~(KM_RADEC | KM_SEPPA | KM_COR)) == 0) return KM_RADEC | KM_SEPPA | KM_COR;
    if ((km_rows & (KM_MARG | KM_ONEIL)) == 0) return KM_ALL & ~KM_MARG & ~KM_ONEIL;
    return KM_ALL;
}

template <int P, bool GRAD, bool NUIS, int KMD>
int launch_all(octo_ctx* ctx, const octo_dataset* cds, EvalArgs& a, const SmallModel* sm, hipStream_t st) {
    // KMD: the dataset's kind set, possibly with KM_HGCA. The epoch-loop kernels never see that bit (KM below); k_small does, when it is
    // compiled with nuisances (an HGCA table without `nuis` is refused at run time, so the nuisance-free variants need no HGCA twin).
    constexpr int KM = KMD & ~KM_HGCA;
    constexpr int KSR = small_kind_set(KM, P);
    constexpr int KS = !(NUIS && (KMD & KM_HGCA)) ? KSR      // with an HGCA table: relative astrometry alone, or everything
                       : ((KSR & ~(KM_RADEC | KM_SEPPA | KM_COR)) == 0 ? (KM_RADEC | KM_SEPPA | KM_COR | KM_HGCA) : (KM_ALL | KM_HGCA));
    using L = Layout<P, GRAD, NUIS, KM>;
    const int64_t cols = (a.W + WAVE - 1) / WAVE;
    const octo_dataset* ds = cds;
    {
        if (sm) return launch_small<P, GRAD, NUIS, KS, true>(ctx, ds, a, sm, st);      // the caller checked small_eligible
        if (small_eligible(ctx, ds, a.W)) return launch_small<P, GRAD, NUIS, KS, false>(ctx, ds, a, nullptr, st);
    }
    if (sm) return fail(ctx, OCTO_EINVAL, "internal: fused model launch requested for an ineligible dataset");
    // Two launches: the orbit constructors in k_main's prologue (octo_kernels.h: k_main<FUSED>), then k_finish. A marginalised-RV GRADIENT
    // keeps the round-2 shape (k_setup -> forward pre-pass -> k_marg -> k_main -> k_finish: the pre-pass and k_marg read `wc`); that path,
    // its non-fused k_main and the k_finish that reads `wc` are compiled for the kind sets with marginalised RV only.
    const bool marg_ds = L::HAS_MARG && (ds->kind_mask & KM_MARG);
    // Three and more planets, kind sets without marginalised RV / O'Neil: one planet per wave (octo_mainp.h), the partials in k_main's layout
    // (measured, same box: 4 planets 774 -> 573 µs per step of the probe; 3 planets 345 -> 427: k_main<3> stays)
    constexpr bool MAINP = OCTO_MAINP && P >= OCTO_MAINP_MINP && !(KM & (KM_MARG | KM_ONEIL));
    constexpr int KMP = mainp_kind_set(KM);      // same partial layout as KM: the sets differ in sep/PA and cor only, or in which RV kinds
    // Occupancy of the GRADIENT variant that is launched (FUSED: more registers, and for several planets more LDS), also for forward-only
    // launches: both then use the same row partition, so the forward value and the value returned with a gradient are the same sum in
    // the same order — bit-identical, like the primal of a ForwardDiff dual. Cached per context (= per device) and variant.
    int& blocks_per_cu = ctx->occupancy[(uint32_t)((P << 16) | ((NUIS ? 1 : 0) << 15) | ((marg_ds ? 1 : 0) << 14) | KM)];
    if (blocks_per_cu == 0) {
        int nb = 0;
        hipError_t qe = hipErrorUnknown;
        if constexpr (L::HAS_MARG) {
            if (marg_ds) qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_main<P, true, NUIS, KM, false>, WAVE * WPB, (main_lds_bytes<P, true, NUIS, KM>()));
        }
        if constexpr (MAINP) { nb = mainp_occupancy(ctx, NUIS, KMP, P); qe = hipSuccess; }
        else if (!marg_ds) qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_main<P, true, NUIS, KM, true>, WAVE * WPB, (fused_lds_bytes<P, true, NUIS, KM>()));
        if (qe != hipSuccess || nb < 1) nb = 2;
        blocks_per_cu = nb;
    }
    TaskTable* tt = nullptr;
    // one planet, fused launch: a one-round grid may take eight-wave blocks (octo_kernels.h: k_main<…, NWV>; plan_key decides)
    // (… and three of its blocks — plan_key offers it up to that many per CU — must fit in a CU's 160 KB of LDS next to each other: their combine
    // buffer holds seven waves' sums. The gradient layout decides for the forward-only launch too: both take the same partition.)
    constexpr bool WIDE_OK = P == 1 && fused_lds_bytes<P, true, NUIS, KM, 2 * WPB>() <= 48 * 1024;      // (stays under the default dynamic-LDS limit of a launch)
    // … and the blocks of the one-round grid must be resident at once in their EIGHT-wave form: the occupancy of that instantiation on this
    // device (512 threads, a combine buffer of seven waves' sums), queried once and cached like the four-wave kernel's (rounds 4's
    // rule hard-coded three blocks per CU and 160 KB of LDS)
    int blocks8_per_cu = 0;
    if constexpr (WIDE_OK) {
        int& nb8 = ctx->occupancy[(uint32_t)((P << 16) | ((NUIS ? 1 : 0) << 15) | (1 << 13) | KM)];
        if (nb8 == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_main<P, true, NUIS, KM, true, 2 * WPB>, WAVE * 2 * WPB, (fused_lds_bytes<P, true, NUIS, KM, 2 * WPB>())) != hipSuccess || nb < 1) nb = -1;
            nb8 = nb;
        }
        blocks8_per_cu = nb8 > 0 ? nb8 : 0;
    }
    bool wide = false;
    // OCTO_OPT_BATCH_INVARIANT: ONE row partition whatever the batch size (64 rows per wave, four-wave blocks), so that a walker's sums are formed in
    // the same order in every batch it is part of
    const int64_t pkey = ctx->opt_invariant ? (int64_t)-64
                         : MAINP ? plan_key_mainp(ctx, a.W, ds->n_rows, blocks_per_cu)
                                 : plan_key(ctx, a.W, ds->n_rows, blocks_per_cu, (WIDE_OK && !marg_ds && blocks8_per_cu > 0) ? &wide : nullptr, blocks8_per_cu);
    int rc0 = get_tasks(ctx, ds, pkey, &tt, NUIS, MAINP ? 1 : (wide ? 2 * WPB : WPB));
    if (rc0) return rc0;
    a.tasks = tt->d_tasks; a.task_const = NUIS ? tt->d_const_raw : tt->d_const_pre;
    a.obs_range = tt->d_obs_range; a.obs_const = NUIS ? tt->d_obs_const_raw : tt->d_obs_const_pre;
    a.n_tasks = tt->n_tasks;
    const int64_t need = (int64_t)a.n_tasks * L::NACC * a.ldw;
    int rc = grow(ctx, ctx->d_partials, ctx->cap_part, need);
    if (rc) return rc;
    a.partials = ctx->d_partials;
    a.extra = nullptr; a.marg = nullptr; a.marg_out = nullptr;
    auto hgca_term = [&]() -> int {      // the proper-motion anomaly (no epoch loop): k_hgca ahead of k_finish, which adds it
        if (ds->n_hgca == 0) return OCTO_OK;
        if constexpr (NUIS) {
            const int n_dir = P * OCTO_N_EL + a.n_obs * OCTO_N_NUIS;
            int rcx = grow(ctx, ctx->d_extra, ctx->cap_extra, (int64_t)(1 + n_dir) * a.ldw);
            if (rcx) return rcx;
            a.extra = ctx->d_extra;
            hipLaunchKernelGGL((k_hgca<P>), dim3((unsigned)cols, (unsigned)n_dir), dim3(WAVE), 0, st, a);
            return OCTO_OK;
        } else {
            return fail(ctx, OCTO_EINVAL, "octo_eval: a dataset with an OCTO_HGCA table needs `nuis` (pmra, pmdec)");
        }
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto timing_begin = [&]() -> int {      // HIP events around k_main of every timing_every-th evaluation, on its launch stream
        if (!(ctx->timing_every > 0 && (ctx->timing_seq++ % ctx->timing_every) == 0)) return OCTO_OK;
        if (ctx->ev_used == ctx->ev_pool.size()) {
            hipEvent_t x, y;
            HIPCHK(ctx, hipEventCreate(&x)); HIPCHK(ctx, hipEventCreate(&y));
            ctx->ev_pool.emplace_back(x, y);
        }
        e0 = ctx->ev_pool[ctx->ev_used].first; e1 = ctx->ev_pool[ctx->ev_used].second; ctx->ev_used++;
        HIPCHK(ctx, hipEventRecord(e0, st));
        return OCTO_OK;
    };
    if (!(GRAD && marg_ds)) {
        rc = hgca_term();
        if (rc) return rc;
        if (a.n_tasks > 0) {      // (a dataset without rows — an HGCA table alone, empty tables — is k_finish's closed forms only)
            // walker tiles made homogeneous for the warm-started loop (octo_tile.h): one sort launch ahead of k_main when it pays
            if constexpr (!MAINP && ((P == 1 && main_warm<P, true, NUIS, KM, true>()) || main_warm_last<P, true, NUIS, KM, true>())) {
                rc = tile_prepare(ctx, ds, a, st);
                if (rc) return rc;
            }
            rc = timing_begin();
            if (rc) return rc;
            bool launched = false;
            if constexpr (WIDE_OK) {
                if (wide) {
                    hipLaunchKernelGGL((k_main<P, GRAD, NUIS, KM, true, 2 * WPB>), dim3((unsigned)cols, (unsigned)a.n_tasks), dim3(WAVE * 2 * WPB),
                                       (fused_lds_bytes<P, GRAD, NUIS, KM, 2 * WPB>()), st, a);
                    launched = true;
                }
            }
            if constexpr (MAINP) {
                static_assert(Layout<2, GRAD, NUIS, KMP>::OFF_PL == L::OFF_PL && Layout<2, GRAD, NUIS, KMP>::PL_N == L::PL_N, "k_mainp writes k_finish<P>'s partial layout");
                rc = launch_mainp(ctx, GRAD, NUIS, KMP, cols, a, st);
                if (rc) return rc;
                launched = true;
            }
            if constexpr (!MAINP) {
                if (!launched) {
                    // ONE task (a table of a few dozen rows: a mid-size callback), one table, nothing from k_hgca: the k_main blocks finish their own
                    // tiles (octo_kernels.h: fin_in_main) — no partials, no k_finish launch
                    if constexpr (main_fin_fused<P, GRAD, NUIS, KM, true, WPB>()) {
                        if (a.n_tasks == 1 && ds->n_obs == 1 && ds->n_hgca == 0 && !ctx->env_no_fin_fused) a.fin_fused = 1;
                        if (a.fin_fused)
                            hipLaunchKernelGGL((k_main<P, GRAD, NUIS, KM, true, WPB, true>), dim3((unsigned)cols, 1u), dim3(WAVE * WPB),
                                               (fused_lds_bytes<P, GRAD, NUIS, KM>()), st, a);
                    }
                    if (!a.fin_fused)
                        hipLaunchKernelGGL((k_main<P, GRAD, NUIS, KM, true>), dim3((unsigned)cols, (unsigned)a.n_tasks), dim3(WAVE * WPB),
                                           (fused_lds_bytes<P, GRAD, NUIS, KM>()), st, a);
                }
            }
            if (e1) HIPCHK(ctx, hipEventRecord(e1, st));
            if (a.fin_fused) {
                HIPCHK(ctx, hipGetLastError());
                ctx->mt_applied = a.mt_lpp != nullptr;
                return OCTO_OK;
            }
        }
        hipLaunchKernelGGL((k_finish<P, GRAD, NUIS, KM, false>), dim3((unsigned)cols), dim3(WAVE * fin_waves<P, GRAD, NUIS, KM>()), (fin_lds_bytes<P, GRAD, NUIS, KM>()), st, a);
        HIPCHK(ctx, hipGetLastError());
        ctx->mt_applied = a.mt_lpp != nullptr;
        return OCTO_OK;
    }
    if constexpr (GRAD && L::HAS_MARG) {
        const Task* tt_tasks = tt->h_tasks.data();
        const dim3 gsetup((unsigned)((a.W + 255) / 256));
        hipLaunchKernelGGL(k_setup, dim3(gsetup.x, (unsigned)a.n_planets), dim3(256), 0, st, a);
        if (a.n_tasks > 0) {
            // marginalised RV: forward pre-pass over those tables' tasks for μ̂ and A, then the gradient pass
            using L0 = Layout<P, false, NUIS, KM>;
            static_assert(L0::NACC <= L::NACC, "forward partials fit in the gradient buffer");
            rc = grow(ctx, ctx->d_marg, ctx->cap_marg, (int64_t)a.n_obs * 2 * a.ldw);
            if (rc) return rc;
            a.marg = nullptr; a.marg_out = ctx->d_marg;
            for (int t0 = 0; t0 < a.n_tasks;) {
                const int o = tt_tasks[t0].obs;
                int t1 = t0;
                while (t1 < a.n_tasks && tt_tasks[t1].obs == o) ++t1;
                if (ds->h_obs[o].kind == OCTO_RV_ABS_MARG) {
                    a.task0 = t0;
                    hipLaunchKernelGGL((k_main<P, false, NUIS, KM, false>), dim3((unsigned)cols, (unsigned)(t1 - t0)), dim3(WAVE * WPB),
                                       (main_lds_bytes<P, false, NUIS, KM>()), st, a);
                }
                t0 = t1;
            }
            a.task0 = 0;
            hipLaunchKernelGGL((k_marg<P, NUIS, KM>), gsetup, dim3(256), 0, st, a);
            a.marg = ctx->d_marg;
            rc = timing_begin();
            if (rc) return rc;
            hipLaunchKernelGGL((k_main<P, GRAD, NUIS, KM, false>), dim3((unsigned)cols, (unsigned)a.n_tasks), dim3(WAVE * WPB),
                               (main_lds_bytes<P, GRAD, NUIS, KM>()), st, a);
            if (e1) HIPCHK(ctx, hipEventRecord(e1, st));
        }
        rc = hgca_term();
        if (rc) return rc;
        hipLaunchKernelGGL((k_finish<P, GRAD, NUIS, KM, true>), dim3((unsigned)cols), dim3(WAVE * fin_waves<P, GRAD, NUIS, KM>()), (fin_lds_bytes<P, GRAD, NUIS, KM>()), st, a);
        HIPCHK(ctx, hipGetLastError());
        ctx->mt_applied = a.mt_lpp != nullptr;
    }
    return OCTO_OK;
}

template <int P, int KM>
int dispatch2(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, bool grad, bool nuis, const SmallModel* sm, hipStream_t st) {
    if (grad) return nuis ? launch_all<P, true, true, KM>(ctx, ds, a, sm, st) : launch_all<P, true, false, KM>(ctx, ds, a, sm, st);
    return nuis ? launch_all<P, false, true, KM>(ctx, ds, a, sm, st) : launch_all<P, false, false, KM>(ctx, ds, a, sm, st);
}

template <int P>
int dispatch1(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, bool grad, bool nuis, const SmallModel* sm, hipStream_t st) {
    // The smallest compiled kind set that covers the dataset: registers and occupancy are not paid for code it never runs.
    const int km = ctx->env_kind_all ? KM_ALL : (ds->kind_mask & ~KM_HGCA);      // OCTO_KIND_ALL: experiments (what the narrower kind sets buy)
    if (ds->kind_mask & KM_HGCA) {
        // an HGCA table next to the rows: two kind sets carry k_small's proper-motion-anomaly block — relative astrometry alone (the
        // usual joint fit) and everything; the epoch-loop kernels behind them are the same instantiations as without the table
        if ((km & ~(KM_RADEC | KM_SEPPA | KM_COR)) == 0) return dispatch2<P, KM_RADEC | KM_SEPPA | KM_COR | KM_HGCA>(ctx, ds, a, grad, nuis, sm, st);
        if ((km & ~(KM_RADEC | KM_SEPPA | KM_COR | KM_ONEIL)) == 0) return dispatch2<P, KM_RADEC | KM_SEPPA | KM_COR | KM_ONEIL | KM_HGCA>(ctx, ds, a, grad, nuis, sm, st);
        if ((km & (KM_MARG | KM_ONEIL)) == 0) return dispatch2<P, (KM_ALL & ~KM_MARG & ~KM_ONEIL) | KM_HGCA>(ctx, ds, a, grad, nuis, sm, st);
        return dispatch2<P, KM_ALL | KM_HGCA>(ctx, ds, a, grad, nuis, sm, st);
    }
    if ((km & ~KM_RADEC) == 0) return dispatch2<P, KM_RADEC>(ctx, ds, a, grad, nuis, sm, st);
    if ((km & ~(KM_RADEC | KM_COR)) == 0) return dispatch2<P, KM_RADEC | KM_COR>(ctx, ds, a, grad, nuis, sm, st);
    if ((km & ~(KM_RADEC | KM_RVABS)) == 0) return dispatch2<P, KM_RADEC | KM_RVABS>(ctx, ds, a, grad, nuis, sm, st);      // BASELINE config 4
    if ((km & ~(KM_RADEC | KM_RVREL)) == 0) return dispatch2<P, KM_RADEC | KM_RVREL>(ctx, ds, a, grad, nuis, sm, st);      // imaging + planet RV
    if ((km & ~(KM_RADEC | KM_SEPPA | KM_COR)) == 0) return dispatch2<P, KM_RADEC | KM_SEPPA | KM_COR>(ctx, ds, a, grad, nuis, sm, st);
    if ((km & ~(KM_RADEC | KM_SEPPA | KM_COR | KM_ONEIL)) == 0) return dispatch2<P, KM_RADEC | KM_SEPPA | KM_COR | KM_ONEIL>(ctx, ds, a, grad, nuis, sm, st);
    if ((km & (KM_MARG | KM_ONEIL)) == 0) return dispatch2<P, KM_ALL & ~KM_MARG & ~KM_ONEIL>(ctx, ds, a, grad, nuis, sm, st);
    return dispatch2<P, KM_ALL>(ctx, ds, a, grad, nuis, sm, st);
}

}  // namespace octo
