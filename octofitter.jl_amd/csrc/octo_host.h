// octo_host.h — host-side internals shared by the translation units of the library: the opaque handle structs, the scratch /
// row-partition helpers and the declaration of the per-planet-count dispatch. The kernel templates are instantiated in
// octo_inst_p{1..4}.hip (one translation unit per planet count, so that they compile in parallel); octo_api.hip holds the
// C ABI and every non-template helper.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "octo_kernels.h"
#include "octo_model.h"
#include "octo_hgca.h"
#include "octo_small.h"
#include "octofitter_hip.h"

namespace octo {

struct TaskTable {
    uint64_t ds_serial = 0;          // the dataset this partition belongs to (octo_dataset::serial)
    bool nuis = false;               // planned with the nuisance kernels' row weights
    int64_t key = 0;                 // > 0: target number of tasks of the plan; < 0: forced uniform rows-per-wave (OCTO_CHUNK)
    int wpb = WPB;                   // waves per block the rows of a task are split over (8: the wide block of one-round launches)
    int n_tasks = 0;
    Task* d_tasks = nullptr;
    double* d_const_pre = nullptr;   // per-task constants for the no-nuisance path
    double* d_const_raw = nullptr;   // per-task constants for the nuisance path
    int32_t* d_obs_range = nullptr;  // [n_obs][2] task range of each observation
    double* d_obs_const_pre = nullptr, *d_obs_const_raw = nullptr;   // [n_obs] Σ of the task constants, in task order
    std::vector<Task> h_tasks;
};


}  // namespace octo

using namespace octo;      // host translation units only: the handle structs below hold kernel-side types

struct octo_dataset {
    int device = 0;
    int n_obs = 0, n_planets = 0;
    int kind_mask = 0;
    int64_t n_rows = 0;
    int n_hgca = 0;                      // OCTO_HGCA tables: evaluated by k_hgca, not by the epoch-loop kernel
    std::vector<DevObs> h_obs;
    std::vector<std::vector<double>> h_rowconst_pre, h_rowconst_raw;   // per obs, per row
    std::vector<std::vector<double>> h_step;                           // per obs, per row: |2π·(t − t of the previous row)| (0 in row 0, +Inf if not finite): Task::key_max
    DevObs* d_obs = nullptr;
    std::vector<double*> d_bufs;
    octo_planet_desc planets[MAXP];
    float tile_dm_ref = 0.0f;            // reference step of the walker-tile sort (octo_tile.h): the preferred rung of the largest table's ladder; 0: none
    int64_t tile_rows = 0;               // rows of the tables that have a ladder (what a homogeneous tile saves cold rows on)
    uint64_t serial = 0;                 // process-unique id: the contexts key their task-table caches by it (the dataset itself
                                         // is immutable after octo_dataset_create, so contexts may share it without locking)
};

struct octo_ofti {
    int device = 0;
    int64_t n = 0;
    double* d_rows = nullptr;
    double lambda = 0, data_quad = 0, log_det_data_cov = 0, log_det_prior_inv = 0, n_log2pi = 0;
};

struct octo_ctx {
    int device = 0;
    int n_cus = 256;
    int64_t max_lds = 65536;                    // largest dynamic LDS allocation one block may ask for on this device
    hipStream_t stream = nullptr;               // the context's own stream (OCTO_STREAM_CTX, and every host-buffer entry point)
    // The scratch below is reused by every evaluation, so evaluations through one context must be ordered. They are when the
    // caller keeps to one stream per context (the documented contract); if it does switch streams, the new stream is made to
    // wait for the last evaluation enqueued on the old one (an event, no host synchronisation).
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t ev_order = nullptr;
    octo_consts consts;
    std::string err;
    // scratch (device)
    int64_t cap_w = 0, cap_part = 0, cap_io = 0, cap_marg = 0;
    double* d_wc = nullptr;
    int32_t* d_valid = nullptr;
    double* d_partials = nullptr;
    double* d_marg = nullptr;
    double* d_extra = nullptr;                  // k_hgca output: ll and input-gradient of the non-epoch-loop terms
    int64_t cap_extra = 0;
    double* d_sctab = nullptr;                  // sin/cos grid of sincos_table, [SCT_N][2]
    int32_t* d_counters = nullptr;              // k_small: finished-block counter per walker, [SMALL_W], zero between launches
    uint64_t* h_flags = nullptr;                // mapped pinned [SMALL_W]: k_small's finishing block of walker w stores the call's
    uint64_t flag_seq = 0;                      // sequence number here after its outputs; host-buffer calls spin on it
    bool flag_request = false, flag_armed = false;
    SmallInline inl = {};                       // n > 0 while a one-θ host-buffer call hands k_small its inputs inside the kernel arguments
    int64_t stage_ws_in = 0, stage_ws_out = 0;  // > 0 while octo_eval hands k_small its walker-major staging buffers
    // octo_model_logpost_device (big batches): the model's tail for k_finish (EvalArgs::mt_*) travels as an explicit argument of eval_impl;
    // mt_applied is how the launch code reports that a k_finish launch carried it (reset by eval_impl on every call)
    struct ModelTail { const double *Jc = nullptr, *gtp = nullptr, *glp = nullptr, *lpp = nullptr; const octo_source *esrc = nullptr, *nsrc = nullptr; double *lp = nullptr, *grad = nullptr; int64_t ld = 0, ldo = 0; int32_t D = 0, n_nu = 0; };
    bool mt_applied = false;
    // octo_eval_begin .. octo_eval_end: what is still to be waited for and copied out
    struct Pending {
        bool active = false, staged = false, walker_major = false;
        int64_t W = 0, ld = 0, ldd = 0, ws_out = 0, o_ge = 0, o_gn = 0;
        int n_el_out = 0, n_nu_out = 0;
        double *ll = nullptr, *g_elems = nullptr, *g_nuis = nullptr;
    } pending;
    // parallel tempering over RCCL (octo_comm.hip)
    void* comm = nullptr;                       // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    int small_w_model = 768;                    // … and single-planet whole-callback batches (octo_model_logpost*) up to this size: three launches is what they would pay otherwise
    int small_w = OCTO_SMALL_BATCH_DEFAULT;     // batches up to this size take the fused small-batch launch (OCTO_SMALL_W: experiments)
    // experiment knobs, read from the environment ONCE at context creation (0 = not set): a getenv per call is a linear scan of the
    // environment on a 12 µs path
    int64_t env_small_blocks = 0, env_small_min_span = 0, env_stage_bytes = 0, env_chunk = 0, env_rounds = 0, env_rv_cost = 0, env_kind_all = 0, env_mainp_tpb = 0;
    int env_warm = 1;                           // OCTO_WARM=0: experiments and tests — datasets created by this context never take k_main's warm-started row loop (DevObs::dm_max = 0)
    // ---- options (octo_ctx_set_option; include/octofitter_hip.h: OCTO_OPT_*)
    int opt_warm = 1;                           // OCTO_OPT_WARM_START: 0 = every launch of this context takes k_main's cold row loop
    int opt_invariant = 0;                      // OCTO_OPT_BATCH_INVARIANT: results independent of the batch's size and composition (cold loop, no tile sort,
                                                // one fixed row partition, no small-batch route)
    int tile_mode = 2;                          // OCTO_OPT_TILE_SORT: 0 never, 1 every eligible evaluation, 2 when a probe says it pays (octo_api.hip: tile_prepare)
    int64_t tile_min_w = 2048;                  // OCTO_OPT_TILE_MIN_WALKERS: smaller batches are never sorted (a strong-scaled shard: the launch would cost more than it saves)
    // the walker-tile sort's state (octo_tile.h)
    int32_t* d_perm = nullptr; int64_t cap_perm = 0;
    float* h_tile_stats = nullptr; int64_t cap_tile_stats = 0;      // mapped pinned: k_tile_sort's expected cold wave-rows per row, as given | sorted, per segment
    hipEvent_t ev_tile = nullptr;
    bool tile_on = false, tile_pending = false;
    int tile_pending_segs = 0;
    int64_t tile_seq = 0, tile_W = 0, tile_sorted_launches = 0, tile_probes = 0;
    uint64_t tile_ds = 0;
    double tile_last_saving_us = 0.0;
    int env_no_fin_fused = 0;                   // OCTO_FIN_FUSED=0: experiments and tests — one-task launches keep the k_finish launch
    int env_wide = 0;                           // OCTO_WIDE: experiments (1: eight-wave k_main blocks for every one-round single-planet launch, -1: never)
    int flag_w = 128;                           // ... and signal completion through per-walker flags the host spins on (OCTO_FLAG_W: experiments)
    int mapped_w = 128;                         // host-buffer calls up to this size let k_small read/write mapped pinned memory; larger
                                                // ones cross the link as one DMA each way (OCTO_MAPPED_W: experiments)
    double *d_in = nullptr, *d_out = nullptr;   // staging for octo_eval (host buffers)
    int64_t cap_in = 0, cap_out = 0;
    double *h_in = nullptr, *h_out = nullptr;   // pinned mirrors of d_in / d_out for SMALL batches: one transfer each way instead of
    int64_t cap_hin = 0, cap_hout = 0;          // one per array — what a single-chain sampler's per-gradient latency is made of
    std::vector<void*> retired_host;            // outgrown mapped-pinned buffers (freed at octo_ctx_destroy: hipHostFree waits for the device)
    std::vector<void*> retired;                 // outgrown scratch buffers: kernels already enqueued may still use them, so they
                                                // are freed at the next host-blocking point (octo_sync, the end of a host-buffer
                                                // call, octo_ctx_destroy) instead of by a device-synchronising hipFree mid-stream
    // row partitions ("task tables") of the datasets this context has evaluated, keyed by (dataset serial, plan key)
    std::vector<octo::TaskTable> tables;
    std::map<uint32_t, int> occupancy;          // resident blocks per CU of each k_main variant (P, NUIS, KM) on THIS device
    std::map<const void*, size_t> lds_raised;   // kernels whose dynamic-LDS limit this context has raised beyond the default 48 KB (raise_dynamic_lds)
    // timing
    int timing_every = 0;                       // 0 = off, n = bracket every n-th evaluation's k_main with events
    bool timing_whole = false;                  // octo_timing_enable(ctx, -1): bracket every host-buffer evaluation WHOLE (copy-in .. last store)
    int64_t timing_seq = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double t_ms = 0.0;
    int64_t t_n = 0;
    std::vector<float> t_samples;               // every timed launch since the last reset (median, spread)
};


namespace octo {

int fail(octo_ctx* ctx, int code, const std::string& msg);

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? OCTO_ENOMEM : OCTO_EHIP,                  \
                        std::string(#call) + ": " + hipGetErrorString(e_));                        \
    } while (0)

template <typename T>
int grow(octo_ctx* ctx, T*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return OCTO_OK;
    if (p) { ctx->retired.push_back((void*)p); p = nullptr; cap = 0; }
    const int64_t n = need + need / 2;
    HIPCHK(ctx, hipMalloc((void**)&p, sizeof(T) * (size_t)n));
    cap = n;
    return OCTO_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for a launch that needs more than the default 48 KB, once per context and kernel
inline int raise_dynamic_lds(octo_ctx* ctx, const void* fn, size_t lds, const char* what) {
    if (lds <= 48 * 1024) return OCTO_OK;
    size_t& r = ctx->lds_raised[fn];
    if (lds <= r) return OCTO_OK;
    if (lds > (size_t)ctx->max_lds) return fail(ctx, OCTO_EINVAL, std::string(what) + ": the block shape needs more LDS than this device has");
    HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    r = lds;
    return OCTO_OK;
}

int get_tasks(octo_ctx* ctx, const octo_dataset* ds, int64_t key, TaskTable** out, bool nuis = false, int wpb = WPB);
int64_t plan_key(const octo_ctx* ctx, int64_t W, int64_t n_rows, int blocks_per_cu, bool* wide = nullptr, int blocks8_per_cu = 0);
int busy(octo_ctx* ctx, const char* what);      // OCTO_EINVAL while an octo_eval_begin of this context is outstanding
bool small_eligible(const octo_ctx* ctx, const octo_dataset* ds, int64_t W, bool model = false);
int tile_prepare(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, hipStream_t st);      // octo_tile.h: sets a.perm (or leaves it null)

// Launch of one evaluation for a dataset with P planets (octo_launch.h; instantiated in octo_inst_p<P>.hip).
template <int P>
int dispatch1(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, bool grad, bool nuis, const SmallModel* sm, hipStream_t st);
inline bool hgca_in_small(int64_t W) { return W <= 16; }      // k_small computes the HGCA term itself (else k_hgca ahead of it)
constexpr int64_t SMALL_MARG_ROWS = 16384;      // longest marginalised-RV table k_small takes (one block runs it twice)
constexpr int64_t SMALL_KEY = (int64_t)1 << 40;   // get_tasks keys at or below −SMALL_KEY: k_small's row partition
constexpr int64_t STAGE_DMA_BYTES = 1 << 20;   // host-buffer calls up to this size (inputs + outputs) are staged in pinned memory

// The planet-per-wave kernels (octo_mainp.h), instantiated ONCE in octo_inst_pn.hip for the kind sets they are compiled for: km_p = 35
// (RA/Dec, sep/PA, cor), 55 (+ absolute and relative RV) or 63 (+ marginalised RV: round 6, more than four planets only). launch_mainp: k_mainp on a planned grid; launch_finishp: k_finishp (P > MAXP_T).
int mainp_occupancy(octo_ctx* ctx, bool nuis, int km_p, int P);
int launch_mainp(octo_ctx* ctx, bool grad, bool nuis, int km_p, int64_t cols, const EvalArgs& a, hipStream_t st);
int launch_finishp(octo_ctx* ctx, bool grad, bool nuis, int km_p, int64_t cols, const EvalArgs& a, hipStream_t st);
constexpr int mainp_kind_set(int km_all) {
    const int km = km_all & ~KM_HGCA;      // (an HGCA table has no rows in the epoch loop: k_hgcap -> `extra` -> k_finishp)
    return (km & ~(KM_RADEC | KM_SEPPA | KM_COR)) == 0 ? (KM_RADEC | KM_SEPPA | KM_COR) : ((km & (KM_MARG | KM_ONEIL)) ? KM_ALL : (KM_ALL & ~KM_MARG & ~KM_ONEIL));
}
int launch_margp(octo_ctx* ctx, bool nuis, int km_p, const EvalArgs& a, hipStream_t st);      // k_marg for the planet-per-wave kernels' forward partials
// more planets than the templated kernels are compiled for: planner + k_mainp + k_finishp (octo_inst_pn.hip)
int dispatch_many(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, bool grad, bool nuis, const SmallModel* sm, hipStream_t st);
int64_t plan_key_mainp(const octo_ctx* ctx, int64_t W, int64_t n_rows, int blocks_per_cu);

extern template int dispatch1<1>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
extern template int dispatch1<2>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
extern template int dispatch1<3>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
extern template int dispatch1<4>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);

}  // namespace octo
