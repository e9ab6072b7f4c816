// octo_api.hip — C ABI of include/octofitter_hip.h over the HIP runtime (gfx950 only).
// Host logic only: uploads, task tables, scratch, kernel dispatch, timing. No CPU compute path:
// every entry point that evaluates fails with OCTO_ENODEV / OCTO_EHIP when no device is usable.
#define OCTO_API_TU 1      // this translation unit owns the non-template kernels (octo_model.h, octo_kernels.h)
#include "octo_host.h"
#include "octo_tile.h"

using namespace octo;

namespace octo {

std::atomic<uint64_t> g_dataset_serial{1};

int fail(octo_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// The host-buffer entry points share the context's staging buffers (h_in / h_out, d_in / d_out) and completion flags with an
// outstanding octo_eval_begin: none of them may run until the matching octo_eval_end.
int busy(octo_ctx* ctx, const char* what) {
    if (ctx->pending.active) return fail(ctx, OCTO_EINVAL, std::string(what) + ": an octo_eval_begin of this context has not been ended");
    return OCTO_OK;
}


DevConsts dev_consts(const octo_consts& c) {
    DevConsts d;
    d.k_yr = c.kepler_year_to_julian_day; d.yd = c.year2day_julian; d.au2m = c.au2m; d.sec2yr = c.sec2year_julian;
    // cart2angle = rad2as·1e3 / (1000/plx · pc2au)  =  plx · rad2as/pc2au   (parameterizations.jl:215-216)
    d.mas_per_au_per_plx = c.rad2as / c.pc2au;
    d.mjup2msol = c.mjup2msol;
    return d;
}

bool grow_pinned(double*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return true;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    if (hipHostMalloc((void**)&p, sizeof(double) * (size_t)(2 * need), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { p = nullptr; return false; }
    cap = 2 * need;
    return true;
}

// Host-buffer calls of a few hundred parameter sets on k_small: the rows arrive in mapped pinned memory as the caller has them
// (SoA, walker fastest); this kernel reads them over PCIe — coalesced along the walkers — and leaves them walker-major in device
// memory, where each k_small block finds its walker's inputs contiguous. Two copy-engine transfers of ~40 KB cost ~17 µs per
// call; this launch ~4 µs, and the results go back through the mapped buffer + per-walker flags like the smallest batches.
// ---- caller-registered host ranges (octo_host_register): process-wide, keyed by base address
struct HostRange { size_t bytes; char* dev; int device; };
static std::mutex g_reg_mu;
static std::map<uintptr_t, HostRange> g_reg;

// device-side address of the host range [p, p + bytes) if it lies inside one registered range for this device, else null
static void* mapped_range(int device, const void* p, size_t bytes) {
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_reg.empty()) return nullptr;
    auto it = g_reg.upper_bound((uintptr_t)p);
    if (it == g_reg.begin()) return nullptr;
    --it;
    const uintptr_t base = it->first;
    if ((uintptr_t)p + bytes > base + it->second.bytes || it->second.device != device) return nullptr;
    return it->second.dev + ((uintptr_t)p - base);
}

static int64_t stage_bytes(const octo_ctx* ctx) {      // host-buffer calls up to this size (inputs + outputs) are staged in mapped pinned memory (OCTO_STAGE_BYTES: experiments)
    return ctx->env_stage_bytes > 0 ? ctx->env_stage_bytes : STAGE_DMA_BYTES;
}

static int64_t env_int(const char* name) {
    const char* ev = std::getenv(name);
    if (!ev) return 0;
    const long long v = std::atoll(ev);
    return v > 0 ? (int64_t)v : 0;
}

static __global__ __launch_bounds__(256) void k_copy_in(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

static __global__ __launch_bounds__(256) void k_stage_in(const double* __restrict__ src, int64_t ld_src, int64_t W, int n_rows, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * n_rows) return;
    const int64_t r = i / W, w = i - r * W;
    dst[w * n_rows + r] = src[r * ld_src + w];
}

// `hip_stream` of the *_device entry points: OCTO_STREAM_CTX = the context's own stream; anything else is handed to HIP as
// it is, so NULL is HIP's NULL stream (the legacy default stream — what torch.cuda.current_stream().cuda_stream is when no
// torch stream is active). Orders the scratch across a change of stream.
int use_stream(octo_ctx* ctx, void* hip_stream, hipStream_t* out) {
    hipStream_t st = hip_stream == OCTO_STREAM_CTX ? ctx->stream : (hipStream_t)hip_stream;
    if (ctx->has_last && ctx->last_stream != st) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_order, ctx->last_stream));
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_order, 0));
    }
    ctx->last_stream = st; ctx->has_last = true;
    *out = st;
    return OCTO_OK;
}

// End of a small host-buffer call: if k_small was handed the completion flags, spin on them (its results are already in host
// memory when a flag flips; ~3 µs less than a stream synchronisation), looking at the stream now and then so that a failed
// launch cannot hang the caller; otherwise synchronise the stream.
int wait_small(octo_ctx* ctx, hipStream_t st, int64_t W) {
    if (ctx->flag_armed) {
        ctx->flag_armed = false;
        volatile uint64_t* f = ctx->h_flags;
        const uint64_t seq = ctx->flag_seq;
        uint32_t spins = 0;
        for (int64_t w = 0; w < W; ++w) {
            while (f[w] != seq) {
                if ((++spins & 0x3fff) == 0) {
                    const hipError_t q = hipStreamQuery(st);
                    if (q == hipSuccess) break;                 // the kernel has retired: the flag store is visible by now
                    if (q != hipErrorNotReady) return fail(ctx, OCTO_EHIP, std::string("k_small: ") + hipGetErrorString(q));
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return OCTO_OK;
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    return OCTO_OK;
}

void free_retired(octo_ctx* ctx) {      // only after the streams that used them have been synchronised
    for (void* p : ctx->retired) (void)hipFree(p);
    ctx->retired.clear();
}

void free_table(TaskTable& t) {
    (void)hipFree(t.d_tasks); (void)hipFree(t.d_const_pre); (void)hipFree(t.d_const_raw);
    (void)hipFree(t.d_obs_range); (void)hipFree(t.d_obs_const_pre); (void)hipFree(t.d_obs_const_raw);
}


// Relative cost of one row of a table (VALU instructions per row, from the ISA of the mixed-kind kernels): used only to
// balance block durations across tables of different kinds.
// Relative issue time of one row of each kind, in units of an RA/Dec row of the same kernel variant: what the planner shares the tasks
// between the tables by. A one-round launch (the multi-planet kernels: grid = what the chip holds at once) ends with its slowest CU, so
// the weights decide the step time: config 4 (2 planets, RA/Dec + absolute RV with nuisances) ran 3 astrometry tasks of 834 rows and 5 RV
// tasks of 500 rows per tile with the single-planet weight 1.35 for an RV row — the CUs holding an astrometry block were busy for 88k
// instructions per SIMD, the others for 57k. In that kernel an RV row costs LESS than an astrometry row (227 against 287 VALU
// instructions: the astrometry row carries jitter, platescale and northangle, two sky projections and their adjoints).
// Calibrated on MI355X with tools/sweep_rv_cost.py (profiles/r3_rv_cost_sweep.txt); OCTO_RV_COST=<percent> overrides it for experiments.
double row_cost(const octo_ctx* ctx, int kind, int n_planets, bool nuis) {
    const bool rv = kind == OCTO_RV_ABS || kind == OCTO_RV_ABS_MARG || kind == OCTO_RV_REL;
    if (rv && ctx->env_rv_cost > 0) return 0.01 * (double)ctx->env_rv_cost;
    switch (kind) {
        case OCTO_ASTROM_SEPPA: case OCTO_ONEIL_SEPPA: return 1.8;
        case OCTO_RV_ABS: case OCTO_RV_ABS_MARG: case OCTO_RV_REL: return n_planets > 1 ? (nuis ? 0.9 : 1.0) : (nuis ? 1.0 : 1.35);
        case OCTO_ONEIL_RADEC: return 1.1;
        default: return 1.0;
    }
}

// Build (or fetch) the row partition. One block = WPB waves × `chunk` rows of ONE table for one walker tile; tasks never
// straddle tables and tables keep their order, so k_finish can sum each observation's partials contiguously and in a
// fixed order. key <= −SMALL_KEY: k_small's partition, −key − SMALL_KEY rows per wave; key > 0: about `key` tasks in total, shared between the tables in proportion to rows × row cost, each
// table cut into EQUAL tasks (no ragged last task); key < 0: −key rows per wave everywhere (OCTO_CHUNK experiments).
int get_tasks(octo_ctx* ctx, const octo_dataset* ds, int64_t key, TaskTable** out, bool nuis, int wpb) {
    // The cache belongs to the CONTEXT (one owner thread), not to the dataset, which stays immutable and can therefore be
    // shared between contexts and host threads.
    for (auto& t : ctx->tables)
        if (t.ds_serial == ds->serial && t.key == key && t.nuis == nuis && t.wpb == wpb) { *out = &t; return OCTO_OK; }
    if (ctx->tables.size() >= 48) {                           // many datasets / batch sizes: drop the oldest half. hipFree waits
        for (size_t k = 0; k < 24; ++k) free_table(ctx->tables[k]);      // for the device, so no launched kernel still reads them
        ctx->tables.erase(ctx->tables.begin(), ctx->tables.begin() + 24);
    }
    TaskTable tt;
    tt.ds_serial = ds->serial;
    tt.key = key;
    tt.nuis = nuis;      // the row weights differ between the nuisance and the nuisance-free kernels
    tt.wpb = wpb;
    std::vector<double> cpre, craw;
    double wsum = 0.0;
    for (int o = 0; o < ds->n_obs; ++o)
        if (ds->h_obs[o].kind != OCTO_HGCA) wsum += (double)ds->h_obs[o].n * row_cost(ctx, ds->h_obs[o].kind, ds->n_planets, nuis);
    // Rows per wave: at least 32 (short blocks pay their prologue, LDS combine and partial store more often, and k_finish walks every
    // task's partials: 1e4 rows × 1024 walkers 56 µs at 32, 62 at 16) — unless the tables are so short that this leaves a handful of
    // blocks whose waves each grind through 30-50 rows one after the other (a 3-planet row is ~1 µs of dependent issue for a lone
    // wave): then down to 8, keeping the task count <= ~32 (4 planets, 440 rows, 1024 walkers: 165 -> 121 µs).
    int64_t rows_min = 32;
    {
        int64_t n_all = 0, t32 = 0;
        for (int o = 0; o < ds->n_obs; ++o)
            if (ds->h_obs[o].kind != OCTO_HGCA && ds->h_obs[o].n > 0) { n_all += ds->h_obs[o].n; t32 += std::max<int64_t>(1, ds->h_obs[o].n / (32 * wpb)); }
        if (t32 < 32) rows_min = std::min<int64_t>(32, std::max<int64_t>(8, n_all / (32 * wpb)));
    }
    // key > 0: `key` tasks shared between the tables in proportion to rows x row cost — largest-remainder apportionment, so that the
    // shares add up to the target (independent rounding gave 7 or 9 tasks for a target of 8: a one-round grid then leaves CUs idle or
    // spills into a second round)
    std::vector<int64_t> t_plan(std::max(ds->n_obs, 1), 0);
    if (key > 0 && wsum > 0.0) {
        std::vector<std::pair<double, int>> frac;
        int64_t given = 0;
        for (int o = 0; o < ds->n_obs; ++o) {
            if (ds->h_obs[o].kind == OCTO_HGCA || ds->h_obs[o].n <= 0) continue;
            const double x = (double)key * (double)ds->h_obs[o].n * row_cost(ctx, ds->h_obs[o].kind, ds->n_planets, nuis) / wsum;
            t_plan[o] = std::max<int64_t>(1, (int64_t)std::floor(x));
            given += t_plan[o];
            frac.emplace_back(x - std::floor(x), o);
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
        for (size_t k = 0; k < frac.size() && given < key; ++k) { t_plan[frac[k].second] += 1; given += 1; }
    }
    for (int o = 0; o < ds->n_obs; ++o) {
        if (ds->h_obs[o].kind == OCTO_HGCA) continue;          // no epoch-loop rows (k_hgca)
        const int64_t n = ds->h_obs[o].n;
        if (n <= 0) continue;
        int64_t chunk;
        if (key <= -SMALL_KEY) chunk = (ds->h_obs[o].kind == OCTO_RV_ABS_MARG) ? (n + wpb - 1) / wpb : -(key + SMALL_KEY);   // k_small: a marginalised-RV table in ONE block (its μ̂)
        else if (key < 0) chunk = -key;
        else {
            int64_t t_o = t_plan[o];
            t_o = std::min<int64_t>(std::max<int64_t>(t_o, 1), std::max<int64_t>(1, n / (rows_min * wpb)));
            const int64_t rows_per_task = (n + t_o - 1) / t_o;
            chunk = (rows_per_task + wpb - 1) / wpb;
        }
        const int64_t span = chunk * wpb;
        for (int64_t r0 = 0; r0 < n; r0 += span) {
            Task t;
            std::memset(&t, 0, sizeof(t));
            t.obs = o; t.row0 = (int32_t)r0; t.nrows = (int32_t)std::min<int64_t>(span, n - r0); t.chunk = (int32_t)chunk;
            {   // largest step among the rows that a wave may solve warm (every row but the first of each wave's slice), rounded up to a float
                double km = 0.0;
                for (int64_t r = r0; r < r0 + t.nrows; ++r)
                    if ((r - r0) % chunk != 0) km = std::max(km, (double)ds->h_step[o][r]);
                float f = (float)km;
                if ((double)f < km) f = std::nextafterf(f, INFINITY);
                t.key_max = std::isfinite(km) ? f : INFINITY;
            }
            tt.h_tasks.push_back(t);
            double a = 0.0, b = 0.0;
            for (int64_t r = r0; r < r0 + t.nrows; ++r) { a += ds->h_rowconst_pre[o][r]; b += ds->h_rowconst_raw[o][r]; }
            cpre.push_back(a); craw.push_back(b);
        }
    }
    tt.n_tasks = (int)tt.h_tasks.size();
    {   // per observation: its task range and the sum of its task constants (same order a sequential device loop would use)
        std::vector<int32_t> range(2 * std::max(ds->n_obs, 1), 0);
        std::vector<double> opre(std::max(ds->n_obs, 1), 0.0), oraw(std::max(ds->n_obs, 1), 0.0);
        int t = 0;
        for (int o = 0; o < ds->n_obs; ++o) {
            range[2 * o] = t;
            while (t < tt.n_tasks && tt.h_tasks[t].obs == o) { opre[o] += cpre[t]; oraw[o] += craw[t]; ++t; }
            range[2 * o + 1] = t;
        }
        HIPCHK(ctx, hipMalloc((void**)&tt.d_obs_range, sizeof(int32_t) * range.size()));
        HIPCHK(ctx, hipMalloc((void**)&tt.d_obs_const_pre, sizeof(double) * opre.size()));
        HIPCHK(ctx, hipMalloc((void**)&tt.d_obs_const_raw, sizeof(double) * oraw.size()));
        HIPCHK(ctx, hipMemcpy(tt.d_obs_range, range.data(), sizeof(int32_t) * range.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(tt.d_obs_const_pre, opre.data(), sizeof(double) * opre.size(), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(tt.d_obs_const_raw, oraw.data(), sizeof(double) * oraw.size(), hipMemcpyHostToDevice));
    }
    if (tt.n_tasks > 0) {
        HIPCHK(ctx, hipMalloc((void**)&tt.d_tasks, sizeof(Task) * tt.n_tasks));
        HIPCHK(ctx, hipMalloc((void**)&tt.d_const_pre, sizeof(double) * tt.n_tasks));
        HIPCHK(ctx, hipMalloc((void**)&tt.d_const_raw, sizeof(double) * tt.n_tasks));
        HIPCHK(ctx, hipMemcpy(tt.d_tasks, tt.h_tasks.data(), sizeof(Task) * tt.n_tasks, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(tt.d_const_pre, cpre.data(), sizeof(double) * tt.n_tasks, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(tt.d_const_raw, craw.data(), sizeof(double) * tt.n_tasks, hipMemcpyHostToDevice));
    }
    if (tt.n_tasks > 65535) {      // grid.y limit; only reachable through the OCTO_CHUNK / OCTO_ROUNDS experiment knobs
        free_table(tt);
        return fail(ctx, OCTO_EINVAL, "row partition has more than 65535 tasks (OCTO_CHUNK / OCTO_ROUNDS too aggressive)");
    }
    ctx->tables.push_back(std::move(tt));
    *out = &ctx->tables.back();
    return OCTO_OK;
}

// How many tasks. The work of a row is the same for every walker (the Kepler solve is non-iterative), so a static
// partition is balanced by construction; what is left to choose is the grain. Measured on MI355X (tools/sweep_chunk.py,
// profiles/README.md): for the 7-blocks-per-CU single-planet kernels the step time at 1e4 × 1e4 is flat within 1 % for
// 48-96 rows per wave and rises on both sides — short blocks pay their prologue (table fill, per-walker constants), LDS
// combine and partial store more often and leave more partials for k_finish; one "round" of long blocks (grid = what
// the chip holds at once) is ~8 % slower there because lock-stepped waves line up those phases instead of overlapping
// them with other waves' arithmetic. The 2-blocks-per-CU multi-planet kernels have nothing to overlap with and are best
// at exactly one round (config 4: 209 µs at 1 round, 215 at 2, 246 at 1.75). So: grid = R × resident capacity with
// R = 3 at 6-7 blocks per CU down to 1 at 1-3, as EXACTLY as the table sizes allow (a fractional last round is idle
// hardware), tasks equal within a table.
// (A persistent kernel pulling (task, tile) items from an atomic queue, with and without a tapered item size, was
// measured against this grid-mapped launch in the same run and was not faster at any batch size: the hardware
// dispatcher already backfills freed slots fast enough for an FP64-issue-bound kernel.)
int64_t plan_key(const octo_ctx* ctx, int64_t W, int64_t n_rows, int blocks_per_cu, bool* wide, int blocks8_per_cu) {
    const int n_cus = ctx->n_cus;
    if (wide) *wide = false;
    if (ctx->env_chunk > 0) return -ctx->env_chunk;      // OCTO_CHUNK, tuning knob for experiments: uniform rows per wave
    const int64_t cols = (W + WAVE - 1) / WAVE;
    const int64_t capacity = std::max<int64_t>((int64_t)blocks_per_cu * n_cus, 256);
    int64_t rounds = std::min<int64_t>(std::max<int64_t>(std::llround(blocks_per_cu * 3.0 / 7.0), 1), 3);
    // few walker tiles: fewer rounds rather than blocks shorter than ~48 rows per wave (W = 4096: 166 µs at 1 round, 177 at 3)
    while (rounds > 1 && n_rows * cols < 48 * WPB * rounds * capacity) --rounds;
    if (ctx->env_rounds > 0) rounds = ctx->env_rounds;      // OCTO_ROUNDS: experiments
    const int64_t key0 = std::max<int64_t>(1, rounds * capacity / cols);
    if (rounds != 1 || ctx->env_rounds > 0 || n_rows < 32 * WPB * 8) return key0;      // (short tables: get_tasks' own rule, calibrated on them)
    // One round (few walker tiles — a strong-scaled shard): every block of the grid is resident at once, each block puts one
    // wave on each SIMD of its CU, so the launch lasts (waves on the fullest SIMD) x (rows per wave), and the task count decides both
    // factors in steps. Filling every slot (1 250 walkers: 76 tasks x 20 tiles, 6 waves per SIMD x 33 rows) is no faster than 38 tasks (3
    // waves x 66 rows) — three waves nearly saturate the FP64 issue port — while every task costs k_finish a partial to fetch:
    // 55.2 -> 53.1 µs per step (tools/step_probe.py, profiles/r4_chunk_sweep_1250.txt); two waves per SIMD are NOT enough (100 rows per
    // wave: 51.9 µs of k_main against 48.3), one wave even less. Cost model fitted to that sweep, in units of one row of one wave on a
    // SIMD that holds seven: n waves on a SIMD take n·g(n) per row, g = 1.59, 1.13, 1.045, 1.03, 1.015, 1.005, 1 for n = 1 … 7;
    // a block's prologue + combine ~ 2.7 rows; a task costs k_finish ~ 0.34.
    const int64_t t_max = std::max<int64_t>(1, n_rows / (32 * WPB));
    int64_t best_t = key0;
    double best_cost = 1e300;
    for (int wps = 1; wps <= blocks_per_cu; ++wps) {
        int64_t t = std::min<int64_t>((int64_t)wps * n_cus / cols, t_max);
        if (t < 1) continue;
        const double chunk = std::ceil((double)n_rows / (double)(t * WPB));
        const double n = std::ceil((double)(cols * t) / (double)n_cus);
        static const double g[8] = {1.59, 1.59, 1.13, 1.045, 1.03, 1.015, 1.005, 1.0};
        const double cost = n * g[(int)std::min(n, 7.0)] * (chunk + 2.7) + 0.34 * (double)t;
        if (cost < best_cost) { best_cost = cost; best_t = t; }
    }
    // The WIDE block (k_main<…, NWV = 8>, offered by the caller for the kernels it is compiled for): the same task count with eight waves per
    // block — twice the waves per SIMD for the row loop, the per-block costs (table fill, orbit-constructor pieces, partials, k_finish) unchanged.
    if (wide && ctx->env_wide >= 0) {
        const double n4 = std::ceil((double)(cols * best_t) / (double)n_cus);
        // (n4 blocks per CU in the chosen partition: as eight-wave blocks they need n4 <= the eight-wave kernel's own occupancy — it is held to a
        // register count of its own, octo_kernels.h: main_min_waves — while a wave keeps >= 32 rows)
        *wide = (n4 <= (double)blocks8_per_cu && n_rows >= (int64_t)best_t * 8 * 32) || (ctx->env_wide > 0 && n4 <= (double)blocks8_per_cu);
    }
    return best_t;
}

// Small and mid-size batches: one fused launch, lane = epoch (octo_small.h: k_small). Its cost grows with the number of blocks
// (one per walker, each deriving P orbits and running the finish), the throughput kernels' with three launches: measured
// crossover at W·P ≈ 400-1000 (tools/latency_vs_w.py, tools/latency_multi.py). An HGCA table adds blocks to the same launch
// (one input direction per wave) for W <= 16, the k_hgca launch ahead of it otherwise.
// model: the whole callback (octo_model_logpost*). There the alternative is THREE launches (k_model_fwd, k_main, k_finish), so the fused launch pays for
// longer: single-planet D = 11 callback at 768 θ_t 40-41 µs fused against 43 on the three-launch route, 46-48 against 45 at 1 024
// (tools/r5_midsize_small.py) — the limit there is ctx->small_w_model (768 unless the caller set a limit of its own).
bool small_eligible(const octo_ctx* ctx, const octo_dataset* ds, int64_t W, bool model) {
    if (ctx->opt_invariant) return false;          // OCTO_OPT_BATCH_INVARIANT: one kernel family, one row partition
    if (ds->n_planets > MAXP_T) return false;      // k_small<P> is compiled for 1 … 4 planets
    const int limit = (model && ds->n_planets == 1) ? std::max(ctx->small_w, ctx->small_w_model) : ctx->small_w;
    if (!(W * ds->n_planets <= limit && W <= SMALL_W)) return false;
    if (ds->kind_mask & KM_MARG)      // a marginalised-RV table is ONE block's work there (two passes): not for a very long table
        for (int o = 0; o < ds->n_obs; ++o)
            if (ds->h_obs[o].kind == OCTO_RV_ABS_MARG && ds->h_obs[o].n > SMALL_MARG_ROWS) return false;
    return true;
}

// The walker-tile sort ahead of a single-planet fused k_main launch (octo_tile.h). Mode 2 (default): every TILE_PROBE_EVERY-th eligible evaluation of
// a (dataset, batch size) PRICES the sort — k_tile_sort leaves the expected number of cold wave-rows per row, in the order given and sorted,
// in mapped host memory; the NEXT eligible evaluation waits for that launch (an event; the kernel ran a whole evaluation ago), reads the two numbers
// and keeps the sort on while  (Δ cold wave-rows per row) x rows x the cost of a cold row > the cost of the launch.  The decision is taken at a fixed
// point of the call sequence, so a rerun of the same calls makes the same decisions: results stay bit-reproducible run to run.
constexpr int TILE_PROBE_EVERY = 64;
constexpr double TILE_COLD_ROW_US = 27.0 * 2.15e-3;      // a cold row costs ~27 more VALU instructions than a warm one, 2.15 ns of a SIMD's issue each (tools/ubench.hip)
constexpr double TILE_LAUNCH_US = 6.5;                   // the kernel (4-5 µs at 1e4 walkers) + one dependent launch on the stream + the gathers through perm (profiles/r6_tile_trace.txt)

// read the pending probe: wait for its launch, sum the segments' two numbers, decide
static int tile_settle(octo_ctx* ctx, const octo_dataset* ds) {
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_tile));
    double d = 0.0;
    for (int k = 0; k < ctx->tile_pending_segs; ++k) d += (double)ctx->h_tile_stats[2 * k] - (double)ctx->h_tile_stats[2 * k + 1];
    ctx->tile_last_saving_us = d * (double)ds->tile_rows * TILE_COLD_ROW_US / (4.0 * (double)ctx->n_cus);
    ctx->tile_on = ctx->tile_last_saving_us > 1.3 * TILE_LAUNCH_US;
    ctx->tile_pending = false;
    return OCTO_OK;
}

int tile_prepare(octo_ctx* ctx, const octo_dataset* ds, EvalArgs& a, hipStream_t st) {
    a.perm = nullptr;
    const int kp = ds->n_planets - 1;      // the planet whose severity is the key: the one planet, or the last of two (main_warm_last)
    if (!a.warm || ctx->tile_mode == 0 || a.W < ctx->tile_min_w || !(ds->tile_dm_ref > 0.0f) || ds->n_planets > 2 ||
        ds->planets[kp].orbit_kind == OCTO_ORBIT_THIELE_INNES)      // (a ThieleInnesOrbit's period needs its constructor: left as drawn)
        return OCTO_OK;
    bool probe = false, sort_now = ctx->tile_mode == 1;
    // Two planets: only when forced (mode 1). Their launches are ONE round of blocks (two per CU), which lasts as long as its slowest block: the tiles
    // that collect the severe lanes keep the launch at their duration whatever the others gain (measured with the per-wave criterion the loop started with:
    // 155.2 µs as drawn, 156.8 sorted, the sort's own 3.8 µs included — profiles/r6_cfg4_warm_last.txt), and with the per-row test the loop has now ~5 % of
    // config 4's wave-rows are cold as drawn: nothing left for a sort to collect (profiles/r6_cfg4_dyn.txt).
    if (ctx->tile_mode == 2 && ds->n_planets > 1) return OCTO_OK;
    if (ctx->tile_mode == 2) {
        if (ctx->tile_ds != ds->serial || ctx->tile_W != a.W) {      // another dataset or batch size: start over (a pending probe of the old shape is dropped)
            ctx->tile_ds = ds->serial; ctx->tile_W = a.W; ctx->tile_seq = 0; ctx->tile_on = false; ctx->tile_pending = false;
        }
        if (ctx->tile_pending) { int rcd = tile_settle(ctx, ds); if (rcd) return rcd; }
        probe = (ctx->tile_seq % TILE_PROBE_EVERY) == 0;
        ctx->tile_seq += 1;
        sort_now = ctx->tile_on;      // (a probe prices the sort, it does not apply it: the same input evaluated twice in a row gives the same bits
                                      // unless the decision itself changes in between)
    }
    if (!sort_now && !probe) return OCTO_OK;
    const int n_seg = (int)((a.W + TILE_SEG - 1) / TILE_SEG);
    int rc = grow(ctx, ctx->d_perm, ctx->cap_perm, a.W);
    if (rc) return rc;
    if (2 * (int64_t)n_seg > ctx->cap_tile_stats) {
        if (ctx->h_tile_stats) ctx->retired_host.push_back((void*)ctx->h_tile_stats);
        ctx->h_tile_stats = nullptr; ctx->cap_tile_stats = 0;
        const int64_t n = 2 * (int64_t)n_seg + 64;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_tile_stats, sizeof(float) * (size_t)n, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
        ctx->cap_tile_stats = n;
    }
    if (!ctx->ev_tile) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_tile, hipEventDisableTiming));
    TileArgs ta;
    ta.elems = a.elems; ta.ld = a.ld; ta.W = a.W; ta.perm = ctx->d_perm; ta.stats = probe ? ctx->h_tile_stats : nullptr;
    ta.dm_ref = ds->tile_dm_ref; ta.inv_k_yr = (float)(1.0 / ctx->consts.kepler_year_to_julian_day);
    ta.planet = kp; ta.pad = 0;
    hipLaunchKernelGGL(k_tile_sort, dim3((unsigned)n_seg), dim3(TILE_TPB), 0, st, ta);
    if (probe) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_tile, st));
        ctx->tile_pending = true; ctx->tile_pending_segs = n_seg; ctx->tile_probes += 1;
        if (ctx->tile_seq == 1) {
            // The FIRST probe of a (dataset, batch size) is read at once (one host wait, once per shape), so that the decision already holds for the
            // evaluation it was taken on: the same inputs then give the same bits on every call of a context's life — later probes (read one
            // evaluation late, without a wait) can only change the decision when the inputs have changed.
            int rcd = tile_settle(ctx, ds);
            if (rcd) return rcd;
            sort_now = ctx->tile_on;
        }
    }
    if (sort_now) { ctx->tile_sorted_launches += 1; a.perm = ctx->d_perm; }
    return OCTO_OK;
}

int drain_timing(octo_ctx* ctx) {
    for (size_t k = 0; k < ctx->ev_used; ++k) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_pool[k].second));
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[k].first, ctx->ev_pool[k].second));
        ctx->t_ms += ms;
        ctx->t_n += 1;
        if (ctx->t_samples.size() < (1u << 20)) ctx->t_samples.push_back(ms);
    }
    ctx->ev_used = 0;
    return OCTO_OK;
}

}  // namespace octo

extern "C" {

int32_t octo_consts_default(octo_consts* out) {
    if (!out) return OCTO_EINVAL;
    // PlanetOrbits.jl / Octofitter.jl values this build assumes unless the host overrides them.
    out->kepler_year_to_julian_day = 365.2568983840419;   // 2π√(au³/GM☉)/86400
    out->year2day_julian = 365.25;
    out->au2m = 1.495978707e11;
    out->sec2year_julian = 3.168808781402895e-8;
    out->pc2au = 206265.0;
    out->rad2as = 206265.0;
    out->mjup2msol = 0.0009545942339693249;
    return OCTO_OK;
}

int32_t octo_version(int32_t* major, int32_t* minor) {
    if (major) *major = OCTO_VERSION_MAJOR;
    if (minor) *minor = OCTO_VERSION_MINOR;
    return OCTO_OK;
}

int32_t octo_ctx_create(octo_ctx** out, int32_t device_id) {
    if (!out) return OCTO_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return OCTO_ENODEV;
    octo_ctx* ctx = new (std::nothrow) octo_ctx();
    if (!ctx) return OCTO_ENOMEM;
    ctx->device = device_id;
    octo_consts_default(&ctx->consts);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
            ctx->n_cus = prop.multiProcessorCount;
            // gfx950: 160 KB per CU, and one block may take all of it once the kernel's attribute has been raised
            ctx->max_lds = std::max<int64_t>((int64_t)prop.sharedMemPerBlock, (int64_t)prop.maxSharedMemoryPerMultiProcessor);
            if (ctx->max_lds <= 0) ctx->max_lds = 65536;
        }
    }
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return OCTO_EHIP;
    }
    if (hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return OCTO_EHIP;
    }
    {
        // sin/cos at the grid points k·SCT_STEP (exact products), k = −(SCT_HALF+SCT_PAD) … +(SCT_HALF+SCT_PAD)
        std::vector<double> tab(2 * SCT_N);
        for (int i = 0; i < SCT_N; ++i) {
            const double x = (double)(i - (SCT_HALF + SCT_PAD)) * SCT_STEP;
            tab[2 * i] = std::sin(x); tab[2 * i + 1] = std::cos(x);
        }
        if (hipMalloc((void**)&ctx->d_sctab, sizeof(double) * tab.size()) != hipSuccess ||
            hipMemcpy(ctx->d_sctab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipStreamDestroy(ctx->stream); (void)hipEventDestroy(ctx->ev_order);
            delete ctx;
            return OCTO_ENOMEM;
        }
    }
    if (hipMalloc((void**)&ctx->d_counters, sizeof(int32_t) * SMALL_W) != hipSuccess ||
        hipMemset(ctx->d_counters, 0, sizeof(int32_t) * SMALL_W) != hipSuccess) {
        (void)hipFree(ctx->d_sctab); (void)hipStreamDestroy(ctx->stream); (void)hipEventDestroy(ctx->ev_order);
        delete ctx;
        return OCTO_ENOMEM;
    }
    if (hipHostMalloc((void**)&ctx->h_flags, sizeof(uint64_t) * (SMALL_W + 32), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
        (void)hipFree(ctx->d_counters); (void)hipFree(ctx->d_sctab); (void)hipStreamDestroy(ctx->stream); (void)hipEventDestroy(ctx->ev_order);
        delete ctx;
        return OCTO_ENOMEM;
    }
    std::memset(ctx->h_flags, 0, sizeof(uint64_t) * (SMALL_W + 32));
    if (const char* ev = std::getenv("OCTO_SMALL_W")) { ctx->small_w = std::min(std::max(std::atoi(ev), 0), SMALL_W); ctx->small_w_model = 0; }      // (as octo_ctx_set_small_batch: the limit holds for the model callback too — ADVICE r5)
    if (const char* ev = std::getenv("OCTO_MAPPED_W")) ctx->mapped_w = std::min(std::max(std::atoi(ev), 0), SMALL_W);
    if (const char* ev = std::getenv("OCTO_FLAG_W")) ctx->flag_w = std::min(std::max(std::atoi(ev), 0), SMALL_W);
    ctx->env_small_blocks = env_int("OCTO_SMALL_BLOCKS"); ctx->env_small_min_span = env_int("OCTO_SMALL_MIN_SPAN");
    ctx->env_stage_bytes = env_int("OCTO_STAGE_BYTES"); ctx->env_chunk = env_int("OCTO_CHUNK"); ctx->env_rounds = env_int("OCTO_ROUNDS"); ctx->env_rv_cost = env_int("OCTO_RV_COST"); ctx->env_kind_all = env_int("OCTO_KIND_ALL"); ctx->env_mainp_tpb = env_int("OCTO_MAINP_TPB");
    if (const char* ev = std::getenv("OCTO_WIDE")) ctx->env_wide = std::atoi(ev);
    if (const char* ev = std::getenv("OCTO_FIN_FUSED")) ctx->env_no_fin_fused = std::atoi(ev) ? 0 : 1;
    if (const char* ev = std::getenv("OCTO_WARM")) { ctx->env_warm = std::atoi(ev); ctx->opt_warm = ctx->env_warm ? 1 : 0; }
    if (const char* ev = std::getenv("OCTO_TILE_SORT")) ctx->tile_mode = std::min(std::max(std::atoi(ev), 0), 2);      // experiments: the default of OCTO_OPT_TILE_SORT
    if (const char* ev = std::getenv("OCTO_TILE_MIN_W")) ctx->tile_min_w = std::max<int64_t>(std::atoll(ev), 64);
    *out = ctx;
    return OCTO_OK;
}

int32_t octo_ctx_destroy(octo_ctx* ctx) {
    if (!ctx) return OCTO_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) (void)octo_comm_destroy(ctx);
    (void)hipDeviceSynchronize();      // evaluations may have been enqueued on caller-owned streams
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    free_retired(ctx);
    for (auto& t : ctx->tables) free_table(t);
    for (auto& e : ctx->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    (void)hipFree(ctx->d_wc); (void)hipFree(ctx->d_valid); (void)hipFree(ctx->d_partials); (void)hipFree(ctx->d_marg); (void)hipFree(ctx->d_sctab); (void)hipFree(ctx->d_extra); (void)hipFree(ctx->d_counters);
    (void)hipHostFree(ctx->h_in); (void)hipHostFree(ctx->h_out); (void)hipHostFree(ctx->h_flags);
    (void)hipFree(ctx->d_in); (void)hipFree(ctx->d_out);
    (void)hipFree(ctx->d_perm); (void)hipHostFree(ctx->h_tile_stats);
    for (void* q : ctx->retired_host) (void)hipHostFree(q);
    if (ctx->ev_tile) (void)hipEventDestroy(ctx->ev_tile);
    delete ctx;
    return OCTO_OK;
}

int32_t octo_consts_set(octo_ctx* ctx, const octo_consts* c) {
    if (!ctx || !c) return OCTO_EINVAL;
    const double* v = (const double*)c;
    for (int k = 0; k < 7; ++k)
        if (!(v[k] > 0.0) || !std::isfinite(v[k])) return fail(ctx, OCTO_EINVAL, "octo_consts_set: constants must be finite and positive");
    ctx->consts = *c;
    return OCTO_OK;
}

// Test hook (not part of the C ABI of include/octofitter_hip.h; tests/test_sweeps_gpu.py binds it by name): fill the LDS of every CU
// with the bit pattern of `value`. LDS keeps what the last block left there, and a kernel that reads a word it has not written
// usually finds zeros or stale finite numbers — which is how an uninitialised read in this round's k_small<MODEL> passed every
// fixture. With NaN (or a huge number) in every word such a read changes the result, deterministically.
static __global__ __launch_bounds__(256) void k_poison_lds(double value, int n) {
    extern __shared__ __attribute__((aligned(16))) double pl[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) pl[i] = value;
    __syncthreads();
    if (pl[(threadIdx.x * 37) % n] != value && value == value) __builtin_trap();      // keep the stores
    // stay resident for a few microseconds so that the launch spreads over every CU instead of reusing the first ones
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}      // 100 MHz ticks: 20 µs
}
int32_t octo_debug_poison_lds(octo_ctx* ctx, double value) {
    if (!ctx) return OCTO_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int bytes = 64 * 1024;      // two such blocks fill most of a CU's 160 KB, each starting at the offsets a small kernel's blocks get
    hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)(ctx->n_cus * 2)), dim3(256), (size_t)bytes, ctx->stream, value, bytes / 8);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return OCTO_OK;
}

#ifdef OCTO_SMALL_TRACE
// development build only: cycle stamps of the last k_small launch's walker-0 finishing block (tools/small_trace.py)
const uint64_t* octo_debug_small_trace(octo_ctx* ctx) { return ctx ? ctx->h_flags + SMALL_W : nullptr; }
#endif

int32_t octo_ctx_set_small_batch(octo_ctx* ctx, int32_t max_walkers) {
    if (!ctx || max_walkers < 0) return OCTO_EINVAL;
    ctx->small_w = std::min<int>(max_walkers, SMALL_W);
    ctx->small_w_model = 0;      // an explicit limit holds for the model callback too
    return OCTO_OK;
}

const char* octo_last_error(const octo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t octo_ctx_set_option(octo_ctx* ctx, int32_t option, int64_t value) {
    if (!ctx) return OCTO_EINVAL;
    switch (option) {
        case OCTO_OPT_BATCH_INVARIANT: if (value != 0 && value != 1) break; ctx->opt_invariant = (int)value; return OCTO_OK;
        case OCTO_OPT_WARM_START: if (value != 0 && value != 1) break; ctx->opt_warm = (int)value; return OCTO_OK;
        case OCTO_OPT_TILE_SORT: if (value < 0 || value > 2) break; ctx->tile_mode = (int)value; ctx->tile_seq = 0; ctx->tile_on = false; ctx->tile_pending = false; return OCTO_OK;
        case OCTO_OPT_TILE_MIN_WALKERS: if (value < 64) break; ctx->tile_min_w = value; return OCTO_OK;
        default: return fail(ctx, OCTO_EINVAL, "octo_ctx_set_option: unknown option " + std::to_string(option));
    }
    return fail(ctx, OCTO_EINVAL, "octo_ctx_set_option: value " + std::to_string(value) + " out of range for option " + std::to_string(option));
}

int32_t octo_ctx_get_option(const octo_ctx* ctx, int32_t option, int64_t* value_out) {
    if (!ctx || !value_out) return OCTO_EINVAL;
    switch (option) {
        case OCTO_OPT_BATCH_INVARIANT: *value_out = ctx->opt_invariant; return OCTO_OK;
        case OCTO_OPT_WARM_START: *value_out = ctx->opt_warm; return OCTO_OK;
        case OCTO_OPT_TILE_SORT: *value_out = ctx->tile_mode; return OCTO_OK;
        case OCTO_OPT_TILE_MIN_WALKERS: *value_out = ctx->tile_min_w; return OCTO_OK;
        default: return OCTO_EINVAL;
    }
}

int32_t octo_dataset_create(octo_ctx* ctx, const octo_obs_desc* obs, int32_t n_obs,
                            const octo_planet_desc* planets, int32_t n_planets, octo_dataset** out) {
    if (!ctx || !out || n_obs < 0 || (n_obs > 0 && !obs) || !planets) return fail(ctx, OCTO_EINVAL, "octo_dataset_create: null argument");
    *out = nullptr;
    if (n_planets < 1) return fail(ctx, OCTO_EINVAL, "octo_dataset_create: a dataset needs at least one planet");
    if (n_planets > MAXP) return fail(ctx, OCTO_ENOTSUP, "octo_dataset_create: 1.." + std::to_string(MAXP) + " planets supported");
    for (int p = 0; p < n_planets; ++p)
        if (planets[p].orbit_kind != OCTO_ORBIT_VISUAL_KEP && planets[p].orbit_kind != OCTO_ORBIT_RADVEL &&
            planets[p].orbit_kind != OCTO_ORBIT_THIELE_INNES && planets[p].orbit_kind != OCTO_ORBIT_KEP)
            return fail(ctx, OCTO_EINVAL, "octo_dataset_create: unknown orbit kind");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    octo_dataset* ds = new (std::nothrow) octo_dataset();
    if (!ds) return fail(ctx, OCTO_ENOMEM, "octo_dataset_create: host allocation failed");
    ds->device = ctx->device; ds->n_obs = n_obs; ds->n_planets = n_planets;
    ds->serial = g_dataset_serial.fetch_add(1);
    for (int p = 0; p < n_planets; ++p) ds->planets[p] = planets[p];
    auto bail = [&](int code, const std::string& msg) { octo_dataset_destroy(ds); return fail(ctx, code, msg); };
    ds->h_obs.resize(n_obs); ds->h_rowconst_pre.resize(n_obs); ds->h_rowconst_raw.resize(n_obs); ds->h_step.resize(n_obs);
    for (int o = 0; o < n_obs; ++o) {
        const octo_obs_desc& d = obs[o];
        if (d.kind < 0 || d.kind >= OCTO_N_KINDS) return bail(OCTO_EINVAL, "octo_dataset_create: unknown observation kind");
        if (d.n_epochs < 0 || d.n_epochs > 0x7fffffff) return bail(OCTO_EINVAL, "octo_dataset_create: bad n_epochs");
        if (d.kind == OCTO_HGCA) {
            // rows {epoch, axis, mission}; `extra` = the 15 catalogue numbers; evaluated by k_hgca
            if (d.n_epochs > 0 && (!d.epoch || !d.y1 || !d.y2)) return bail(OCTO_EINVAL, "octo_dataset_create: missing column");
            if (!d.extra || d.n_extra != OCTO_HGCA_N_EXTRA) return bail(OCTO_EINVAL, "octo_dataset_create: OCTO_HGCA needs extra[15]");
            for (int p = 0; p < n_planets; ++p)       // mass * mjup2msol of every Visual planet is read (hgca.jl:279-290)
                if (planets[p].orbit_kind != OCTO_ORBIT_RADVEL && planets[p].orbit_kind != OCTO_ORBIT_KEP && !planets[p].has_mass)
                    return bail(OCTO_EINVAL, "octo_dataset_create: OCTO_HGCA needs a mass on every Visual{KepOrbit} / ThieleInnesOrbit planet");
            for (int k = 0; k < 3; ++k)
                if (!(d.extra[5 * k + 2] > 0.0) || !(d.extra[5 * k + 3] > 0.0) || !(std::fabs(d.extra[5 * k + 4]) < 1.0))
                    return bail(OCTO_EINVAL, "octo_dataset_create: OCTO_HGCA covariance is not positive definite");
            const int64_t n = d.n_epochs;
            std::vector<double> raw((size_t)std::max<int64_t>(n, 1) * ROW_STRIDE, 0.0);
            for (int64_t r = 0; r < n; ++r) {
                const int ax = (int)d.y1[r], ms = (int)d.y2[r];
                if ((ax != OCTO_HGCA_RA && ax != OCTO_HGCA_DEC) || (ms != OCTO_HGCA_HIP && ms != OCTO_HGCA_GAIA))
                    return bail(OCTO_EINVAL, "octo_dataset_create: OCTO_HGCA rows need y1 in {RA, DEC}, y2 in {HIP, GAIA}");
                if (!std::isfinite(d.epoch[r])) return bail(OCTO_EINVAL, "octo_dataset_create: OCTO_HGCA epochs must be finite");
                raw[(size_t)r * ROW_STRIDE] = d.epoch[r]; raw[(size_t)r * ROW_STRIDE + 1] = ax; raw[(size_t)r * ROW_STRIDE + 2] = ms;
            }
            DevObs& h = ds->h_obs[o];
            h.kind = d.kind; h.planet = -1; h.has_cor = 0; h.dm_max = 0.0f; h.n = n;
            double *dr = nullptr, *dx = nullptr;
            if (hipMalloc((void**)&dr, sizeof(double) * raw.size()) != hipSuccess) return bail(OCTO_ENOMEM, "octo_dataset_create: hipMalloc failed");
            ds->d_bufs.push_back(dr);
            if (hipMalloc((void**)&dx, sizeof(double) * OCTO_HGCA_N_EXTRA) != hipSuccess) return bail(OCTO_ENOMEM, "octo_dataset_create: hipMalloc failed");
            ds->d_bufs.push_back(dx);
            if (hipMemcpy(dr, raw.data(), sizeof(double) * raw.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dx, d.extra, sizeof(double) * OCTO_HGCA_N_EXTRA, hipMemcpyHostToDevice) != hipSuccess)
                return bail(OCTO_EHIP, "octo_dataset_create: upload failed");
            h.raw = dr; h.pre = dx;
            ds->n_hgca += 1;
            ds->kind_mask |= KM_HGCA;
            continue;
        }
        const bool oneil = d.kind == OCTO_ONEIL_RADEC || d.kind == OCTO_ONEIL_SEPPA;
        const bool astrom = d.kind == OCTO_ASTROM_RADEC || d.kind == OCTO_ASTROM_SEPPA || oneil;
        const bool planet_obs = astrom || d.kind == OCTO_RV_REL;
        if (planet_obs && (d.planet < 0 || d.planet >= n_planets)) return bail(OCTO_EINVAL, "octo_dataset_create: planet index out of range");
        if (d.n_epochs > 0 && (!d.epoch || !d.y1 || !d.s1 || (astrom && (!d.y2 || !d.s2))))
            return bail(OCTO_EINVAL, "octo_dataset_create: missing column");
        if (astrom && (planets[d.planet].orbit_kind == OCTO_ORBIT_RADVEL || planets[d.planet].orbit_kind == OCTO_ORBIT_KEP))
            return bail(OCTO_EINVAL, "octo_dataset_create: astrometry needs a Visual{KepOrbit} or ThieleInnesOrbit planet");
        if (!astrom)       // RV of a ThieleInnesOrbit (PlanetOrbits derives i, ω from A, B, F, G for it) is not on this path
            for (int p = 0; p < n_planets; ++p)
                if (planets[p].orbit_kind == OCTO_ORBIT_THIELE_INNES && (!planet_obs || p == d.planet || planets[p].has_mass))
                    return bail(OCTO_ENOTSUP, "octo_dataset_create: RV tables with a ThieleInnesOrbit planet are not supported");
        if (!planet_obs)   // every planet contributes to absolute RV and requires a mass (rv-absolute.jl:146-155)
            for (int p = 0; p < n_planets; ++p)
                if (!planets[p].has_mass) return bail(OCTO_EINVAL, "octo_dataset_create: absolute RV needs a mass on every planet");
        switch (d.kind) {
            case OCTO_ASTROM_RADEC: ds->kind_mask |= KM_RADEC; break;
            case OCTO_ASTROM_SEPPA: ds->kind_mask |= KM_SEPPA; break;
            case OCTO_RV_ABS: ds->kind_mask |= KM_RVABS; break;
            case OCTO_RV_ABS_MARG: ds->kind_mask |= KM_MARG; break;
            case OCTO_RV_REL: ds->kind_mask |= KM_RVREL; break;
            case OCTO_ONEIL_RADEC: ds->kind_mask |= KM_RADEC | KM_ONEIL; break;
            default: ds->kind_mask |= KM_SEPPA | KM_ONEIL; break;
        }
        if (astrom && d.cor) ds->kind_mask |= KM_COR;
        const int64_t n = d.n_epochs;
        // RV kinds: `extra` is the trend basis column, one value per row (OCTO_NU_RV_TREND)
        const bool has_basis = !astrom && d.extra != nullptr && d.n_extra > 0;
        if (has_basis && d.n_extra != n) return bail(OCTO_EINVAL, "octo_dataset_create: an RV table's trend basis column (extra) needs n_extra == n_epochs");
        if (astrom && d.extra != nullptr && d.n_extra > 0) return bail(OCTO_EINVAL, "octo_dataset_create: astrometry tables take no `extra`");
        std::vector<double> raw((size_t)n * ROW_STRIDE, 0.0), pre((size_t)n * ROW_STRIDE, 0.0);
        ds->h_rowconst_pre[o].resize(n); ds->h_rowconst_raw[o].resize(n); ds->h_step[o].resize(n);
        for (int64_t r = 0; r < n; ++r) {
            double* a = &raw[(size_t)r * ROW_STRIDE];
            double* b = &pre[(size_t)r * ROW_STRIDE];
            a[0] = b[0] = d.epoch[r]; a[1] = b[1] = d.y1[r];
            // a non-finite value or σ <= 0 would be baked into 1/σ² and the log-constants and silently turn every walker into -Inf
            if (!std::isfinite(d.epoch[r]) || !std::isfinite(d.y1[r]) || !(d.s1[r] > 0.0) || !std::isfinite(d.s1[r]) ||
                (astrom && (!std::isfinite(d.y2[r]) || !(d.s2[r] > 0.0) || !std::isfinite(d.s2[r]))))
                return bail(OCTO_EINVAL, "octo_dataset_create: table " + std::to_string(o) + " row " + std::to_string(r) +
                                         ": epochs and measurements must be finite and uncertainties finite and > 0");
            if (astrom) {
                const double s1 = d.s1[r], s2 = d.s2[r], c = d.cor ? d.cor[r] : 0.0;
                if (d.cor && !(std::fabs(c) <= 1.0 - 1e-5))   // relative-astrometry.jl:70-72
                    return bail(OCTO_EINVAL, "octo_dataset_create: correlation values may not be well-specified");
                a[2] = b[2] = d.y2[r]; a[3] = s1; a[4] = s2; a[5] = c;
                const double omc = 1.0 - c * c;
                b[3] = 1.0 / (s1 * s1 * omc); b[4] = 1.0 / (s2 * s2 * omc); b[5] = -c / (s1 * s2 * omc);
                ds->h_rowconst_pre[o][r] = -LOG2PI - 0.5 * std::log(s1 * s1 * s2 * s2 * omc);
                ds->h_rowconst_raw[o][r] = -LOG2PI;
            } else {
                const double s = d.s1[r];
                a[2] = s; b[2] = 1.0 / (s * s);
                if (has_basis) {
                    if (!std::isfinite(d.extra[r])) return bail(OCTO_EINVAL, "octo_dataset_create: table " + std::to_string(o) + " row " + std::to_string(r) + ": trend basis must be finite");
                    a[3] = d.extra[r];      // read by the nuisance path only: without nuisances the trend coefficient is 0
                }
                if (d.kind == OCTO_RV_ABS_MARG) {
                    ds->h_rowconst_pre[o][r] = -std::log(TWO_PI * s * s);   // −log(2π var), rv-absolute-margin.jl:179
                    ds->h_rowconst_raw[o][r] = 0.0;
                } else {
                    ds->h_rowconst_pre[o][r] = -0.5 * (LOG2PI + std::log(s * s));
                    ds->h_rowconst_raw[o][r] = -0.5 * LOG2PI;
                }
            }
        }
        // slot 6 of every record: 2π·(t − t of the previous row), 0 in row 0 — the mean-anomaly step per unit mean motion that k_main's
        // warm-started row loop multiplies by 1/P (octo_device.h: KWarm). Slot 7: the row's warm KEY — |that step|, or +Inf where the row must
        // start cold whatever the wave's bound is: row 0, every WARM_RESTART-th row (the chain of warm rows never sees E or M as numbers, so its
        // rounding accumulates until the next cold row: this bounds the chain whatever chunk the planner picks), a non-finite step.
        double dm_max = 0.0;
        std::vector<double> steps;
        steps.reserve((size_t)std::max<int64_t>(n, 1));
        for (int64_t r = 0; r < n; ++r) {
            const double dm = r > 0 ? TWO_PI * (d.epoch[r] - d.epoch[r - 1]) : 0.0;
            double key = std::fabs(dm);
            if (r > 0) { dm_max = std::max(dm_max, key); steps.push_back(key); }
            ds->h_step[o][r] = std::isfinite(key) ? key : INFINITY;
            if (r == 0 || (r % WARM_RESTART) == 0 || !std::isfinite(key)) key = INFINITY;
            raw[(size_t)r * ROW_STRIDE + 6] = pre[(size_t)r * ROW_STRIDE + 6] = dm;
            raw[(size_t)r * ROW_STRIDE + 7] = pre[(size_t)r * ROW_STRIDE + 7] = key;
        }
        DevObs& h = ds->h_obs[o];
        h.kind = d.kind; h.planet = planet_obs ? d.planet : -1; h.has_cor = d.cor ? 1 : 0; h.dm_max = ctx->env_warm ? (float)(dm_max * 1.000001) : 0.0f; h.n = n;
        // The ladder of candidate bounds (DevObs::dm_ladder), preferred first: the 97 % quantile of the steps (the largest 3 % of a table's gaps
        // are not worth a lower bound on 1/D for every other row: a cold row costs a third more than a warm one), then smaller quantiles for
        // waves whose periods are too short for it. Each rounded UP to a float; equal neighbours collapse; 0 = no entry.
        for (int k = 0; k < WARM_LADDER; ++k) h.dm_ladder[k] = 0.0f;
        if (ctx->env_warm && !steps.empty()) {
            std::sort(steps.begin(), steps.end());
            static const double q[WARM_LADDER] = {0.97, 0.90, 0.75, 0.50, 0.25, 0.10, 0.03, 0.01};
            int m = 0;
            for (int k = 0; k < WARM_LADDER; ++k) {
                const size_t idx = (size_t)std::min<double>((double)steps.size() - 1.0, std::ceil(q[k] * (double)steps.size()) - 1.0 < 0.0 ? 0.0 : std::ceil(q[k] * (double)steps.size()) - 1.0);
                const double v = steps[idx];
                if (!(v > 0.0) || !std::isfinite(v)) continue;
                float f = (float)v;
                if ((double)f < v) f = std::nextafterf(f, INFINITY);
                if (m > 0 && h.dm_ladder[m - 1] == f) continue;
                h.dm_ladder[m++] = f;
            }
        }
        h.raw = h.pre = nullptr;
        if (n > 0) {
            double *dr = nullptr, *dp = nullptr;
            const size_t bytes = sizeof(double) * (size_t)n * ROW_STRIDE;
            if (hipMalloc((void**)&dr, bytes) != hipSuccess) return bail(OCTO_ENOMEM, "octo_dataset_create: hipMalloc failed");
            ds->d_bufs.push_back(dr);
            if (hipMalloc((void**)&dp, bytes) != hipSuccess) return bail(OCTO_ENOMEM, "octo_dataset_create: hipMalloc failed");
            ds->d_bufs.push_back(dp);
            if (hipMemcpy(dr, raw.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dp, pre.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
                return bail(OCTO_EHIP, "octo_dataset_create: upload failed");
            h.raw = dr; h.pre = dp;
        }
        ds->n_rows += n;
    }
    {   // the walker-tile sort's reference step: the preferred rung of the largest table that has a ladder
        int64_t best = 0;
        for (int o = 0; o < n_obs; ++o) {
            const DevObs& h = ds->h_obs[o];
            if (h.kind == OCTO_HGCA || !(h.dm_ladder[0] > 0.0f)) continue;
            ds->tile_rows += h.n;
            if (h.n > best) { best = h.n; ds->tile_dm_ref = h.dm_ladder[0]; }
        }
    }
    if (n_obs > 0) {
        if (hipMalloc((void**)&ds->d_obs, sizeof(DevObs) * n_obs) != hipSuccess) return bail(OCTO_ENOMEM, "octo_dataset_create: hipMalloc failed");
        if (hipMemcpy(ds->d_obs, ds->h_obs.data(), sizeof(DevObs) * n_obs, hipMemcpyHostToDevice) != hipSuccess)
            return bail(OCTO_EHIP, "octo_dataset_create: upload failed");
    }
    *out = ds;
    return OCTO_OK;
}

int32_t octo_dataset_destroy(octo_dataset* ds) {
    if (!ds) return OCTO_OK;
    (void)hipSetDevice(ds->device);
    for (double* p : ds->d_bufs) (void)hipFree(p);
    (void)hipFree(ds->d_obs);
    delete ds;
    return OCTO_OK;
}

int64_t octo_dataset_n_rows(const octo_dataset* ds) { return ds ? ds->n_rows : -1; }

// The evaluation behind octo_eval_device and octo_model_logpost_device. `sm` non-null: the fused small-batch launch with the
// standard parameterisation inside (θ_t in, log-posterior out; d_elems / outputs unused), `grad` / `nuis` given by the caller.
// `mt` non-null: the model's tail for k_finish (octo_model_logpost_device on the throughput kernels); *tail_applied then says whether a k_finish /
// k_finishp launch carried it (a batch that k_small took has no k_finish: the caller launches k_model_bwd). Both are explicit arguments — rounds
// 3-4 passed them through mutable context state (mt_req / a flag the caller had to reset), which a new early-return path could have left stale
// (ADVICE r4). ctx->mt_applied remains only as the launch code's way to report back, reset here on every call.
static int eval_impl(octo_ctx* ctx, const octo_dataset* ds, const double* d_elems, const double* d_nuis, int64_t ld, int64_t W,
                     double* d_ll, double* d_g_elems, double* d_g_nuis, hipStream_t st, const SmallModel* sm, bool grad, bool nuis,
                     const octo_ctx::ModelTail* mt = nullptr, bool* tail_applied = nullptr) {
    ctx->mt_applied = false;
    if (tail_applied) *tail_applied = false;
    if (ctx->timing_every > 0 && ctx->ev_used >= 4096) { int rc = drain_timing(ctx); if (rc) return rc; }
    const int64_t ldw = (W + WAVE - 1) / WAVE * WAVE;
    if (ldw > ctx->cap_w) {
        int64_t need = ldw;
        // both buffers share the walker capacity
        if (ctx->d_wc) { ctx->retired.push_back(ctx->d_wc); ctx->d_wc = nullptr; }
        if (ctx->d_valid) { ctx->retired.push_back(ctx->d_valid); ctx->d_valid = nullptr; }
        need += need / 2;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_wc, sizeof(double) * (size_t)need * NWC * MAXP));
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_valid, sizeof(int32_t) * (size_t)need * MAXP));
        ctx->cap_w = need;
    }
    EvalArgs a;
    std::memset(&a, 0, sizeof(a));
    a.obs = ds->d_obs;
    a.n_obs = ds->n_obs; a.n_planets = ds->n_planets;
    for (int p = 0; p < ds->n_planets; ++p) { a.orbit_kind[p] = ds->planets[p].orbit_kind; a.has_mass[p] = ds->planets[p].has_mass; }
    a.elems = d_elems; a.nuis = d_nuis; a.ld = ld; a.W = W;
    a.warm = (ctx->opt_warm && !ctx->opt_invariant) ? 1 : 0;
    a.ws_in = a.ws_out = 1;
    if (ctx->stage_ws_in > 0) { a.ws_in = ctx->stage_ws_in; a.ws_out = ctx->stage_ws_out; }      // octo_eval's walker-major staging (k_small only)
    a.wc = ctx->d_wc; a.valid = ctx->d_valid; a.ldw = ctx->cap_w; a.sctab = ctx->d_sctab;
    a.ll_out = d_ll; a.g_elems = d_g_elems; a.g_nuis = d_g_nuis;
    if (mt) {
        a.mt_Jc = mt->Jc; a.mt_gtp = mt->gtp; a.mt_esrc = mt->esrc; a.mt_nsrc = mt->nsrc; a.mt_glp = mt->glp; a.mt_lpp = mt->lpp; a.mt_lp = mt->lp; a.mt_grad = mt->grad;
        a.mt_ld = mt->ld; a.mt_ldo = mt->ldo; a.mt_D = mt->D; a.mt_n_nu = mt->n_nu;
    }
    a.c = dev_consts(ctx->consts);
    int rc;
    switch (ds->n_planets) {
        case 1: rc = dispatch1<1>(ctx, ds, a, grad, nuis, sm, st); break;
        case 2: rc = dispatch1<2>(ctx, ds, a, grad, nuis, sm, st); break;
        case 3: rc = dispatch1<3>(ctx, ds, a, grad, nuis, sm, st); break;
        case 4: rc = dispatch1<4>(ctx, ds, a, grad, nuis, sm, st); break;
        default: rc = dispatch_many(ctx, ds, a, grad, nuis, sm, st); break;
    }
    if (tail_applied) *tail_applied = rc == OCTO_OK && mt != nullptr && ctx->mt_applied;
    return rc;
}

int32_t octo_eval_device(octo_ctx* ctx, const octo_dataset* cds, const double* d_elems, const double* d_nuis,
                         int64_t ld, int64_t W, double* d_ll, double* d_g_elems, double* d_g_nuis, void* hip_stream) {
    if (!ctx || !cds || !d_elems || !d_ll) return fail(ctx, OCTO_EINVAL, "octo_eval_device: null argument");
    if (W < 0 || (ld < W && ctx->stage_ws_in == 0)) return fail(ctx, OCTO_EINVAL, "octo_eval_device: need 0 <= W <= ld");
    if (d_g_nuis && (!d_g_elems || !d_nuis)) return fail(ctx, OCTO_EINVAL, "octo_eval_device: g_nuis needs g_elems and nuis");
    if (d_g_elems && d_nuis && !d_g_nuis) return fail(ctx, OCTO_EINVAL, "octo_eval_device: nuis given with g_elems but no g_nuis");
    if (cds->device != ctx->device) return fail(ctx, OCTO_EINVAL, "octo_eval_device: dataset lives on another device");
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    { int rcs = use_stream(ctx, hip_stream, &st); if (rcs) return rcs; }
    return eval_impl(ctx, cds, d_elems, d_nuis, ld, W, d_ll, d_g_elems, d_g_nuis, st, nullptr, d_g_elems != nullptr, d_nuis != nullptr);
}

int32_t octo_sync(octo_ctx* ctx) {
    if (!ctx) return OCTO_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->has_last || ctx->last_stream == ctx->stream) free_retired(ctx);      // a caller-owned stream may still be running
    return OCTO_OK;
}

static int32_t eval_begin_impl(octo_ctx* ctx, const octo_dataset* ds, const double* elems, const double* nuis, int64_t ld, int64_t W,
                               double* ll_out, double* g_elems, double* g_nuis) {
    if (!ctx || !ds || !elems || !ll_out) return fail(ctx, OCTO_EINVAL, "octo_eval: null argument");
    if (W < 0 || ld < W) return fail(ctx, OCTO_EINVAL, "octo_eval: need 0 <= W <= ld");
    if (ctx->pending.active) return fail(ctx, OCTO_EINVAL, "octo_eval_begin: the previous octo_eval_begin of this context has not been ended");
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n_el = ds->n_planets * OCTO_N_EL, n_nu = ds->n_obs * OCTO_N_NUIS;
    const int64_t ldd = (W + 63) / 64 * 64;
    const int64_t n_in = (int64_t)(n_el + (nuis ? n_nu : 0)) * ldd;
    const int64_t n_out = (int64_t)(1 + (g_elems ? n_el : 0) + (g_nuis ? n_nu : 0)) * ldd;
    hipStream_t st;
    { int rcs = use_stream(ctx, OCTO_STREAM_CTX, &st); if (rcs) return rcs; }
    octo_ctx::Pending& pd = ctx->pending;
    pd = octo_ctx::Pending();
    pd.W = W; pd.ld = ld; pd.ldd = ldd; pd.ll = ll_out; pd.g_elems = g_elems; pd.g_nuis = g_nuis;
    pd.n_el_out = g_elems ? n_el : 0; pd.n_nu_out = g_nuis ? n_nu : 0;
    if (small_eligible(ctx, ds, W) && W <= ctx->mapped_w && (ds->n_hgca == 0 || hgca_in_small(W))) {      // (k_hgca, lane = walker, would fetch every input over PCIe once per direction)
        // A handful of parameter sets (a sampler's one θ per call): no copy engine at all. k_small reads the inputs from, and
        // writes the results to, mapped pinned host memory — walker-major: [elems | nuis] of one walker contiguous (one PCIe read
        // per block), [ll | g_elems | g_nuis] likewise on the way back; completion through per-walker flags instead of a stream sync.
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_eval: pinned staging allocation failed");
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        pd.staged = true;
        const int nn = nuis ? n_nu : 0;
        const int64_t ws_in = n_el + nn, ws_out = 1 + pd.n_el_out + pd.n_nu_out;
        ctx->inl.n = 0;
        if (W == 1 && ws_in <= SMALL_INL) {
            // one parameter set: inside the kernel arguments (device memory) instead of the mapped buffer (a PCIe read at kernel start)
            ctx->inl.n = (int32_t)ws_in; ctx->inl.pad = 0;
            for (int r = 0; r < n_el; ++r) ctx->inl.v[r] = elems[(size_t)r * ld];
            for (int r = 0; r < nn; ++r) { ctx->inl.v[n_el + r] = nuis[(size_t)r * ld]; if (!std::isfinite(nuis[(size_t)r * ld])) ctx->inl.pad = 1; }
        } else {
            for (int64_t w = 0; w < W; ++w) {
                double* dst = ctx->h_in + w * ws_in;
                for (int r = 0; r < n_el; ++r) dst[r] = elems[(size_t)r * ld + w];
                for (int r = 0; r < nn; ++r) dst[n_el + r] = nuis[(size_t)r * ld + w];
            }
        }
        pd.walker_major = true; pd.ws_out = ws_out;
        ctx->stage_ws_in = ws_in; ctx->stage_ws_out = ws_out;
        ctx->flag_request = W <= ctx->flag_w; ctx->flag_armed = false;
        int rcz = octo_eval_device(ctx, ds, m_in, nuis ? m_in + n_el : nullptr, 1, W, m_out, g_elems ? m_out + 1 : nullptr,
                                   g_nuis ? m_out + 1 + pd.n_el_out : nullptr, st);
        ctx->flag_request = false; ctx->stage_ws_in = ctx->stage_ws_out = 0; ctx->inl.n = 0;
        if (rcz) return rcz;
        pd.active = true;
        return OCTO_OK;
    }
    int rc = grow(ctx, ctx->d_in, ctx->cap_in, n_in);
    if (rc) return rc;
    rc = grow(ctx, ctx->d_out, ctx->cap_out, n_out);
    if (rc) return rc;
    if (small_eligible(ctx, ds, W)) {
        // k_small beyond the mapped-input range: rows into the mapped buffer as they are, k_stage_in transposes them into device
        // memory, results come back walker-major through the mapped buffer + flags (no copy engine, no stream synchronisation)
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_eval: pinned staging allocation failed");
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        const int nn = nuis ? n_nu : 0;
        const int n_rows_in = n_el + nn;
        const int64_t ws_out = 1 + pd.n_el_out + pd.n_nu_out;
        for (int r = 0; r < n_el; ++r) std::memcpy(ctx->h_in + (size_t)r * ldd, elems + (size_t)r * ld, sizeof(double) * W);
        for (int r = 0; r < nn; ++r) std::memcpy(ctx->h_in + (size_t)(n_el + r) * ldd, nuis + (size_t)r * ld, sizeof(double) * W);
        hipLaunchKernelGGL(k_stage_in, dim3((unsigned)((W * n_rows_in + 255) / 256)), dim3(256), 0, st, m_in, ldd, W, n_rows_in, ctx->d_in);
        pd.staged = true; pd.walker_major = true; pd.ws_out = ws_out;
        ctx->stage_ws_in = n_rows_in; ctx->stage_ws_out = ws_out;
        ctx->flag_request = false; ctx->flag_armed = false;      // completion by stream synchronisation: per-walker system-scope releases cost more than they save beyond ~100 walkers (W = 512: 38 µs with flags, 30 without)
        int rcz = octo_eval_device(ctx, ds, ctx->d_in, nuis ? ctx->d_in + n_el : nullptr, 1, W, m_out, g_elems ? m_out + 1 : nullptr,
                                   g_nuis ? m_out + 1 + pd.n_el_out : nullptr, st);
        ctx->flag_request = false; ctx->stage_ws_in = ctx->stage_ws_out = 0;
        if (rcz) return rcz;
        pd.active = true;
        return OCTO_OK;
    }
    double* d_nuis = nuis ? ctx->d_in + (int64_t)n_el * ldd : nullptr;
    double* d_ll = ctx->d_out;
    double* d_ge = g_elems ? ctx->d_out + ldd : nullptr;
    double* d_gn = g_nuis ? ctx->d_out + (int64_t)(1 + (g_elems ? n_el : 0)) * ldd : nullptr;
    if ((n_in + n_out) * (int64_t)sizeof(double) <= stage_bytes(ctx)) {
        // Mid-size batches of the throughput kernels (10³ walkers): the rows are packed into the mapped pinned buffer, a copy KERNEL
        // brings them into device memory (k_main blocks must not fetch their walkers' nuisances over PCIe one by one) and k_finish
        // writes the results straight into the mapped buffer — coalesced rows. Pageable 2-D copies cost ~8 µs apiece, and even
        // two pinned copy-engine transfers cost 11-14 µs more per call than this (W = 1024, 300 epochs: 47 -> 35 µs).
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_eval: pinned staging allocation failed");
        for (int r = 0; r < n_el; ++r) std::memcpy(ctx->h_in + (size_t)r * ldd, elems + (size_t)r * ld, sizeof(double) * W);
        if (nuis) for (int r = 0; r < n_nu; ++r) std::memcpy(ctx->h_in + (size_t)(n_el + r) * ldd, nuis + (size_t)r * ld, sizeof(double) * W);
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        hipLaunchKernelGGL(k_copy_in, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, m_in, ctx->d_in, n_in);
        rc = octo_eval_device(ctx, ds, ctx->d_in, d_nuis, ldd, W, m_out, g_elems ? m_out + ldd : nullptr,
                              g_nuis ? m_out + (int64_t)(1 + (g_elems ? n_el : 0)) * ldd : nullptr, st);
        if (rc) return rc;
        pd.staged = true; pd.walker_major = false;
        pd.o_ge = ldd; pd.o_gn = (int64_t)(1 + pd.n_el_out) * ldd;
        pd.active = true;
        return OCTO_OK;
    }
    {
        // Buffers the caller registered (octo_host_register): no copy engine and no staging copy. One copy kernel brings the inputs
        // into device memory over PCIe (coalesced; k_main blocks must not fetch nuisances from the host one by one), and k_finish
        // writes ll and the gradients straight into the caller's arrays. Everything keeps the caller's leading dimension.
        const size_t sp_el = sizeof(double) * ((size_t)(n_el - 1) * ld + W), sp_nu = sizeof(double) * ((size_t)(n_nu > 0 ? n_nu - 1 : 0) * ld + W);
        double* m_el = (double*)mapped_range(ctx->device, elems, sp_el);
        double* m_nu = nuis ? (double*)mapped_range(ctx->device, nuis, sp_nu) : nullptr;
        double* m_ll = (double*)mapped_range(ctx->device, ll_out, sizeof(double) * W);
        double* m_ge = g_elems ? (double*)mapped_range(ctx->device, g_elems, sp_el) : nullptr;
        double* m_gn = g_nuis ? (double*)mapped_range(ctx->device, g_nuis, sp_nu) : nullptr;
        if (m_el && m_ll && (!nuis || m_nu) && (!g_elems || m_ge) && (!g_nuis || m_gn)) {
            const int64_t n_el_flat = (int64_t)(n_el - 1) * ld + W, n_nu_flat = nuis ? (int64_t)(n_nu - 1) * ld + W : 0;
            rc = grow(ctx, ctx->d_in, ctx->cap_in, (int64_t)(n_el + (nuis ? n_nu : 0)) * ld);
            if (rc) return rc;
            double* z_nuis = nuis ? ctx->d_in + (int64_t)n_el * ld : nullptr;
            hipLaunchKernelGGL(k_copy_in, dim3((unsigned)((n_el_flat + 255) / 256)), dim3(256), 0, st, m_el, ctx->d_in, n_el_flat);
            if (nuis) hipLaunchKernelGGL(k_copy_in, dim3((unsigned)((n_nu_flat + 255) / 256)), dim3(256), 0, st, m_nu, z_nuis, n_nu_flat);
            rc = octo_eval_device(ctx, ds, ctx->d_in, z_nuis, ld, W, m_ll, m_ge, m_gn, st);
            if (rc) return rc;
            pd.active = true;      // not staged: octo_eval_end synchronises the stream, the results are already in place
            return OCTO_OK;
        }
    }
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_in, sizeof(double) * ldd, elems, sizeof(double) * ld, sizeof(double) * W, n_el,
                                 hipMemcpyHostToDevice, st));
    if (nuis)
        HIPCHK(ctx, hipMemcpy2DAsync(d_nuis, sizeof(double) * ldd, nuis, sizeof(double) * ld, sizeof(double) * W, n_nu,
                                     hipMemcpyHostToDevice, st));
    rc = octo_eval_device(ctx, ds, ctx->d_in, d_nuis, ldd, W, d_ll, d_ge, d_gn, st);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ll_out, d_ll, sizeof(double) * W, hipMemcpyDeviceToHost, st));
    if (g_elems)
        HIPCHK(ctx, hipMemcpy2DAsync(g_elems, sizeof(double) * ld, d_ge, sizeof(double) * ldd, sizeof(double) * W, n_el,
                                     hipMemcpyDeviceToHost, st));
    if (g_nuis)
        HIPCHK(ctx, hipMemcpy2DAsync(g_nuis, sizeof(double) * ld, d_gn, sizeof(double) * ldd, sizeof(double) * W, n_nu,
                                     hipMemcpyDeviceToHost, st));
    pd.active = true;
    return OCTO_OK;
}

int32_t octo_eval_begin(octo_ctx* ctx, const octo_dataset* ds, const double* elems, const double* nuis, int64_t ld, int64_t W,
                        double* ll_out, double* g_elems, double* g_nuis) {
    // octo_timing_enable(ctx, -1): the DEVICE time of the whole host-buffer evaluation — SURVEY §8(d)'s clock for the metric ("device time of
    // the octo_eval call (hipEvent), H2D of elems and D2H of ll/grad included") — between an event ahead of the first copy and one behind
    // the last kernel / copy on the context's stream
    if (!ctx || !ctx->timing_whole || ctx->pending.active || W <= 0) return eval_begin_impl(ctx, ds, elems, nuis, ld, W, ll_out, g_elems, g_nuis);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->ev_used >= 4096) { int rcd = drain_timing(ctx); if (rcd) return rcd; }
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t x, y;
        HIPCHK(ctx, hipEventCreate(&x)); HIPCHK(ctx, hipEventCreate(&y));
        ctx->ev_pool.emplace_back(x, y);
    }
    hipEvent_t e0 = ctx->ev_pool[ctx->ev_used].first, e1 = ctx->ev_pool[ctx->ev_used].second;
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    const int32_t rc = eval_begin_impl(ctx, ds, elems, nuis, ld, W, ll_out, g_elems, g_nuis);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    ctx->ev_used++;
    return OCTO_OK;
}

int32_t octo_eval_end(octo_ctx* ctx) {
    if (!ctx) return OCTO_EINVAL;
    octo_ctx::Pending& pd = ctx->pending;
    if (!pd.active) return OCTO_OK;
    pd.active = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!pd.staged) {
        HIPCHK(ctx, hipStreamSynchronize(st));
        free_retired(ctx);
        return OCTO_OK;
    }
    const int64_t W = pd.W, ld = pd.ld, ldd = pd.ldd;
    if (pd.walker_major) {
        int rcz = wait_small(ctx, st, W);
        if (rcz) return rcz;
        for (int64_t w = 0; w < W; ++w) {
            const double* src = ctx->h_out + w * pd.ws_out;
            pd.ll[w] = src[0];
            for (int r = 0; r < pd.n_el_out; ++r) pd.g_elems[(size_t)r * ld + w] = src[1 + r];
            for (int r = 0; r < pd.n_nu_out; ++r) pd.g_nuis[(size_t)r * ld + w] = src[1 + pd.n_el_out + r];
        }
    } else {
        HIPCHK(ctx, hipStreamSynchronize(st));
        std::memcpy(pd.ll, ctx->h_out, sizeof(double) * W);
        for (int r = 0; r < pd.n_el_out; ++r) std::memcpy(pd.g_elems + (size_t)r * ld, ctx->h_out + pd.o_ge + (size_t)r * ldd, sizeof(double) * W);
        for (int r = 0; r < pd.n_nu_out; ++r) std::memcpy(pd.g_nuis + (size_t)r * ld, ctx->h_out + pd.o_gn + (size_t)r * ldd, sizeof(double) * W);
    }
    free_retired(ctx);
    return OCTO_OK;
}

int32_t octo_eval(octo_ctx* ctx, const octo_dataset* ds, const double* elems, const double* nuis, int64_t ld, int64_t W,
                  double* ll_out, double* g_elems, double* g_nuis) {
    // Refused while an octo_eval_begin of this context is outstanding — and that evaluation stays intact (its octo_eval_end still
    // waits and copies out): only a begin of THIS call may be rolled back below.
    if (ctx && ctx->pending.active) return fail(ctx, OCTO_EINVAL, "octo_eval: an octo_eval_begin of this context has not been ended");
    const int rc = octo_eval_begin(ctx, ds, elems, nuis, ld, W, ll_out, g_elems, g_nuis);
    if (rc) { if (ctx) ctx->pending.active = false; return rc; }
    return octo_eval_end(ctx);
}

int32_t octo_host_register(octo_ctx* ctx, void* ptr, int64_t bytes) {
    if (!ctx || !ptr || bytes <= 0) return fail(ctx, OCTO_EINVAL, "octo_host_register: null pointer or empty range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (g_reg.count((uintptr_t)ptr)) return fail(ctx, OCTO_EINVAL, "octo_host_register: this address is already registered");
    }
    if (hipHostRegister(ptr, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, OCTO_EHIP, "octo_host_register: hipHostRegister failed (range not owned by the process, or already pinned)");
    }
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess || !dev) {
        (void)hipGetLastError(); (void)hipHostUnregister(ptr);
        return fail(ctx, OCTO_EHIP, "octo_host_register: no device mapping for the range");
    }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_reg[(uintptr_t)ptr] = HostRange{(size_t)bytes, (char*)dev, ctx->device};
    return OCTO_OK;
}

int32_t octo_host_unregister(octo_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(ctx, OCTO_EINVAL, "octo_host_unregister: null argument");
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_reg.find((uintptr_t)ptr);
        if (it == g_reg.end()) return fail(ctx, OCTO_EINVAL, "octo_host_unregister: address was not registered (pass the base address given to octo_host_register)");
        g_reg.erase(it);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // The registry is process-wide: another context of this device (a second slot of the Julia pool, octo_eval_multi) or a
    // caller-owned stream may still have a copy kernel reading from, or k_finish writing into, the range. Wait for the whole
    // device, not just this context's stream, before the mapping goes away. (The caller must not START an evaluation on the
    // range concurrently with this call — see the header.)
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipHostUnregister(ptr));
    return OCTO_OK;
}

static int32_t kepler_solve_host(octo_ctx* ctx, const double* MA, const double* e, int64_t n, double* E_out, double* sinE_out,
                                 double* cosE_out, bool table) {
    if (!ctx || !MA || !e || !E_out || n < 0) return fail(ctx, OCTO_EINVAL, "octo_kepler_solve: null argument");
    { int rcb = busy(ctx, "octo_kepler_solve"); if (rcb) return rcb; }
    if (n == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = grow(ctx, ctx->d_in, ctx->cap_in, 2 * n);
    if (rc) return rc;
    rc = grow(ctx, ctx->d_out, ctx->cap_out, 3 * n);
    if (rc) return rc;
    hipStream_t st;
    { int rcs = use_stream(ctx, OCTO_STREAM_CTX, &st); if (rcs) return rcs; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_in, MA, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_in + n, e, sizeof(double) * n, hipMemcpyHostToDevice, st));
    if (table)
        hipLaunchKernelGGL(k_kepler<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * 2 * SCT_N, st, ctx->d_in, ctx->d_in + n, n,
                           ctx->d_out, ctx->d_out + n, ctx->d_out + 2 * n, ctx->d_sctab);
    else
        hipLaunchKernelGGL(k_kepler<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ctx->d_in, ctx->d_in + n, n, ctx->d_out,
                           ctx->d_out + n, ctx->d_out + 2 * n, nullptr);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(E_out, ctx->d_out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (sinE_out) HIPCHK(ctx, hipMemcpyAsync(sinE_out, ctx->d_out + n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (cosE_out) HIPCHK(ctx, hipMemcpyAsync(cosE_out, ctx->d_out + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    free_retired(ctx);
    return OCTO_OK;
}

// Test / measurement hook (not part of the C ABI, like octo_debug_poison_lds; bench.py and tests/test_tile_sort.py bind it by name): the state of the
// walker-tile sort of this context — launches that were sorted, probes taken, whether the sort is on for the current (dataset, batch size) and the
// saving the last probe estimated [µs per evaluation].
int32_t octo_debug_tile_state(octo_ctx* ctx, int64_t* sorted_launches, int64_t* probes, int32_t* on, double* last_saving_us) {
    if (!ctx) return OCTO_EINVAL;
    if (sorted_launches) *sorted_launches = ctx->tile_sorted_launches;
    if (probes) *probes = ctx->tile_probes;
    if (on) *on = ctx->tile_on ? 1 : 0;
    if (last_saving_us) *last_saving_us = ctx->tile_last_saving_us;
    return OCTO_OK;
}

// Test hook (not part of the C ABI, like octo_debug_poison_lds): k_main's warm-started Kepler step on its own — octo_kernels.h: k_kepler_warm.
int32_t octo_debug_kepler_warm(octo_ctx* ctx, const double* MA, const double* dM, const double* e, int64_t n, double* sinE_out, double* cosE_out,
                               double* used_warm_out) {
    if (!ctx || !MA || !dM || !e || !sinE_out || !cosE_out || !used_warm_out || n < 0) return fail(ctx, OCTO_EINVAL, "octo_debug_kepler_warm: null argument");
    { int rcb = busy(ctx, "octo_debug_kepler_warm"); if (rcb) return rcb; }
    if (n == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = grow(ctx, ctx->d_in, ctx->cap_in, 3 * n);
    if (rc) return rc;
    rc = grow(ctx, ctx->d_out, ctx->cap_out, 3 * n);
    if (rc) return rc;
    hipStream_t st;
    { int rcs = use_stream(ctx, OCTO_STREAM_CTX, &st); if (rcs) return rcs; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_in, MA, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_in + n, dM, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_in + 2 * n, e, sizeof(double) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_kepler_warm, dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * 2 * SCT_N, st, ctx->d_in, ctx->d_in + n, ctx->d_in + 2 * n, n,
                       ctx->d_out, ctx->d_out + n, ctx->d_out + 2 * n, ctx->d_sctab);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(sinE_out, ctx->d_out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(cosE_out, ctx->d_out + n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(used_warm_out, ctx->d_out + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    free_retired(ctx);
    return OCTO_OK;
}

int32_t octo_kepler_solve(octo_ctx* ctx, const double* MA, const double* e, int64_t n, double* E_out, double* sinE_out,
                          double* cosE_out) {
    return kepler_solve_host(ctx, MA, e, n, E_out, sinE_out, cosE_out, false);
}

int32_t octo_kepler_solve_table(octo_ctx* ctx, const double* MA, const double* e, int64_t n, double* E_out, double* sinE_out,
                                double* cosE_out) {
    return kepler_solve_host(ctx, MA, e, n, E_out, sinE_out, cosE_out, true);
}

int32_t octo_timing_enable(octo_ctx* ctx, int32_t on) {
    if (!ctx) return OCTO_EINVAL;
    ctx->timing_every = on > 0 ? on : 0;
    ctx->timing_whole = on < 0;
    ctx->timing_seq = 0;
    return OCTO_OK;
}

int32_t octo_timing_read(octo_ctx* ctx, double* avg_ms, int64_t* n_launches, int32_t reset) {
    if (!ctx) return OCTO_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = drain_timing(ctx);
    if (rc) return rc;
    if (avg_ms) *avg_ms = ctx->t_n > 0 ? ctx->t_ms / (double)ctx->t_n : 0.0;
    if (n_launches) *n_launches = ctx->t_n;
    if (reset) { ctx->t_ms = 0.0; ctx->t_n = 0; ctx->t_samples.clear(); }
    return OCTO_OK;
}

int32_t octo_timing_stats(octo_ctx* ctx, double* median_ms, double* min_ms, double* max_ms, int64_t* n_launches) {
    if (!ctx) return OCTO_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = drain_timing(ctx);
    if (rc) return rc;
    std::vector<float> v = ctx->t_samples;
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    if (median_ms) *median_ms = n ? (n % 2 ? v[n / 2] : 0.5 * ((double)v[n / 2 - 1] + v[n / 2])) : 0.0;
    if (min_ms) *min_ms = n ? v.front() : 0.0;
    if (max_ms) *max_ms = n ? v.back() : 0.0;
    if (n_launches) *n_launches = (int64_t)n;
    return OCTO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- standard parameterisation
struct octo_model {
    int device = 0;
    const octo_dataset* ds = nullptr;
    int D = 0, n_el = 0, n_nu = 0;
    bool has_nuis = false;
    octo_prior* d_priors = nullptr;
    octo_source* d_esrc = nullptr;
    octo_source* d_nsrc = nullptr;
    int32_t* d_circ = nullptr;   // [n_el + n_nu] LDS slot of each UniformCircular pair in k_model_fwd, or -1
    int n_circ = 0;
    int32_t* d_circ_pair = nullptr;   // [n_circ][2] (i0, i1) of each slot (k_small<MODEL>: one pair per lane)
    bool all_circ_slotted = true;     // every CIRCULAR / TPERI source has a slot (<= MODEL_MAXCIRC of them): required by the fused launch
    double* d_logz = nullptr;         // [D][PRIOR_NC] constants of each prior (prior_density_lanes)
    // the same descriptors as ONE block for the fused small-batch launch (octo_small.h: SmallModel), byte offsets of its sections
    double* d_blob = nullptr;
    int32_t blob_n = 0, off_logz = 0, off_esrc = 0, off_nsrc = -1, off_cslot = 0, off_cpair = 0;
    bool any_ti = false;              // some planet's tp is θ_at_epoch_to_tperi of a Thiele-Innes basis (k_model_fwd<·, TI>)
    bool fused_ok = false;            // all_circ_slotted and the block + the nuisance values fit k_small<MODEL>'s LDS staging
    double* d_buf = nullptr;   // elems | nuis | Jc | gtp | lpp | glp | ll | g_el | g_nu, all [rows][ldw]
    int64_t cap_w = 0;
    double *d_th = nullptr, *d_res = nullptr;   // staging for host buffers
    int64_t cap_th = 0, cap_res = 0;
    int64_t lds_bytes = 0;     // dynamic LDS of k_model_fwd: (4·D + 6·n_circ) × 64 doubles
};

extern "C" {

int32_t octo_model_create(octo_ctx* ctx, const octo_dataset* ds, const octo_prior* priors, int32_t D, const octo_source* elem_src,
                          const octo_source* nuis_src, octo_model** out) {
    if (!ctx || !ds || !priors || !elem_src || !out) return fail(ctx, OCTO_EINVAL, "octo_model_create: null argument");
    if (D < 1) return fail(ctx, OCTO_EINVAL, "octo_model_create: D >= 1");
    if (D > 64) return fail(ctx, OCTO_ENOTSUP, "octo_model_create: 1 <= D <= 64 supported");
    if (ds->device != ctx->device) return fail(ctx, OCTO_EINVAL, "octo_model_create: dataset lives on another device");
    *out = nullptr;
    const int n_el = ds->n_planets * OCTO_N_EL, n_nu = ds->n_obs * OCTO_N_NUIS;
    auto check_src = [&](const octo_source& s, bool elem) {
        if (s.kind < OCTO_SRC_CONST || s.kind > OCTO_SRC_TPERI) return false;
        if (s.kind == OCTO_SRC_THETA && (s.i0 < 0 || s.i0 >= D)) return false;
        if ((s.kind == OCTO_SRC_CIRCULAR || s.kind == OCTO_SRC_TPERI) && (s.i0 < 0 || s.i0 >= D || s.i1 < 0 || s.i1 >= D)) return false;
        if (s.kind == OCTO_SRC_TPERI && !elem) return false;
        return true;
    };
    for (int k = 0; k < D; ++k)
        if (priors[k].kind < OCTO_PRIOR_UNIFORM || priors[k].kind > OCTO_PRIOR_SINE) return fail(ctx, OCTO_EINVAL, "octo_model_create: unknown prior kind");
    for (int k = 0; k < n_el; ++k) {
        if (!check_src(elem_src[k], true)) return fail(ctx, OCTO_EINVAL, "octo_model_create: bad element source");
        if (elem_src[k].kind == OCTO_SRC_TPERI && k % OCTO_N_EL != OCTO_EL_TP)
            return fail(ctx, OCTO_EINVAL, "octo_model_create: OCTO_SRC_TPERI belongs in the tp row");
    }
    bool has_nuis = false;
    if (nuis_src)
        for (int k = 0; k < n_nu; ++k) {
            if (!check_src(nuis_src[k], false)) return fail(ctx, OCTO_EINVAL, "octo_model_create: bad nuisance source");
            const int r = k % OCTO_N_NUIS; const int kind = ds->h_obs[k / OCTO_N_NUIS].kind;
            const bool astrom = kind <= OCTO_ASTROM_SEPPA || kind == OCTO_ONEIL_RADEC || kind == OCTO_ONEIL_SEPPA;
            const double dflt = (astrom && r == OCTO_NU_PLATESCALE) ? 1.0 : 0.0;
            if (nuis_src[k].kind != OCTO_SRC_CONST || nuis_src[k].value != dflt || kind == OCTO_HGCA) has_nuis = true;
        }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    octo_model* m = new (std::nothrow) octo_model();
    if (!m) return fail(ctx, OCTO_ENOMEM, "octo_model_create: host allocation failed");
    m->device = ctx->device; m->ds = ds; m->D = D; m->n_el = n_el; m->n_nu = n_nu; m->has_nuis = has_nuis;
    for (int k = 0; k < n_el; ++k) m->any_ti = m->any_ti || (elem_src[k].kind == OCTO_SRC_TPERI && (elem_src[k].flags & OCTO_SRC_FLAG_TI));
    auto bail = [&](int code, const char* msg) { octo_model_destroy(m); return fail(ctx, code, msg); };
    if (hipMalloc((void**)&m->d_priors, sizeof(octo_prior) * D) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
    if (hipMalloc((void**)&m->d_esrc, sizeof(octo_source) * n_el) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
    if (hipMemcpy(m->d_priors, priors, sizeof(octo_prior) * D, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_esrc, elem_src, sizeof(octo_source) * n_el, hipMemcpyHostToDevice) != hipSuccess)
        return bail(OCTO_EHIP, "octo_model_create: upload failed");
    if (nuis_src && n_nu > 0) {
        if (hipMalloc((void**)&m->d_nsrc, sizeof(octo_source) * n_nu) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
        if (hipMemcpy(m->d_nsrc, nuis_src, sizeof(octo_source) * n_nu, hipMemcpyHostToDevice) != hipSuccess) return bail(OCTO_EHIP, "octo_model_create: upload failed");
    }
    {
        std::vector<int32_t> slot(n_el + n_nu, -1), pairs;
        for (int k = 0; k < n_el + n_nu; ++k) {
            if (k >= n_el && !nuis_src) break;
            const octo_source& sc = k < n_el ? elem_src[k] : nuis_src[k - n_el];
            if (sc.kind == OCTO_SRC_CIRCULAR || sc.kind == OCTO_SRC_TPERI) {
                if (m->n_circ < MODEL_MAXCIRC) {
                    slot[k] = m->n_circ++;
                    pairs.push_back(sc.i0); pairs.push_back(sc.i1);
                } else {
                    m->all_circ_slotted = false;
                }
            }
        }
        pairs.resize(std::max<size_t>(pairs.size(), 2), 0);
        if (hipMalloc((void**)&m->d_circ_pair, sizeof(int32_t) * pairs.size()) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
        if (hipMemcpy(m->d_circ_pair, pairs.data(), sizeof(int32_t) * pairs.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(OCTO_EHIP, "octo_model_create: upload failed");
        // constants of each prior (prior_density_lanes): −log(Φ(hi) − Φ(lo)), the truncated Normal's normalisation (Distributions.jl: truncated),
        // 1/(b − a), −log(b − a) | log(b/a) | −log σ, 1/σ
        std::vector<double> logz((size_t)D * PRIOR_NC, std::nan(""));
        for (int k = 0; k < D; ++k) {
            const octo_prior& pr = priors[k];
            double* c = &logz[(size_t)k * PRIOR_NC];
            double a = -INFINITY, b = INFINITY;
            if (pr.kind == OCTO_PRIOR_UNIFORM || pr.kind == OCTO_PRIOR_LOGUNIFORM) { a = pr.p0; b = pr.p1; }
            else if (pr.kind == OCTO_PRIOR_TRUNCNORMAL) { a = pr.lo; b = pr.hi; }
            else if (pr.kind == OCTO_PRIOR_SINE) { a = 0.0 + 2.220446049250313e-16; b = PI - 2.220446049250313e-16; }
            c[1] = 1.0 / (b - a);
            if (pr.kind == OCTO_PRIOR_UNIFORM) c[2] = -std::log(b - a);
            else if (pr.kind == OCTO_PRIOR_LOGUNIFORM) c[2] = std::log(b / a);
            else if (pr.kind == OCTO_PRIOR_NORMAL || pr.kind == OCTO_PRIOR_TRUNCNORMAL) { c[2] = -std::log(pr.p1); c[3] = 1.0 / pr.p1; }
            if (pr.kind == OCTO_PRIOR_TRUNCNORMAL) {
                const double lo = std::isfinite(pr.lo) ? 0.5 * std::erfc(-((pr.lo - pr.p0) / pr.p1) * 0.70710678118654752440) : 0.0;
                const double hi = std::isfinite(pr.hi) ? 0.5 * std::erfc(-((pr.hi - pr.p0) / pr.p1) * 0.70710678118654752440) : 1.0;
                c[0] = -std::log(hi - lo);
            }
        }
        if (hipMalloc((void**)&m->d_logz, sizeof(double) * logz.size()) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
        if (hipMemcpy(m->d_logz, logz.data(), sizeof(double) * logz.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(OCTO_EHIP, "octo_model_create: upload failed");
        if (hipMalloc((void**)&m->d_circ, sizeof(int32_t) * slot.size()) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
        if (hipMemcpy(m->d_circ, slot.data(), sizeof(int32_t) * slot.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(OCTO_EHIP, "octo_model_create: upload failed");
        {
            std::vector<unsigned char> blob;
            auto put = [&](const void* src, size_t bytes) {
                const size_t off = blob.size();
                blob.resize(off + (bytes + 7) / 8 * 8, 0);
                std::memcpy(blob.data() + off, src, bytes);
                return (int32_t)off;
            };
            put(priors, sizeof(octo_prior) * D);      // at byte 0
            m->off_logz = put(logz.data(), sizeof(double) * logz.size());
            m->off_esrc = put(elem_src, sizeof(octo_source) * n_el);
            m->off_nsrc = (nuis_src && n_nu > 0) ? put(nuis_src, sizeof(octo_source) * n_nu) : -1;
            m->off_cslot = put(slot.data(), sizeof(int32_t) * slot.size());
            m->off_cpair = put(pairs.data(), sizeof(int32_t) * pairs.size());
            m->blob_n = (int32_t)(blob.size() / 8);
            if (hipMalloc((void**)&m->d_blob, blob.size()) != hipSuccess) return bail(OCTO_ENOMEM, "octo_model_create: hipMalloc failed");
            if (hipMemcpy(m->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(OCTO_EHIP, "octo_model_create: upload failed");
            m->fused_ok = m->all_circ_slotted && m->blob_n <= SMALL_BLOB_MAX && (!has_nuis || n_nu <= SMALL_MX_NU);
        }
    }
    // k_model_fwd shares x, dx, p, dp of every prior and 6 numbers per UniformCircular pair through LDS: 512 B each.
    m->lds_bytes = (int64_t)sizeof(double) * (4 * D + 6 * m->n_circ) * WAVE;
    // (the kernel also declares 1 KB of STATIC LDS — sfin, octo_model.h — which counts against the same per-block limits: ADVICE r4)
    const int64_t lds_static = 16 * WAVE;
    if (m->lds_bytes + lds_static > ctx->max_lds)
        return bail(OCTO_ENOTSUP, "octo_model_create: the model needs more LDS per block than this device has ((4·D + 6·n_circular)·512 B)");
    if (m->lds_bytes + lds_static > 48 * 1024 &&
        (hipFuncSetAttribute((const void*)k_model_fwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes) != hipSuccess ||
         hipFuncSetAttribute((const void*)k_model_fwd<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes) != hipSuccess ||
         hipFuncSetAttribute((const void*)k_model_fwd<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes) != hipSuccess ||
         hipFuncSetAttribute((const void*)k_model_fwd<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes) != hipSuccess)) {
        (void)hipGetLastError();
        if (m->lds_bytes + lds_static > 64 * 1024) return bail(OCTO_ENOTSUP, "octo_model_create: cannot raise k_model_fwd's dynamic LDS limit for this model");
    }
    *out = m;
    return OCTO_OK;
}

int32_t octo_model_destroy(octo_model* m) {
    if (!m) return OCTO_OK;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->d_priors); (void)hipFree(m->d_esrc); (void)hipFree(m->d_nsrc); (void)hipFree(m->d_circ); (void)hipFree(m->d_circ_pair); (void)hipFree(m->d_logz); (void)hipFree(m->d_blob); (void)hipFree(m->d_buf);
    (void)hipFree(m->d_th); (void)hipFree(m->d_res);
    delete m;
    return OCTO_OK;
}

int32_t octo_model_logpost_device(octo_ctx* ctx, octo_model* m, const double* d_theta_t, int64_t ld, int64_t W, double* d_lp,
                                  double* d_grad, void* hip_stream) {
    if (!ctx || !m || !d_theta_t || !d_lp) return fail(ctx, OCTO_EINVAL, "octo_model_logpost_device: null argument");
    if (W < 0 || (ld < W && ctx->stage_ws_in == 0)) return fail(ctx, OCTO_EINVAL, "octo_model_logpost_device: need 0 <= W <= ld");
    if (m->device != ctx->device) return fail(ctx, OCTO_EINVAL, "octo_model_logpost_device: model lives on another device");
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    { int rcs = use_stream(ctx, hip_stream, &st); if (rcs) return rcs; }
    if (small_eligible(ctx, m->ds, W, true) && m->fused_ok && (m->ds->n_hgca == 0 || hgca_in_small(W))) {
        // one launch: θ_t -> priors, elements, likelihood, ∇θ_t inside k_small<MODEL> (octo_small.h)
        SmallModel sm;
        std::memset(&sm, 0, sizeof(sm));
        sm.blob = m->d_blob; sm.blob_n = m->blob_n; sm.off_logz = m->off_logz; sm.off_esrc = m->off_esrc; sm.off_nsrc = m->off_nsrc;
        sm.off_cslot = m->off_cslot; sm.off_cpair = m->off_cpair; sm.n_circ = m->n_circ; sm.n_el = m->n_el; sm.n_nu = m->n_nu; sm.D = m->D;
        sm.theta_t = d_theta_t; sm.lp_out = d_lp; sm.grad_out = d_grad;
        if (ctx->stage_ws_in > 0) { sm.ld_t = 1; sm.ws_t = ctx->stage_ws_in; sm.ld_o = 1; sm.ws_o = ctx->stage_ws_out; }      // walker-major staging
        else { sm.ld_t = ld; sm.ws_t = 1; sm.ld_o = ld; sm.ws_o = 1; }
        sm.k_yr = ctx->consts.kepler_year_to_julian_day; sm.yd = ctx->consts.year2day_julian;
        return eval_impl(ctx, m->ds, nullptr, nullptr, 1, W, nullptr, nullptr, nullptr, st, &sm, d_grad != nullptr, m->has_nuis);
    }
    const int64_t ldw = (W + WAVE - 1) / WAVE * WAVE;
    const int n_in = m->n_el + m->n_nu;
    const int64_t rows = (int64_t)n_in + 2 * (int64_t)n_in + m->n_el + 1 + m->D + 1 + n_in;
    if (ldw > m->cap_w) {
        if (m->d_buf) { ctx->retired.push_back(m->d_buf); m->d_buf = nullptr; m->cap_w = 0; }
        const int64_t cap = ldw + ldw / 2;
        HIPCHK(ctx, hipMalloc((void**)&m->d_buf, sizeof(double) * (size_t)(rows * cap)));
        m->cap_w = cap;
    }
    const int64_t L = m->cap_w;
    ModelArgs a;
    std::memset(&a, 0, sizeof(a));
    a.priors = m->d_priors; a.prior_logz = m->d_logz; a.esrc = m->d_esrc; a.nsrc = m->d_nsrc; a.obs = m->ds->d_obs;
    a.D = m->D; a.n_el = m->n_el; a.n_nu = m->n_nu; a.n_planets = m->ds->n_planets;
    a.theta_t = d_theta_t; a.ld = ld; a.W = W; a.ldw = L;
    double* p = m->d_buf;
    a.elems = p; p += (int64_t)m->n_el * L;
    a.nuis = p; p += (int64_t)m->n_nu * L;
    a.Jc = p; p += 2 * (int64_t)n_in * L;
    a.gtp = p; p += (int64_t)m->n_el * L;
    a.lpp = p; p += L;
    a.glp = p; p += (int64_t)m->D * L;
    double* d_ll = p; p += L;
    double* d_gel = p; p += (int64_t)m->n_el * L;
    double* d_gnu = p;
    a.ll = d_ll; a.g_el = d_gel; a.g_nu = m->has_nuis ? d_gnu : nullptr;
    a.lp_out = d_lp; a.grad_out = d_grad;
    a.k_yr = ctx->consts.kepler_year_to_julian_day; a.yd = ctx->consts.year2day_julian;
    // θ_t -> kernel inputs, prior sum (+ the compact Jacobian with a gradient): one block of NW waves per tile of 64 walkers (octo_model.h).
    // Tried and not kept (profiles/r4_model_streams_ab.txt, with the dense forward-mode kernel of rounds 1-3): values on the caller's stream +
    // the Jacobian launch on a second stream beside k_main — fork / join events cost more than the hidden half saved.
    const bool grad = d_grad != nullptr;
    a.circ_slot = m->d_circ; a.circ_pair = m->d_circ_pair; a.n_circ = m->n_circ;
    {
        const int want = std::max(2 * m->n_circ + m->D, a.n_planets + 2);
        const dim3 grid((unsigned)((W + 63) / 64)), block(64, (unsigned)std::max(1, std::min(want, m->any_ti ? 8 : 16)));
        if (m->any_ti) {
            if (grad) hipLaunchKernelGGL((k_model_fwd<true, true>), grid, block, (size_t)m->lds_bytes, st, a);
            else hipLaunchKernelGGL((k_model_fwd<false, true>), grid, block, (size_t)m->lds_bytes, st, a);
        } else {
            if (grad) hipLaunchKernelGGL((k_model_fwd<true, false>), grid, block, (size_t)m->lds_bytes, st, a);
            else hipLaunchKernelGGL((k_model_fwd<false, false>), grid, block, (size_t)m->lds_bytes, st, a);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    // lp = prior + ll and ∇θ_t = Jᵀḡ + ∇prior: inside k_finish (model_tail), tile by tile as the adjoints become known
    octo_ctx::ModelTail mt;
    mt.Jc = a.Jc; mt.gtp = a.gtp; mt.esrc = m->d_esrc; mt.nsrc = m->d_nsrc; mt.glp = a.glp; mt.lpp = a.lpp; mt.lp = d_lp; mt.grad = d_grad;
    mt.ld = L; mt.ldo = ld; mt.D = m->D; mt.n_nu = m->has_nuis ? m->n_nu : 0;
    bool tail_applied = false;
    int rc = eval_impl(ctx, m->ds, a.elems, m->has_nuis ? a.nuis : nullptr, L, W, d_ll, grad ? d_gel : nullptr,
                       (grad && m->has_nuis) ? d_gnu : nullptr, st, nullptr, grad, m->has_nuis, &mt, &tail_applied);
    if (rc) return rc;
    if (tail_applied) return OCTO_OK;
    hipLaunchKernelGGL(k_model_bwd, dim3((unsigned)((W + 255) / 256), (unsigned)(d_grad ? m->D : 1)), dim3(256), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return OCTO_OK;
}

int32_t octo_model_logpost(octo_ctx* ctx, octo_model* m, const double* theta_t, int64_t ld, int64_t W, double* lp_out, double* grad_out) {
    if (!ctx || !m || !theta_t || !lp_out) return fail(ctx, OCTO_EINVAL, "octo_model_logpost: null argument");
    if (W < 0 || ld < W) return fail(ctx, OCTO_EINVAL, "octo_model_logpost: need 0 <= W <= ld");
    { int rcb = busy(ctx, "octo_model_logpost"); if (rcb) return rcb; }
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t ldd = (W + 63) / 64 * 64;
    hipStream_t st;
    { int rcs = use_stream(ctx, OCTO_STREAM_CTX, &st); if (rcs) return rcs; }
    const int64_t n_in = (int64_t)m->D * ldd, n_out = (int64_t)(grad_out ? m->D + 1 : 1) * ldd;
    if (small_eligible(ctx, m->ds, W, true) && m->fused_ok && (m->ds->n_hgca == 0 || hgca_in_small(W)) && W <= ctx->mapped_w) {
        // one θ_t per call (NUTS): the fused launch on mapped pinned buffers, no copy engine (see octo_eval) — θ_t of one walker
        // contiguous on the way in, [lp | ∇θ_t] on the way out, completion by flag
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_model_logpost: pinned staging allocation failed");
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        const int64_t D = m->D, ws_o = grad_out ? D + 1 : 1;
        for (int64_t w = 0; w < W; ++w)
            for (int r = 0; r < D; ++r) ctx->h_in[w * D + r] = theta_t[(size_t)r * ld + w];
        ctx->inl.n = 0;
        if (W == 1 && D <= SMALL_INL) {      // one θ_t: inside the kernel arguments (see octo_eval)
            ctx->inl.n = D; ctx->inl.pad = 0;
            for (int r = 0; r < D; ++r) ctx->inl.v[r] = theta_t[(size_t)r * ld];
        }
        ctx->stage_ws_in = D; ctx->stage_ws_out = ws_o;
        ctx->flag_request = W <= ctx->flag_w; ctx->flag_armed = false;
        int rcz = octo_model_logpost_device(ctx, m, m_in, 1, W, m_out, grad_out ? m_out + 1 : nullptr, st);
        ctx->flag_request = false; ctx->stage_ws_in = ctx->stage_ws_out = 0; ctx->inl.n = 0;
        if (rcz) return rcz;
        rcz = wait_small(ctx, st, W);
        if (rcz) return rcz;
        for (int64_t w = 0; w < W; ++w) {
            lp_out[w] = ctx->h_out[w * ws_o];
            if (grad_out) for (int r = 0; r < D; ++r) grad_out[(size_t)r * ld + w] = ctx->h_out[w * ws_o + 1 + r];
        }
        free_retired(ctx);
        return OCTO_OK;
    }
    int rc = grow(ctx, m->d_th, m->cap_th, (int64_t)m->D * ldd);
    if (rc) return rc;
    rc = grow(ctx, m->d_res, m->cap_res, (int64_t)(m->D + 1) * ldd);
    if (rc) return rc;
    if (small_eligible(ctx, m->ds, W, true) && m->fused_ok && (m->ds->n_hgca == 0 || hgca_in_small(W))) {
        // the fused launch beyond the mapped-input range: k_stage_in brings θ_t walker-major into device memory, [lp | ∇θ_t] come
        // back through the mapped buffer + flags (see octo_eval)
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_model_logpost: pinned staging allocation failed");
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        const int64_t D = m->D, ws_o = grad_out ? D + 1 : 1;
        for (int r = 0; r < D; ++r) std::memcpy(ctx->h_in + (size_t)r * ldd, theta_t + (size_t)r * ld, sizeof(double) * W);
        hipLaunchKernelGGL(k_stage_in, dim3((unsigned)((W * D + 255) / 256)), dim3(256), 0, st, m_in, ldd, W, (int)D, m->d_th);
        ctx->stage_ws_in = D; ctx->stage_ws_out = ws_o;
        ctx->flag_request = false; ctx->flag_armed = false;
        int rcz = octo_model_logpost_device(ctx, m, m->d_th, 1, W, m_out, grad_out ? m_out + 1 : nullptr, st);
        ctx->flag_request = false; ctx->stage_ws_in = ctx->stage_ws_out = 0;
        if (rcz) return rcz;
        rcz = wait_small(ctx, st, W);
        if (rcz) return rcz;
        for (int64_t w = 0; w < W; ++w) {
            lp_out[w] = ctx->h_out[w * ws_o];
            if (grad_out) for (int r = 0; r < D; ++r) grad_out[(size_t)r * ld + w] = ctx->h_out[w * ws_o + 1 + r];
        }
        free_retired(ctx);
        return OCTO_OK;
    }
    if ((n_in + n_out) * (int64_t)sizeof(double) <= stage_bytes(ctx)) {      // mid-size batch: mapped pinned buffers + a copy kernel (see octo_eval)
        if (!grow_pinned(ctx->h_in, ctx->cap_hin, n_in) || !grow_pinned(ctx->h_out, ctx->cap_hout, n_out))
            return fail(ctx, OCTO_ENOMEM, "octo_model_logpost: pinned staging allocation failed");
        for (int r = 0; r < m->D; ++r) std::memcpy(ctx->h_in + (size_t)r * ldd, theta_t + (size_t)r * ld, sizeof(double) * W);
        double *m_in = nullptr, *m_out = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_in, ctx->h_in, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&m_out, ctx->h_out, 0));
        // θ_t's only reader on this route is k_model_fwd, which reads every θ_t[k][w] exactly once (coalesced rows): it takes the mapped
        // buffer itself — a PCIe read under its first phase instead of a copy kernel in front of the launch (rounds 2-4: k_copy_in, one more
        // launch boundary per callback)
        rc = octo_model_logpost_device(ctx, m, m_in, ldd, W, m_out, grad_out ? m_out + ldd : nullptr, st);
        if (rc) return rc;
        HIPCHK(ctx, hipStreamSynchronize(st));
        std::memcpy(lp_out, ctx->h_out, sizeof(double) * W);
        if (grad_out) for (int r = 0; r < m->D; ++r) std::memcpy(grad_out + (size_t)r * ld, ctx->h_out + (size_t)(1 + r) * ldd, sizeof(double) * W);
        free_retired(ctx);
        return OCTO_OK;
    }
    HIPCHK(ctx, hipMemcpy2DAsync(m->d_th, sizeof(double) * ldd, theta_t, sizeof(double) * ld, sizeof(double) * W, m->D, hipMemcpyHostToDevice, st));
    rc = octo_model_logpost_device(ctx, m, m->d_th, ldd, W, m->d_res, grad_out ? m->d_res + ldd : nullptr, st);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(lp_out, m->d_res, sizeof(double) * W, hipMemcpyDeviceToHost, st));
    if (grad_out)
        HIPCHK(ctx, hipMemcpy2DAsync(grad_out, sizeof(double) * ld, m->d_res + ldd, sizeof(double) * ldd, sizeof(double) * W, m->D, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    free_retired(ctx);
    return OCTO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- OFTI marginal likelihood
extern "C" {

int32_t octo_ofti_create(octo_ctx* ctx, const double* epochs, const double* ra, const double* dec, const double* s_ra,
                         const double* s_dec, const double* cor, int64_t n, double sigma_abfg, octo_ofti** out) {
    if (!ctx || !out || n < 0 || (n > 0 && (!epochs || !ra || !dec || !s_ra || !s_dec))) return fail(ctx, OCTO_EINVAL, "octo_ofti_create: null argument");
    if (!(sigma_abfg > 0.0) || n > 0x7fffffff) return fail(ctx, OCTO_EINVAL, "octo_ofti_create: bad sigma_ABFG or size");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    octo_ofti* h = new (std::nothrow) octo_ofti();
    if (!h) return fail(ctx, OCTO_ENOMEM, "octo_ofti_create: host allocation failed");
    h->device = ctx->device; h->n = n;
    std::vector<double> rows((size_t)n * ROW_STRIDE, 0.0);
    for (int64_t j = 0; j < n; ++j) {
        // inverse covariance of one epoch, parameterizations.jl:359-366
        const double sr = s_ra[j], sd = s_dec[j], rho = cor ? cor[j] : 0.0;
        if (!std::isfinite(epochs[j]) || !std::isfinite(ra[j]) || !std::isfinite(dec[j]) || !(sr > 0.0) || !std::isfinite(sr) ||
            !(sd > 0.0) || !std::isfinite(sd) || !(std::fabs(rho) < 1.0)) {
            delete h;
            return fail(ctx, OCTO_EINVAL, "octo_ofti_create: row " + std::to_string(j) + ": epochs and positions must be finite, "
                                          "uncertainties finite and > 0, |cor| < 1");
        }
        const double det = sr * sr * sd * sd * (1.0 - rho * rho);
        const double wrr = sd * sd / det, wdd = sr * sr / det, wrd = -rho * sr * sd / det;
        double* r = &rows[(size_t)j * ROW_STRIDE];
        r[0] = epochs[j]; r[1] = wrr; r[2] = wdd; r[3] = wrd;
        r[4] = wrr * ra[j] + wrd * dec[j]; r[5] = wdd * dec[j] + wrd * ra[j];
        h->data_quad += ra[j] * r[4] + dec[j] * r[5];              // dot(d, W, d), :387
        h->log_det_data_cov += std::log(det);                      // :394-400
    }
    h->lambda = 1.0 / (sigma_abfg * sigma_abfg);
    h->log_det_prior_inv = 4.0 * std::log(h->lambda);              // :391
    h->n_log2pi = (double)n * std::log(TWO_PI);
    if (n > 0) {
        if (hipMalloc((void**)&h->d_rows, sizeof(double) * rows.size()) != hipSuccess) { delete h; return fail(ctx, OCTO_ENOMEM, "octo_ofti_create: hipMalloc failed"); }
        if (hipMemcpy(h->d_rows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(h->d_rows); delete h; return fail(ctx, OCTO_EHIP, "octo_ofti_create: upload failed");
        }
    }
    *out = h;
    return OCTO_OK;
}

int32_t octo_ofti_destroy(octo_ofti* h) {
    if (!h) return OCTO_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_rows);
    delete h;
    return OCTO_OK;
}

int32_t octo_ofti_eval_device(octo_ctx* ctx, const octo_ofti* h, const double* d_nl, int64_t ld, int64_t W, double* d_abfg,
                              double* d_logml, void* hip_stream) {
    if (!ctx || !h || !d_nl || !d_logml) return fail(ctx, OCTO_EINVAL, "octo_ofti_eval_device: null argument");
    if (W < 0 || ld < W) return fail(ctx, OCTO_EINVAL, "octo_ofti_eval_device: need 0 <= W <= ld");
    if (h->device != ctx->device) return fail(ctx, OCTO_EINVAL, "octo_ofti_eval_device: handle lives on another device");
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    { int rcs = use_stream(ctx, hip_stream, &st); if (rcs) return rcs; }
    OftiArgs a;
    std::memset(&a, 0, sizeof(a));
    const int64_t cols = (W + WAVE - 1) / WAVE;
    int chunk = 32;
    {   // same sizing rule as the likelihood kernel: an exact number of rounds of resident blocks, equal tasks
        int& blocks_per_cu = ctx->occupancy[0xffff0001u];
        if (blocks_per_cu == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ofti_main, WAVE * WPB, sizeof(double) * (2 * SCT_N + OFTI_NACC * WAVE)) != hipSuccess || nb < 1) nb = 2;
            blocks_per_cu = nb;
        }
        const int64_t key = plan_key(ctx, W, h->n, blocks_per_cu);
        if (key < 0) chunk = (int)-key;
        else {
            const int64_t t_o = std::min<int64_t>(std::max<int64_t>(key, 1), std::max<int64_t>(1, h->n / (32 * WPB)));
            const int64_t rows_per_task = (h->n + t_o - 1) / t_o;
            chunk = (int)std::max<int64_t>(1, (rows_per_task + WPB - 1) / WPB);
        }
    }
    a.rows = h->d_rows; a.n_rows = (int32_t)h->n; a.chunk = chunk; a.sctab = ctx->d_sctab;
    a.n_tasks = (int32_t)((h->n + (int64_t)chunk * WPB - 1) / ((int64_t)chunk * WPB));
    a.nl = d_nl; a.ld = ld; a.W = W; a.ldw = cols * WAVE;
    int rc = grow(ctx, ctx->d_partials, ctx->cap_part, (int64_t)std::max(a.n_tasks, 1) * OFTI_NACC * a.ldw);
    if (rc) return rc;
    a.partials = ctx->d_partials; a.abfg = d_abfg; a.logml = d_logml;
    a.k_yr = ctx->consts.kepler_year_to_julian_day; a.lambda = h->lambda; a.data_quad = h->data_quad;
    a.log_det_data_cov = h->log_det_data_cov; a.log_det_prior_inv = h->log_det_prior_inv; a.n_log2pi = h->n_log2pi;
    if (a.n_tasks > 0)
        hipLaunchKernelGGL(k_ofti_main, dim3((unsigned)cols, (unsigned)a.n_tasks), dim3(WAVE * WPB), sizeof(double) * (2 * SCT_N + OFTI_NACC * WAVE), st, a);
    hipLaunchKernelGGL(k_ofti_finish, dim3((unsigned)cols), dim3(WAVE), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    return OCTO_OK;
}

int32_t octo_ofti_eval(octo_ctx* ctx, const octo_ofti* h, const double* nl, int64_t ld, int64_t W, double* abfg_out, double* logml_out) {
    if (!ctx || !h || !nl || !logml_out) return fail(ctx, OCTO_EINVAL, "octo_ofti_eval: null argument");
    if (W < 0 || ld < W) return fail(ctx, OCTO_EINVAL, "octo_ofti_eval: need 0 <= W <= ld");
    { int rcb = busy(ctx, "octo_ofti_eval"); if (rcb) return rcb; }
    if (W == 0) return OCTO_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t ldd = (W + 63) / 64 * 64;
    int rc = grow(ctx, ctx->d_in, ctx->cap_in, 5 * ldd);
    if (rc) return rc;
    rc = grow(ctx, ctx->d_out, ctx->cap_out, 5 * ldd);
    if (rc) return rc;
    hipStream_t st;
    { int rcs = use_stream(ctx, OCTO_STREAM_CTX, &st); if (rcs) return rcs; }
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_in, sizeof(double) * ldd, nl, sizeof(double) * ld, sizeof(double) * W, 5, hipMemcpyHostToDevice, st));
    rc = octo_ofti_eval_device(ctx, h, ctx->d_in, ldd, W, abfg_out ? ctx->d_out + ldd : nullptr, ctx->d_out, st);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(logml_out, ctx->d_out, sizeof(double) * W, hipMemcpyDeviceToHost, st));
    if (abfg_out)
        HIPCHK(ctx, hipMemcpy2DAsync(abfg_out, sizeof(double) * ld, ctx->d_out + ldd, sizeof(double) * ldd, sizeof(double) * W, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    free_retired(ctx);
    return OCTO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- parallel tempering swap
namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_pt_swap(const double* __restrict__ ll, const double* __restrict__ beta,
                                                 int32_t* slot2rep, int n_temps, int64_t n_chains, int parity,
                                                 uint64_t seed, uint64_t step, int32_t* accepted) {
    const int n_pairs = (n_temps - parity) / 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_chains * n_pairs) return;
    const int64_t c = idx / n_pairs;
    const int t = parity + 2 * (int)(idx % n_pairs);
    int32_t* s2r = slot2rep + c * n_temps;
    const int ri = s2r[t], rj = s2r[t + 1];
    const double li = ll[(int64_t)ri * n_chains + c], lj = ll[(int64_t)rj * n_chains + c];      // [replica][chain], as gathered
    const double logA = (beta[t] - beta[t + 1]) * (lj - li);
    // counter-based uniform in (0,1]: identical on every rank for the same (seed, step, chain, slot)
    const uint64_t h = mix64(mix64(mix64(seed ^ 0x6f63746f50545357ull) + step) + (uint64_t)c * 0x100000001b3ull + (uint64_t)t);
    const double u = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const bool acc = isfinite(li) && isfinite(lj) ? (log(u) < logA) : (isfinite(lj) && !isfinite(li) && beta[t] > beta[t + 1]);
    if (acc) {
        s2r[t] = rj; s2r[t + 1] = ri;
        if (accepted) atomicAdd(&accepted[t], 1);
    }
}

}  // namespace

extern "C" int32_t octo_pt_swap_device(octo_ctx* ctx, const double* d_ll_by_replica, const double* d_beta,
                                       int32_t* d_slot2rep, int32_t n_temps, int64_t n_chains, int32_t parity,
                                       uint64_t seed, uint64_t step, int32_t* d_accepted, void* hip_stream) {
    if (!ctx || !d_ll_by_replica || !d_beta || !d_slot2rep) return fail(ctx, OCTO_EINVAL, "octo_pt_swap_device: null argument");
    if (n_temps < 2 || n_chains < 1 || (parity != 0 && parity != 1)) return fail(ctx, OCTO_EINVAL, "octo_pt_swap_device: bad sizes");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    { int rcs = use_stream(ctx, hip_stream, &st); if (rcs) return rcs; }
    const int n_pairs = (n_temps - parity) / 2;
    if (n_pairs == 0) return OCTO_OK;
    const int64_t n = n_chains * n_pairs;
    hipLaunchKernelGGL(k_pt_swap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_ll_by_replica, d_beta, d_slot2rep,
                       n_temps, n_chains, parity, seed, step, d_accepted);
    HIPCHK(ctx, hipGetLastError());
    return OCTO_OK;
}
