// octo_comm.hip — the one collective of the path, inside the C ABI: the parallel-tempering swap step over RCCL
// (BASELINE config 5; the reference's exchange is Pigeons' communication step reached through
// ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:76-128, docs/src/parallel-sampling.md:64-80), and the single-process
// multi-device split of a host batch ("one host thread (or one stream) per device").
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already maps an RCCL — torch's bundled copy, or an MPI
// build's — must keep exactly one, and RTLD_NOLOAD finds that one first; a plain Julia host gets /opt/rocm/lib/librccl.so.1.
// Only four entry points are used and they are declared here with RCCL's own prototypes (rccl.h: ncclGetUniqueId :187,
// ncclCommInitRank :220, ncclCommDestroy :260, ncclAllGather :678; ncclFloat64 = 8).
#include <dlfcn.h>

#include <functional>

#include "octo_host.h"

using namespace octo;

namespace {

struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
typedef int (*fn_get_unique_id)(RcclUniqueId*);
typedef int (*fn_comm_init_rank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(RcclComm);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, RcclComm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_FLOAT64 = 8;

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    std::string err;
};

void rccl_load(Rccl& r) {
    const char* cands[] = {std::getenv("OCTO_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy this process already maps, if any
    for (const char* c : cands) {
        if (h) break;
        if (c && *c) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        const char* e = dlerror();      // one call: dlerror() clears the message it returns
        r.err = std::string("cannot load librccl.so.1: ") + (e ? e : "not found");
        return;
    }
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) { r.err = "librccl.so.1 lacks the NCCL API"; return; }
    r.handle = h;
}

Rccl& rccl() {      // bound once, whichever host thread asks first
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, rccl_load, std::ref(r));
    return r;
}

std::string rccl_msg(const char* what, int code) {
    Rccl& r = rccl();
    return std::string(what) + ": " + (r.error_string ? r.error_string(code) : "RCCL error") + " (" + std::to_string(code) + ")";
}

}  // namespace

extern "C" {

int32_t octo_comm_unique_id(uint8_t* out128) {
    if (!out128) return OCTO_EINVAL;
    Rccl& r = rccl();
    if (!r.handle) return OCTO_ENODEV;
    RcclUniqueId id;
    if (r.get_unique_id(&id) != 0) return OCTO_EHIP;
    std::memcpy(out128, id.internal, 128);
    return OCTO_OK;
}

int32_t octo_comm_create(octo_ctx* ctx, const uint8_t* unique_id128, int32_t rank, int32_t world) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return fail(ctx, OCTO_EINVAL, "octo_comm_create: bad rank / world");
    if (ctx->comm) return fail(ctx, OCTO_EINVAL, "octo_comm_create: this context already has a communicator");
    ctx->comm_rank = rank; ctx->comm_world = world;
    if (!unique_id128) {
        if (world == 1) return OCTO_OK;                   // nothing to exchange: the step is the swap kernel alone, RCCL is not loaded
        return fail(ctx, OCTO_EINVAL, "octo_comm_create: null unique id");
    }
    Rccl& r = rccl();
    if (!r.handle) return fail(ctx, OCTO_ENODEV, r.err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RcclUniqueId id;
    std::memcpy(id.internal, unique_id128, 128);
    RcclComm c = nullptr;
    const int rc = r.comm_init_rank(&c, world, id, rank);
    if (rc != 0) return fail(ctx, OCTO_EHIP, rccl_msg("ncclCommInitRank", rc));
    ctx->comm = c;
    return OCTO_OK;
}

int32_t octo_comm_destroy(octo_ctx* ctx) {
    if (!ctx) return OCTO_EINVAL;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        rccl().comm_destroy((RcclComm)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_world = 1; ctx->comm_rank = 0;
    return OCTO_OK;
}

int32_t octo_pt_step_device(octo_ctx* ctx, const double* d_ll_local, double* d_ll_all, const double* d_beta, int32_t* d_slot2rep,
                            int32_t n_temps, int64_t n_chains, int32_t parity, uint64_t seed, uint64_t step, int32_t* d_accepted,
                            void* hip_stream) {
    if (!ctx || !d_ll_local || !d_beta || !d_slot2rep) return fail(ctx, OCTO_EINVAL, "octo_pt_step_device: null argument");
    const int world = ctx->comm_world;
    if (n_temps < 2 || n_chains < 1 || n_temps % world) return fail(ctx, OCTO_EINVAL, "octo_pt_step_device: n_temps must be >= 2 and divide evenly over the ranks");
    const double* d_all = d_ll_local;
    if (world > 1 && !ctx->comm) return fail(ctx, OCTO_EINVAL, "octo_pt_step_device: call octo_comm_create first");
    if (ctx->comm) {      // (a one-rank communicator made with an id still goes through RCCL: the single-GPU test of this path)
        if (!d_ll_all) return fail(ctx, OCTO_EINVAL, "octo_pt_step_device: d_ll_all is required with a communicator");
        HIPCHK(ctx, hipSetDevice(ctx->device));
        hipStream_t st = hip_stream == OCTO_STREAM_CTX ? ctx->stream : (hipStream_t)hip_stream;
        // [replica][chain] blocks of contiguous replicas, rank after rank: exactly the layout k_pt_swap reads — no transpose
        const size_t count = (size_t)(n_temps / world) * (size_t)n_chains;
        const int rc = rccl().all_gather(d_ll_local, d_ll_all, count, RCCL_FLOAT64, (RcclComm)ctx->comm, st);
        if (rc != 0) return fail(ctx, OCTO_EHIP, rccl_msg("ncclAllGather", rc));
        d_all = d_ll_all;
    }
    return octo_pt_swap_device(ctx, d_all, d_beta, d_slot2rep, n_temps, n_chains, parity, seed, step, d_accepted, hip_stream);
}

// The same step on HOST arrays (round 6): what a driver without a device-array package calls — the reference's host language has
// its replicas' log-likelihoods in a Vector (ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:76-128 leaves the replicas to Pigeons.jl, one model(θ) at a
// time). ll_local [n_temps/world][n_chains], beta [n_temps], slot2rep [n_chains][n_temps] in/out, accepted [n_temps] in/out or NULL. Staged through
// the context's scratch on its own stream: three small copies in, the all-gather (world > 1) and the swap kernel, two copies out, one wait.
int32_t octo_pt_step(octo_ctx* ctx, const double* ll_local, const double* beta, int32_t* slot2rep, int32_t n_temps, int64_t n_chains, int32_t parity,
                     uint64_t seed, uint64_t step, int32_t* accepted) {
    if (!ctx || !ll_local || !beta || !slot2rep) return fail(ctx, OCTO_EINVAL, "octo_pt_step: null argument");
    const int world = ctx->comm_world;
    if (n_temps < 2 || n_chains < 1 || n_temps % world) return fail(ctx, OCTO_EINVAL, "octo_pt_step: n_temps must be >= 2 and divide evenly over the ranks");
    if (world > 1 && !ctx->comm) return fail(ctx, OCTO_EINVAL, "octo_pt_step: call octo_comm_create first");
    { int rcb = busy(ctx, "octo_pt_step"); if (rcb) return rcb; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n_loc = (int64_t)(n_temps / world) * n_chains, n_all = (int64_t)n_temps * n_chains;
    // one device block: [ll_local | ll_all | beta] doubles, then [slot2rep | accepted] int32 (in doubles' worth of space)
    const int64_t n_d = n_loc + n_all + n_temps, n_i = n_all + n_temps;
    int rc = grow(ctx, ctx->d_in, ctx->cap_in, n_d + (n_i + 1) / 2);
    if (rc) return rc;
    double* d_loc = ctx->d_in; double* d_all = d_loc + n_loc; double* d_beta = d_all + n_all;
    int32_t* d_s2r = reinterpret_cast<int32_t*>(d_beta + n_temps); int32_t* d_acc = d_s2r + n_all;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(d_loc, ll_local, sizeof(double) * (size_t)n_loc, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_beta, beta, sizeof(double) * (size_t)n_temps, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_s2r, slot2rep, sizeof(int32_t) * (size_t)n_all, hipMemcpyHostToDevice, st));
    if (accepted) HIPCHK(ctx, hipMemcpyAsync(d_acc, accepted, sizeof(int32_t) * (size_t)n_temps, hipMemcpyHostToDevice, st));
    else HIPCHK(ctx, hipMemsetAsync(d_acc, 0, sizeof(int32_t) * (size_t)n_temps, st));
    rc = octo_pt_step_device(ctx, d_loc, d_all, d_beta, d_s2r, n_temps, n_chains, parity, seed, step, d_acc, OCTO_STREAM_CTX);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(slot2rep, d_s2r, sizeof(int32_t) * (size_t)n_all, hipMemcpyDeviceToHost, st));
    if (accepted) HIPCHK(ctx, hipMemcpyAsync(accepted, d_acc, sizeof(int32_t) * (size_t)n_temps, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return OCTO_OK;
}

int32_t octo_eval_multi(octo_ctx* const* ctxs, const octo_dataset* const* dss, int32_t n_dev, const double* elems, const double* nuis,
                        int64_t ld, int64_t W, double* ll_out, double* g_elems, double* g_nuis) {
    if (!ctxs || !dss || n_dev < 1) return OCTO_EINVAL;
    for (int i = 0; i < n_dev; ++i)
        if (!ctxs[i] || !dss[i]) return OCTO_EINVAL;
    if (!elems || !ll_out) return fail(ctxs[0], OCTO_EINVAL, "octo_eval_multi: null argument");
    if (W < 0 || ld < W) return fail(ctxs[0], OCTO_EINVAL, "octo_eval_multi: need 0 <= W <= ld");
    // contiguous, balanced split: device i owns walkers [lo_i, hi_i); every device's copies and kernels are enqueued on its
    // own context stream before any of them is waited for, so the devices work concurrently from one host thread
    std::vector<int64_t> lo(n_dev + 1, 0);
    for (int i = 0; i < n_dev; ++i) lo[i + 1] = lo[i] + W / n_dev + (i < W % n_dev ? 1 : 0);
    int rc_first = OCTO_OK;
    std::vector<char> began(n_dev, 0);      // only a begin of OURS is ended: a context that refused (e.g. somebody else's begin is
                                             // still outstanding on it) keeps that other evaluation untouched
    for (int i = 0; i < n_dev; ++i) {
        const int64_t w0 = lo[i], n = lo[i + 1] - lo[i];
        if (n == 0) continue;
        const int rc = octo_eval_begin(ctxs[i], dss[i], elems + w0, nuis ? nuis + w0 : nullptr, ld, n, ll_out + w0, g_elems ? g_elems + w0 : nullptr,
                                       g_nuis ? g_nuis + w0 : nullptr);
        began[i] = rc == OCTO_OK;
        if (rc && !rc_first) {
            rc_first = rc;
            if (i > 0) ctxs[0]->err = "octo_eval_multi: device " + std::to_string(i) + ": " + ctxs[i]->err;      // the caller reads ctxs[0]'s message
        }
    }
    for (int i = 0; i < n_dev; ++i) {
        if (!began[i]) continue;
        const int rc = octo_eval_end(ctxs[i]);
        if (rc && !rc_first) {
            rc_first = rc;
            if (i > 0) ctxs[0]->err = "octo_eval_multi: device " + std::to_string(i) + ": " + ctxs[i]->err;
        }
    }
    return rc_first;
}

}  // extern "C"
