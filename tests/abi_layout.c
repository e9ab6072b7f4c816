/* abi_layout.c — a compiled-C consumer of include/octofitter_hip.h (test infrastructure).
 *   gcc -std=c11 -I include -o abi_layout tests/abi_layout.c -ldl
 *   ./abi_layout layout                 prints sizeof / offsetof of every struct of the header as JSON: tests/test_abi.py holds the
 *                                       ctypes mirror (host/capi.py) and the Julia mirror (julia/OctofitterHIP.jl) to it
 *   ./abi_layout symbols  <lib.so>      dlopen + dlsym of every function the header declares
 *   ./abi_layout eval     <lib.so>      one octo_eval call through plain C (needs a GPU): 3 RA/Dec epochs, 2 walkers, checks that the
 *                                       second walker (e = 1.5) comes back -Inf with zero gradient and the first one finite
 * The static asserts pin what the kernels rely on: 8-byte doubles, no padding surprises. */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "octofitter_hip.h"

_Static_assert(sizeof(double) == 8 && sizeof(void*) == 8, "LP64 with IEEE doubles");
_Static_assert(sizeof(octo_consts) == 7 * 8, "octo_consts is seven doubles");
_Static_assert(sizeof(octo_planet_desc) == 8, "octo_planet_desc is two int32");
_Static_assert(sizeof(octo_prior) == 40 && offsetof(octo_prior, p0) == 8, "octo_prior: {int32 kind, pad; double p0, p1, lo, hi}");
_Static_assert(sizeof(octo_source) == 24 && offsetof(octo_source, value) == 16, "octo_source: {int32 kind, i0, i1, flags; double value}");
_Static_assert(sizeof(octo_obs_desc) == 80 && offsetof(octo_obs_desc, epoch) == 16 && offsetof(octo_obs_desc, n_extra) == 72, "octo_obs_desc layout");

#define F(S, f) printf("%s[\"%s\", %zu, %zu]", first ? "" : ", ", #f, offsetof(S, f), sizeof(((S*)0)->f)), first = 0
#define BEGIN(S) printf("%s\"%s\": {\"size\": %zu, \"fields\": [", firsts ? "" : ", ", #S, sizeof(S)), firsts = 0, first = 1
#define END() printf("]}")

static const char* SYMBOLS[] = {
    "octo_consts_default", "octo_version", "octo_ctx_create", "octo_ctx_destroy", "octo_consts_set", "octo_ctx_set_small_batch", "octo_last_error",
    "octo_dataset_create", "octo_dataset_destroy", "octo_dataset_n_rows", "octo_eval", "octo_eval_begin", "octo_eval_end", "octo_eval_multi",
    "octo_eval_device", "octo_sync", "octo_kepler_solve", "octo_ofti_create", "octo_ofti_destroy", "octo_ofti_eval", "octo_ofti_eval_device",
    "octo_model_create", "octo_model_destroy", "octo_model_logpost", "octo_model_logpost_device", "octo_timing_enable", "octo_timing_read",
    "octo_timing_stats", "octo_pt_swap_device", "octo_comm_unique_id", "octo_comm_create", "octo_comm_destroy", "octo_pt_step_device", "octo_host_register", "octo_host_unregister", "octo_kepler_solve_table",
    "octo_ctx_set_option", "octo_ctx_get_option", "octo_pt_step"};

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "layout";
    if (!strcmp(mode, "layout")) {
        int first = 1, firsts = 1;
        printf("{");
        BEGIN(octo_consts); F(octo_consts, kepler_year_to_julian_day); F(octo_consts, year2day_julian); F(octo_consts, au2m); F(octo_consts, sec2year_julian);
        F(octo_consts, pc2au); F(octo_consts, rad2as); F(octo_consts, mjup2msol); END();
        BEGIN(octo_obs_desc); F(octo_obs_desc, kind); F(octo_obs_desc, planet); F(octo_obs_desc, n_epochs); F(octo_obs_desc, epoch); F(octo_obs_desc, y1);
        F(octo_obs_desc, y2); F(octo_obs_desc, s1); F(octo_obs_desc, s2); F(octo_obs_desc, cor); F(octo_obs_desc, extra); F(octo_obs_desc, n_extra); END();
        BEGIN(octo_planet_desc); F(octo_planet_desc, orbit_kind); F(octo_planet_desc, has_mass); END();
        BEGIN(octo_prior); F(octo_prior, kind); F(octo_prior, pad); F(octo_prior, p0); F(octo_prior, p1); F(octo_prior, lo); F(octo_prior, hi); END();
        BEGIN(octo_source); F(octo_source, kind); F(octo_source, i0); F(octo_source, i1); F(octo_source, flags); F(octo_source, value); END();
        printf(", \"n_symbols\": %zu, \"OCTO_N_EL\": %d, \"OCTO_N_NUIS\": %d, \"OCTO_SMALL_BATCH_MAX\": %d}\n", sizeof(SYMBOLS) / sizeof(SYMBOLS[0]), OCTO_N_EL, OCTO_N_NUIS,
               OCTO_SMALL_BATCH_MAX);
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: abi_layout symbols|eval <liboctofitter_hip.so>\n"); return 2; }
    void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    for (size_t k = 0; k < sizeof(SYMBOLS) / sizeof(SYMBOLS[0]); ++k)
        if (!dlsym(h, SYMBOLS[k])) { fprintf(stderr, "missing symbol %s\n", SYMBOLS[k]); return 4; }
    if (!strcmp(mode, "symbols")) { printf("{\"symbols_ok\": %zu}\n", sizeof(SYMBOLS) / sizeof(SYMBOLS[0])); return 0; }

    /* ---- one evaluation through plain C */
    int32_t (*ctx_create)(octo_ctx**, int32_t) = (int32_t(*)(octo_ctx**, int32_t))dlsym(h, "octo_ctx_create");
    int32_t (*ctx_destroy)(octo_ctx*) = (int32_t(*)(octo_ctx*))dlsym(h, "octo_ctx_destroy");
    int32_t (*ds_create)(octo_ctx*, const octo_obs_desc*, int32_t, const octo_planet_desc*, int32_t, octo_dataset**) =
        (int32_t(*)(octo_ctx*, const octo_obs_desc*, int32_t, const octo_planet_desc*, int32_t, octo_dataset**))dlsym(h, "octo_dataset_create");
    int32_t (*ds_destroy)(octo_dataset*) = (int32_t(*)(octo_dataset*))dlsym(h, "octo_dataset_destroy");
    int32_t (*eval)(octo_ctx*, const octo_dataset*, const double*, const double*, int64_t, int64_t, double*, double*, double*) =
        (int32_t(*)(octo_ctx*, const octo_dataset*, const double*, const double*, int64_t, int64_t, double*, double*, double*))dlsym(h, "octo_eval");
    const char* (*last_error)(const octo_ctx*) = (const char* (*)(const octo_ctx*))dlsym(h, "octo_last_error");
    octo_ctx* ctx = NULL;
    int32_t st = ctx_create(&ctx, 0);
    if (st != OCTO_OK) { printf("{\"ctx_status\": %d}\n", st); return st == OCTO_ENODEV ? 10 : 5; }
    double epoch[3] = {50000.0, 50120.0, 50240.0}, ra[3] = {-505.76, -502.57, -498.21}, dec[3] = {-66.93, -37.47, -7.93}, s[3] = {10.0, 10.0, 10.0};
    octo_obs_desc ob;
    memset(&ob, 0, sizeof(ob));
    ob.kind = OCTO_ASTROM_RADEC; ob.planet = 0; ob.n_epochs = 3; ob.epoch = epoch; ob.y1 = ra; ob.y2 = dec; ob.s1 = s; ob.s2 = s;
    octo_planet_desc pl = {OCTO_ORBIT_VISUAL_KEP, 0};
    octo_dataset* ds = NULL;
    st = ds_create(ctx, &ob, 1, &pl, 1, &ds);
    if (st != OCTO_OK) { fprintf(stderr, "octo_dataset_create: %s\n", last_error(ctx)); return 6; }
    /* elems[(k) * ld + w], ld = W = 2: a, e, i, ω, Ω, tp, M, plx, mass */
    double el[OCTO_N_EL][2] = {{12.0, 12.0}, {0.11, 1.5}, {0.7, 0.7}, {0.66, 0.66}, {0.28, 0.28}, {41479.0, 41479.0}, {1.2, 1.2}, {50.0, 50.0}, {0.0, 0.0}};
    double ll[2] = {0, 0}, g[OCTO_N_EL][2];
    st = eval(ctx, ds, &el[0][0], NULL, 2, 2, ll, &g[0][0], NULL);
    if (st != OCTO_OK) { fprintf(stderr, "octo_eval: %s\n", last_error(ctx)); return 7; }
    int ok = isfinite(ll[0]) && isinf(ll[1]) && ll[1] < 0;
    for (int k = 0; k < OCTO_N_EL; ++k) ok = ok && g[k][1] == 0.0 && isfinite(g[k][0]);
    printf("{\"ll0\": %.17g, \"ll1_is_minus_inf\": %d, \"g_a\": %.17g, \"ok\": %d}\n", ll[0], isinf(ll[1]) && ll[1] < 0, g[0][0], ok);
    ds_destroy(ds); ctx_destroy(ctx);
    return ok ? 0 : 8;
}
