/* latency.c — per-call latency of the small-batch path from PLAIN C through the C ABI: what a Julia `ccall` pays (tools/latency_w1.py
 * measures the same calls through ctypes, which adds ~1-2 us of argument marshalling per call).
 *   gcc -O2 -std=c11 -I include -o tools/latency.bin tools/latency.c -ldl -lm && tools/latency.bin octofitter.jl_amd/lib/liboctofitter_hip.so
 * Synthetic table: E RA/Dec epochs of a fixed orbit + deterministic pseudo-noise; W parameter sets near it. */
#define _POSIX_C_SOURCE 199309L
#define _DEFAULT_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "octofitter_hip.h"

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: latency.bin <liboctofitter_hip.so>\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    int32_t (*ctx_create)(octo_ctx**, int32_t) = (int32_t(*)(octo_ctx**, int32_t))dlsym(h, "octo_ctx_create");
    int32_t (*ctx_destroy)(octo_ctx*) = (int32_t(*)(octo_ctx*))dlsym(h, "octo_ctx_destroy");
    int32_t (*ds_create)(octo_ctx*, const octo_obs_desc*, int32_t, const octo_planet_desc*, int32_t, octo_dataset**) =
        (int32_t(*)(octo_ctx*, const octo_obs_desc*, int32_t, const octo_planet_desc*, int32_t, octo_dataset**))dlsym(h, "octo_dataset_create");
    int32_t (*ds_destroy)(octo_dataset*) = (int32_t(*)(octo_dataset*))dlsym(h, "octo_dataset_destroy");
    int32_t (*eval)(octo_ctx*, const octo_dataset*, const double*, const double*, int64_t, int64_t, double*, double*, double*) =
        (int32_t(*)(octo_ctx*, const octo_dataset*, const double*, const double*, int64_t, int64_t, double*, double*, double*))dlsym(h, "octo_eval");
    octo_ctx* ctx = NULL;
    if (ctx_create(&ctx, 0) != OCTO_OK) { fprintf(stderr, "no device\n"); return 10; }
    const int Es[2] = {50, 10000}, Ws[2] = {1, 32};
    for (int ie = 0; ie < 2; ++ie) {
        const int E = Es[ie];
        double *ep = malloc(sizeof(double) * E), *ra = malloc(sizeof(double) * E), *dec = malloc(sizeof(double) * E), *s = malloc(sizeof(double) * E);
        for (int j = 0; j < E; ++j) {
            ep[j] = 50000.0 + j; const double ph = 2 * M_PI * j / 9000.0;
            ra[j] = 400 * cos(ph) + 7.0 * sin(12.9898 * j); dec[j] = 300 * sin(ph) + 7.0 * cos(78.233 * j); s[j] = 10.0;
        }
        octo_obs_desc ob; memset(&ob, 0, sizeof(ob));
        ob.kind = OCTO_ASTROM_RADEC; ob.planet = 0; ob.n_epochs = E; ob.epoch = ep; ob.y1 = ra; ob.y2 = dec; ob.s1 = s; ob.s2 = s;
        octo_planet_desc pl = {OCTO_ORBIT_VISUAL_KEP, 0};
        octo_dataset* ds = NULL;
        if (ds_create(ctx, &ob, 1, &pl, 1, &ds) != OCTO_OK) return 6;
        for (int iw = 0; iw < 2; ++iw) {
            const int W = Ws[iw];
            double* el = malloc(sizeof(double) * OCTO_N_EL * W); double* ll = malloc(sizeof(double) * W); double* g = malloc(sizeof(double) * OCTO_N_EL * W);
            const double base[OCTO_N_EL] = {10.0, 0.3, 1.0, 0.5, 2.0, 50000.0, 1.2, 50.0, 0.0};
            for (int k = 0; k < OCTO_N_EL; ++k) for (int w = 0; w < W; ++w) el[k * W + w] = base[k] * (1.0 + 0.001 * w);
            for (int grad = 0; grad < 2; ++grad) {
                for (int i = 0; i < 500; ++i) eval(ctx, ds, el, NULL, W, W, ll, grad ? g : NULL, NULL);
                double best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    const double t0 = now_us();
                    for (int i = 0; i < 1000; ++i) eval(ctx, ds, el, NULL, W, W, ll, grad ? g : NULL, NULL);
                    const double dt = (now_us() - t0) / 1000.0;
                    if (dt < best) best = dt;
                }
                printf("octo_eval from C: E=%5d W=%2d grad=%d  %6.2f us per call   (ll[0] = %.6f)\n", E, W, grad, best, ll[0]);
            }
            free(el); free(ll); free(g);
        }
        ds_destroy(ds); free(ep); free(ra); free(dec); free(s);
    }
    ctx_destroy(ctx);
    return 0;
}
