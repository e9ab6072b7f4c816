// Instruction-throughput microbenchmark for the VALU ops the likelihood kernel is made of (gfx950).
// Each kernel runs a long chain of 8 independent accumulators of ONE instruction; prints cycles per
// wave-instruction per SIMD at the measured clock. Development aid (hipcc --offload-arch=gfx950 tools/ubench.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, long long* cyc) {
    double x[NACC]; float f[NACC];
    for (int i = 0; i < NACC; ++i) { x[i] = a + threadIdx.x * 1e-9 + i; f[i] = (float)x[i]; }
    float2 p[NACC];
    for (int i = 0; i < NACC; ++i) p[i] = make_float2(f[i], f[i] + 1.f);
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) x[i] = fma(x[i], b, a);
            if (OP == 1) x[i] = x[i] * b;
            if (OP == 2) x[i] = x[i] + b;
            if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
            if (OP == 4) x[i] = rint(x[i] * b);
            if (OP == 5) f[i] = fmaf(f[i], (float)b, (float)a);
            if (OP == 6) f[i] = __builtin_amdgcn_rcpf(f[i]);
            if (OP == 7) f[i] = __builtin_amdgcn_sqrtf(f[i]);
            if (OP == 8) f[i] = __builtin_amdgcn_logf(f[i]);
            if (OP == 9) f[i] = __builtin_amdgcn_exp2f(f[i]);
            if (OP == 10) f[i] = (float)x[i] + f[i];           // cvt f64->f32 + add
            if (OP == 11) x[i] = (double)f[i] + x[i];          // cvt f32->f64 + add
            if (OP == 12) { p[i].x = fmaf(p[i].x, (float)b, (float)a); p[i].y = fmaf(p[i].y, (float)b, (float)a); }   // may become v_pk_fma_f32
            if (OP == 13) x[i] = __builtin_amdgcn_rsq(x[i]);
            if (OP == 14) x[i] = __builtin_amdgcn_fract(x[i] * b);
            if (OP == 15) x[i] = fabs(x[i]) > b ? x[i] : a;    // cmp + cndmask x2
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < NACC; ++i) s += x[i] + f[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP> void run(const char* name, int extra_per_iter) {
    double* d; long long* c; hipMalloc(&d, 8 * 256 * 2048); hipMalloc(&c, 8);
    const int blocks = 256 * 8;   // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1.0000001, 0.9999999, c); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 1.0000001, 0.9999999, c); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    // wave-instructions per SIMD = waves/SIMD * ITER * NACC ; waves per SIMD = blocks*4 waves / (256 CU * 4 SIMD) = 8
    const double winst = 8.0 * ITER * NACC;
    printf("%-28s %8.3f ms   wall-cycles(@clock64 ticks) %10lld   ns per wave-inst per SIMD %7.3f  (x2.4GHz = %6.2f cyc; extra ops/iter %d)\n",
           name, ms, cyc, ms * 1e6 / winst, ms * 1e6 / winst * 2.4, extra_per_iter);
    hipFree(d); hipFree(c);
}

int main() {
    run<0>("v_fma_f64", 0); run<1>("v_mul_f64", 0); run<2>("v_add_f64", 0); run<3>("v_rcp_f64", 0); run<13>("v_rsq_f64", 0);
    run<4>("v_mul_f64+v_rndne_f64", 1); run<14>("v_mul_f64+v_fract_f64", 1);
    run<5>("v_fma_f32", 0); run<12>("2x v_fma_f32 (pk?)", 1); run<6>("v_rcp_f32", 0); run<7>("v_sqrt_f32", 0); run<8>("v_log_f32", 0); run<9>("v_exp_f32", 0);
    run<10>("v_cvt_f32_f64+v_add_f32", 1); run<11>("v_cvt_f64_f32+v_add_f64", 1); run<15>("v_cmp_f64+2 cndmask", 2);
    return 0;
}
