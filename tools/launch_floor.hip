// Floor of "one launch + one synchronisation" on this box, the cost every small-batch call pays whatever the kernel does:
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_floor.bin tools/launch_floor.hip && tools/launch_floor.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_flag(volatile double* out, const double* in) { out[0] = in[0] + 1.0; }
__global__ void k_spin(volatile double* out, const double* in, int n) {
    double x = in[0];
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
    out[0] = x;
}
// shader clock during a short, isolated kernel: s_memtime (core clock) against s_memrealtime (100 MHz)
__global__ void k_clock(volatile double* out, const double* in, int n) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    double x = in[0];
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[0] = x; out[1] = (double)(c1 - c0); out[2] = (double)(r1 - r0);
}
template <class F> double bench(F f, int n = 5000) {
    for (int i = 0; i < 500; ++i) f();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *h, *d;
    hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent); hipMalloc((void**)&d, 4096);
    h[0] = 1.0;
    printf("empty kernel + hipStreamSynchronize            %6.2f us\n", bench([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipStreamSynchronize(st); }));
    printf("empty kernel x3 + hipStreamSynchronize         %6.2f us\n", bench([&] { for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipStreamSynchronize(st); }));
    printf("kernel reading+writing pinned host + sync      %6.2f us\n", bench([&] { hipLaunchKernelGGL(k_flag, 1, 64, 0, st, h + 8, h); hipStreamSynchronize(st); }));
    printf("H2D 512 B + kernel + D2H 512 B + sync          %6.2f us\n", bench([&] { hipMemcpyAsync(d, h, 512, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_flag, 1, 64, 0, st, d + 8, d); hipMemcpyAsync(h + 16, d + 8, 512, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }));
    for (int n : {1000, 4000, 16000}) printf("pinned kernel with %5d dependent FP64 FMAs + sync %6.2f us\n", n, bench([&] { hipLaunchKernelGGL(k_spin, 1, 64, 0, st, h + 8, h, n); hipStreamSynchronize(st); }));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_clock, 1, 64, 0, st, h + 8, h, 4000); hipStreamSynchronize(st);
        printf("isolated kernel, 4000 dependent FMAs: %.0f core cycles (%.2f per FMA), %.1f us by the 100 MHz clock -> %.0f MHz\n", h[9], h[9] / 4000, h[10] / 100.0, h[9] / (h[10] / 100.0));
    }
    // back-to-back for 0.2 s, then the same measurement: does the clock ramp?
    for (int i = 0; i < 20000; ++i) { hipLaunchKernelGGL(k_spin, 1, 64, 0, st, h + 8, h, 1000); hipStreamSynchronize(st); }
    hipLaunchKernelGGL(k_clock, 1, 64, 0, st, h + 8, h, 4000); hipStreamSynchronize(st);
    printf("after 20000 back-to-back small launches: %.2f cycles per FMA, %.0f MHz\n", h[9] / 4000, h[9] / (h[10] / 100.0));
    // spin on a flag in pinned memory instead of hipStreamSynchronize
    printf("kernel + host spin on pinned flag (no sync call)  %6.2f us\n", bench([&] { volatile double* f = h + 8; f[0] = 0.0; hipLaunchKernelGGL(k_flag, 1, 64, 0, st, h + 8, h); while (f[0] == 0.0) {} }));
    hipStreamSynchronize(st);
    return 0;
}
