// Can the host write device memory directly (large BAR)? hipExtMallocWithFlags(hipDeviceMallocFinegrained) + CPU stores + kernel read.
// Development probe for the small-batch staging path:  hipcc --offload-arch=gfx950 -O2 -o /tmp/bar tools/bar_write_test.hip && /tmp/bar
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
__global__ void k_sum(const double* p, int n, double* out, volatile unsigned long* flag, unsigned long seq) {
    double s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) { *out = s; __threadfence_system(); *flag = seq; }
}
int main() {
    double* d = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s ptr=%p\n", hipGetErrorString(e), (void*)d);
    if (e != hipSuccess) return 1;
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    e = hipPointerGetAttributes(&at, d);
    printf("attrs: %s type=%d host=%p dev=%p managed=%d\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer, at.isManaged);
    double *h_out; unsigned long* h_flag;
    hipHostMalloc((void**)&h_out, 64, hipHostMallocMapped); hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped);
    double *m_in; hipHostMalloc((void**)&m_in, 4096, hipHostMallocMapped);
    *h_flag = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    fflush(stdout);
    for (int mode = 0; mode < 2; ++mode) {
        double* src = mode ? d : m_in;
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            const int N = 2000;
            for (int it = 1; it <= N; ++it) {
                for (int i = 0; i < 64; ++i) src[i] = it + i;       // CPU stores (mode 1: straight into device memory)
                _mm_sfence();
                unsigned long seq = (unsigned long)(mode * 100000 + rep * 10000 + it);
                hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, st, src, 64, h_out, h_flag, seq);
                while (*(volatile unsigned long*)h_flag != seq) { }
                double want = 64.0 * it + 2016.0;
                if (*h_out != want) { printf("MISMATCH mode %d it %d got %f want %f\n", mode, it, *h_out, want); return 2; }
            }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (us < best) best = us;
        }
        printf("%s: %.2f us per (write 512 B, launch, flag)\n", mode ? "inputs in DEVICE memory written by the host (BAR)" : "inputs in mapped pinned HOST memory", best);
        fflush(stdout);
    }
    // host store bandwidth into each kind of memory (is the BAR mapping write-combining?)
    double* big_d = nullptr; double* big_h = nullptr;
    if (hipExtMallocWithFlags((void**)&big_d, 1 << 20, hipDeviceMallocFinegrained) != hipSuccess) return 3;
    hipHostMalloc((void**)&big_h, 1 << 20, hipHostMallocMapped);
    static double srcbuf[1 << 17];
    for (int i = 0; i < (1 << 17); ++i) srcbuf[i] = i;
    for (size_t bytes : {512ul, 4096ul, 16384ul, 131072ul, 1048576ul}) {
        for (int mode = 0; mode < 2; ++mode) {
            double* dst = mode ? big_d : big_h;
            const int N = bytes <= 16384 ? 2000 : 100;
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < N; ++it) { memcpy(dst, srcbuf, bytes); _mm_sfence(); }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("host memcpy of %7zu B into %s: %8.2f us (%.2f GB/s)\n", bytes, mode ? "device memory (BAR)" : "pinned host memory ", us, bytes / us * 1e-3);
        }
    }
    return 0;
}
