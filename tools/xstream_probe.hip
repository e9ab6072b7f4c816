// What does a cross-stream hand-off cost? VERDICT r4 item 6 proposes k_finish on a SECOND stream, resident and spinning on per-tile counters while k_main
// (on the caller's stream) is still running, so that only the partial reads stay on the critical path of a one-round launch (1 250 walkers: k_main 46.5 µs +
// k_finish and boundaries 4.5 µs). Per evaluation that design needs two event edges (caller's stream -> finisher's stream at the start: the previous results
// must have been consumed; finisher's stream -> caller's stream at the end) and a device-side spin. This probe prices the pieces on the box:
//   A  two dependent kernels back to back on ONE stream (what the library does now): per pair
//   B  the same two kernels, the second on another stream behind an event, and the first stream waiting for it (the two event edges): per pair
//   C  the second kernel launched FIRST on the other stream, spinning on a flag in device memory that the first kernel's last block releases (agent scope),
//      then the edge back: per pair — the proposed design with everything but the finish itself
// Kernel 1 runs ~45 µs (one block per CU, clock spin), kernel 2 is 20 blocks that do nothing but (C) wait.   hipcc --offload-arch=gfx950 -O2 -o xstream tools/xstream_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(int ticks, unsigned* counter, unsigned* flag, unsigned n_blocks, unsigned seq) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) {}
    if (flag && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(counter, 1u) == n_blocks - 1) { *counter = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
}
__global__ void k_fin(const unsigned* flag, unsigned seq, unsigned* out) {
    if (flag && threadIdx.x == 0) {
        unsigned long long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq && ++spins < (1ull << 26)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = seq;
}

int main() {
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    unsigned *counter, *flag, *out;
    CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&flag, 4)); CHK(hipMalloc(&out, 4 * 64));
    CHK(hipMemset(counter, 0, 4)); CHK(hipMemset(flag, 0, 4));
    const int ticks = 4500;      // 100 MHz wall clock: 45 µs
    const unsigned nb = 256;
    auto timeit = [&](auto&& body, const char* what) {
        for (int i = 0; i < 50; ++i) body(i + 1);
        hipDeviceSynchronize();
        std::vector<double> v;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 200; ++i) body(1000 * (rep + 1) + i);
            hipDeviceSynchronize();
            v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
        }
        std::sort(v.begin(), v.end());
        printf("%-110s %7.2f us per pair (best of 5 x 200; median %.2f)\n", what, v[0], v[2]);
    };
    timeit([&](unsigned) { hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, ticks, counter, (unsigned*)nullptr, nb, 0u);
                           hipLaunchKernelGGL(k_fin, dim3(20), dim3(1024), 0, s1, (const unsigned*)nullptr, 0u, out); },
           "A  k_work -> k_fin on one stream");
    timeit([&](unsigned) { hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, ticks, counter, (unsigned*)nullptr, nb, 0u);
                           hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
                           hipLaunchKernelGGL(k_fin, dim3(20), dim3(1024), 0, s2, (const unsigned*)nullptr, 0u, out);
                           hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); },
           "B  k_work on s1, event, k_fin on s2, event back to s1");
    timeit([&](unsigned seq) { hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
                               hipLaunchKernelGGL(k_fin, dim3(20), dim3(1024), 0, s2, (const unsigned*)flag, seq, out);
                               hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, ticks, counter, flag, nb, seq);
                               hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); },
           "C  k_fin resident on s2 spinning on a flag, k_work on s1 releases it, event back to s1 (the proposed design)");
    timeit([&](unsigned) { hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, ticks, counter, (unsigned*)nullptr, nb, 0u); },
           "   k_work alone");
    return 0;
}
