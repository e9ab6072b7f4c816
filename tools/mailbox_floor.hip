// Round trip of a PERSISTENT kernel serving requests through a mailbox in mapped pinned host memory (no launch per request), against the
// launch + flag floor of tools/launch_floor.hip: what a resident "server" block would save a one-θ-per-call sampler.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mailbox_floor.bin tools/mailbox_floor.hip && tools/mailbox_floor.bin
// Protocol: request = 8 cache lines of {seq, 7 payload doubles} (the sequence number in EVERY 64-byte line: a line is snooped as one
// coherent snapshot, so a line that shows the new seq shows the new payload); reply = payload sum + the seq as the done flag.
// The kernel leaves by itself after `idle_ticks` of the 100 MHz clock without a request (a crashed host cannot leave it spinning).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <immintrin.h>
constexpr int LINES = 8;
__global__ void k_server(const uint64_t* box, uint64_t* reply, uint64_t* exited, unsigned long long idle_ticks, int n_fma) {
    const int lane = threadIdx.x;
    uint64_t served = 0;
    unsigned long long t_last = wall_clock64();
    for (;;) {
        const uint64_t word = __hip_atomic_load(box + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t seq = __shfl(word, 0);
        const bool line_ok = (lane % 8 != 0) || word == seq;
        if (seq != served && __all(line_ok)) {
            double x = (lane % 8 != 0) ? __longlong_as_double((long long)word) : 0.0;
            for (int i = 0; i < n_fma; ++i) x = x * 1.0000001 + 1e-9;      // stand-in for the evaluation
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if (lane == 0) {
                reply[1] = (uint64_t)__double_as_longlong(x);
                __hip_atomic_store(reply, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            served = seq;
            t_last = wall_clock64();
        } else if (wall_clock64() - t_last > idle_ticks) {
            if (lane == 0) __hip_atomic_store(exited, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}
__global__ void k_flag(volatile double* out, const double* in, int n_fma) {
    double x = in[threadIdx.x % 8 ? threadIdx.x : 1];
    for (int i = 0; i < n_fma; ++i) x = x * 1.0000001 + 1e-9;
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (threadIdx.x == 0) { out[1] = x; __threadfence_system(); out[0] = in[0]; }
}
int main() {
    hipStream_t st, st2; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
    uint64_t* h;
    hipHostMalloc((void**)&h, 8192, hipHostMallocMapped | hipHostMallocCoherent);
    volatile uint64_t* box = h; volatile uint64_t* reply = h + 256; volatile uint64_t* exited = h + 512;
    for (int i = 0; i < 1024; ++i) h[i] = 0;
    for (int n_fma : {0, 1000, 4000}) {
        *exited = 0;
        hipLaunchKernelGGL(k_server, 1, 64, 0, st, (const uint64_t*)h, h + 256, h + 512, 100ull * 20000 /* 20 ms */, n_fma);
        uint64_t seq = 0;
        auto post = [&] {
            ++seq;
            for (int l = 0; l < LINES; ++l) {
                for (int k = 1; k < 8; ++k) { double v = 1.0 + l + 0.1 * k; box[l * 8 + k] = *(uint64_t*)&v; }
                box[l * 8] = seq;
            }
            _mm_sfence();
            while (reply[0] != seq) { if (*exited) { printf("server exited early\n"); return; } }
        };
        for (int i = 0; i < 2000; ++i) post();
        auto t0 = std::chrono::steady_clock::now();
        const int n = 20000;
        for (int i = 0; i < n; ++i) post();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("persistent server, %5d dependent FMAs per request: %6.2f us per round trip\n", n_fma, us);
        while (!*exited) {}      // idle timeout
        hipStreamSynchronize(st);
        // the same work as one launch + flag per request
        double* hd = (double*)(h + 600);
        for (int i = 0; i < 64; ++i) hd[i] = 1.0 + i;
        volatile double* out = (double*)(h + 700);
        auto launch = [&] { hd[0] += 1.0; hipLaunchKernelGGL(k_flag, 1, 64, 0, st2, (double*)(h + 700), hd, n_fma); while (out[0] != hd[0]) {} };
        for (int i = 0; i < 500; ++i) launch();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 5000; ++i) launch();
        printf("launch + flag per request, %5d dependent FMAs:        %6.2f us\n", n_fma, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5000);
        hipStreamSynchronize(st2);
    }
    return 0;
}
