// What does the HOST side of a synchronous three-kernel callback cost, and can a captured graph or a completion flag shorten it? A mid-size
// octo_model_logpost call (1 024 theta_t x 50 epochs: 43-45 µs by the host's clock) is k_model_fwd -> k_main -> k_finish, ~29 µs of GPU time, then a
// stream synchronisation. This probe replays that shape with clock-spinning kernels (12 / 8 / 9 µs) and prices, per call
// by the host's clock:
//   A  three launches + hipStreamSynchronize                       (what the library does above the fused small-batch limit)
//   B  three launches + the host spinning on a flag the last kernel stores to mapped pinned memory (what k_small's callers do)
//   C  ONE hipGraphLaunch of the captured three-kernel chain + hipStreamSynchronize
//   D  hipGraphLaunch + flag spin
//   E  one kernel of the summed duration + hipStreamSynchronize     (floor of a fully fused launch)
//   F  one kernel of the summed duration + flag spin
// hipcc --offload-arch=gfx950 -O2 -o graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(int ticks, volatile unsigned long long* flag, const unsigned long long* seq_src) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) {}
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) { __threadfence_system(); *flag = *seq_src; }
}
__global__ void k_bump(unsigned long long* seq) { if (threadIdx.x == 0) *seq += 1; }      // the sequence number lives on the device: a graph replays fixed arguments

int main() {
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* h_flag; unsigned long long* d_flag; unsigned long long* d_seq;
    CHK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *h_flag = 0;
    CHK(hipHostGetDevicePointer((void**)&d_flag, h_flag, 0));
    CHK(hipMalloc(&d_seq, 8)); CHK(hipMemset(d_seq, 0, 8));
    const int t1 = 1200, t2 = 800, t3 = 900;      // 100 MHz clock
    unsigned long long seq = 0;
    auto three = [&](bool flag) {
        hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d_seq);      // (stands for nothing in the library; kept in every variant so they stay comparable: +~2 µs)
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(1024), 0, st, t1, (volatile unsigned long long*)nullptr, (const unsigned long long*)d_seq);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, t2, (volatile unsigned long long*)nullptr, (const unsigned long long*)d_seq);
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(1024), 0, st, t3, flag ? (volatile unsigned long long*)d_flag : nullptr, (const unsigned long long*)d_seq);
    };
    auto one = [&](bool flag) {
        hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, st, d_seq);
        hipLaunchKernelGGL(k_spin, dim3(16), dim3(1024), 0, st, t1 + t2 + t3, flag ? (volatile unsigned long long*)d_flag : nullptr, (const unsigned long long*)d_seq);
    };
    hipGraph_t g; hipGraphExec_t ge[2];
    for (int f = 0; f < 2; ++f) {
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        three(f == 1);
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge[f], g, nullptr, nullptr, 0));
        CHK(hipGraphDestroy(g));
    }
    auto wait_flag = [&]() { ++seq; while (*(volatile unsigned long long*)h_flag != seq) {} };
    auto timeit = [&](auto&& body, const char* what) {
        for (int i = 0; i < 100; ++i) body();
        std::vector<double> v;
        for (int rep = 0; rep < 7; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; ++i) body();
            v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 300);
        }
        std::sort(v.begin(), v.end());
        printf("%-90s %7.2f us per call (best of 7 x 300; median %.2f)\n", what, v[0], v[3]);
    };
    timeit([&] { three(false); ++seq; hipStreamSynchronize(st); }, "A  three launches + hipStreamSynchronize");
    timeit([&] { three(true); wait_flag(); }, "B  three launches + flag spin");
    timeit([&] { hipGraphLaunch(ge[0], st); ++seq; hipStreamSynchronize(st); }, "C  graph launch + hipStreamSynchronize");
    timeit([&] { hipGraphLaunch(ge[1], st); wait_flag(); }, "D  graph launch + flag spin");
    timeit([&] { one(false); ++seq; hipStreamSynchronize(st); }, "E  one kernel of the summed duration + hipStreamSynchronize");
    timeit([&] { one(true); wait_flag(); }, "F  one kernel of the summed duration + flag spin");
    CHK(hipStreamSynchronize(st));
    printf("# GPU time of the chain: %d + %d + %d ticks of the 100 MHz clock = %.0f us (+ k_bump)\n", t1, t2, t3, (t1 + t2 + t3) / 100.0);
    return 0;
}
