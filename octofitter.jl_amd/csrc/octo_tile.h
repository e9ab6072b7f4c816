// octo_tile.h — walker tiles made homogeneous for k_main's warm-started row loop (round 6).
//
// k_main's fallback from the warm start to the cold starter is WAVE-uniform, and so is the choice of the wave's step bound: one walker whose period is
// too short for the table's cadence keeps its 63 neighbours on the cold loop, and the lanes with e > 0.6 near their periastra — 0.1-1 % of their rows
// each — add up to 8 % of config 3's wave-rows when they are spread over every tile. Both losses go away when walkers that fail alike share a tile.
// k_tile_sort computes, per walker, the EXPECTED share p of its rows that fail the a-priori test (the share of the orbit, in mean anomaly, where
// 1/(1 − e cos E) >= thr: closed form in (e, ΔM)), buckets it (32 classes: never / 24 log-spaced classes / lanes that veto the step bound, by how
// much), and sorts the walkers of each SEGMENT (TILE_SEG) by bucket with a STABLE counting sort in LDS: one block of 1 024 threads per segment, one
// launch, no atomics — the permutation is a pure function of the inputs, so results stay bit-reproducible run to run. k_main<1, …, FUSED> reads its
// tile's elements and nuisances through `perm` (a gather of 9-12 values per walker per block), the partials stay in tile order, and k_finish writes
// ll and the adjoints back through `perm`: nothing outside the two kernels sees the order.
//
// It also prices itself: per segment the expected number of cold wave-rows per row, Σ_tiles (1 − Π_lanes (1 − p)), in the order given and in the
// sorted order (tools/warm_rates.py: the closed form agrees with a row-by-row simulation to the third digit). The host compares the difference
// × rows × the cost of a cold row with the cost of this launch (octo_api.hip: tile_decide) — config 3 gains 3.5 µs for a 4.5 µs launch and
// stays as drawn; a batch with a ~ LogU(0.3, 100) AU goes from every wave cold to a fifth of them.
#pragma once
#include "octo_kernels.h"

namespace octo {

#ifndef OCTO_TILE_WPT
#define OCTO_TILE_WPT 1
#endif
constexpr int TILE_WPT = OCTO_TILE_WPT;            // consecutive walkers per thread (1, 2 or 4): a tile of 64 = 64 / TILE_WPT consecutive threads
constexpr int TILE_TPB = 1024;                     // threads per block
constexpr int TILE_SEG = TILE_TPB * TILE_WPT;      // walkers per segment = per block. Measured / simulated (tools/warm_rates.py): segments of 1 024 leave
                                                   // 20.7 % of wide_prior's wave-rows cold, 2 048 20.0 %, 4 096 19.9 % (the whole batch: 19.9 %) — and the
                                                   // kernel, whose blocks each run on ONE CU, takes 8.9 µs at four walkers per thread
constexpr int TILE_K = 32;            // severity buckets

struct TileArgs {
    const double* elems;      // [9][ld]: a, e, …, M of the one planet
    int64_t ld, W;
    int32_t* perm;            // [W]: perm[sorted position] = walker
    float* stats;             // [2·n_segments] mapped pinned: expected cold wave-rows per row, as given | sorted
    float dm_ref;             // the dataset's reference step 2π·Δt [rad·day]: the preferred rung of its main table's ladder
    float inv_k_yr;
    int32_t planet;           // the planet whose severity is the key (0; two planets: the last one — the planet the two-planet kernels start warm)
    int32_t pad;
};

__device__ __forceinline__ float wave_excl_scan(float x, int lane, float& total) {
    float s = x;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const float y = __shfl_up(s, d, WAVE);
        if (lane >= d) s += y;
    }
    total = __shfl(s, WAVE - 1, WAVE);
    return s - x;
}

#ifdef OCTO_API_TU
static __global__ __launch_bounds__(TILE_TPB) void k_tile_sort(TileArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t cnt[TILE_K * TILE_TPB];      // [bucket][thread] counts, then exclusive prefixes within a group of 16 threads
    __shared__ uint16_t goff[TILE_K * WAVE];                                     // [bucket][group of 16 threads] exclusive prefix over the groups
    __shared__ uint16_t bbase[TILE_K + 1];
    __shared__ float slq[TILE_SEG];                                              // log2(1 − p) at the SORTED positions
    __shared__ float red[2 * (TILE_TPB / WAVE)];
    const int t = threadIdx.x, lane = t & (WAVE - 1), wv = t >> 6;
    const int64_t seg0 = (int64_t)blockIdx.x * TILE_SEG;
    const int n_seg = (int)(a.W - seg0 < TILE_SEG ? a.W - seg0 : TILE_SEG);
    {
        uint4* z = reinterpret_cast<uint4*>(cnt);
        z[t] = make_uint4(0, 0, 0, 0); z[t + TILE_TPB] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TILE_WPT; ++i) slq[TILE_WPT * t + i] = 0.0f;
    }
    // ---- severity of my walkers
    int bk[TILE_WPT]; float lq[TILE_WPT];
#pragma unroll
    for (int i = 0; i < TILE_WPT; ++i) {
        const int k = TILE_WPT * t + i;
        bk[i] = -1; lq[i] = 0.0f;
        if (k >= n_seg) continue;
        const int64_t w = seg0 + k;
        const double* el = a.elems + (int64_t)a.planet * OCTO_N_EL * a.ld + w;
        const double sma = el[(int64_t)OCTO_EL_A * a.ld], ecc = el[(int64_t)OCTO_EL_E * a.ld], Mt = el[(int64_t)OCTO_EL_M * a.ld];
        const float e = (float)ecc, af = (float)sma, mf = (float)Mt;
        const float invP = __builtin_amdgcn_rsqf(af * af * af * __builtin_amdgcn_rcpf(mf)) * a.inv_k_yr;      // FP32 throughout: a bucket is a factor 1.6 wide
        const float dM = fabsf(a.dm_ref * invP);
        int b; float p;
        if (!(dM <= WARM_DM_VETO) || !(e >= 0.0f) || !(e < 1.0f)) {      // vetoes every rung the reference step stands for (or invalid / NaN): last, fastest at the very end
            const float x = __builtin_amdgcn_logf(dM * (1.0f / WARM_DM_VETO)) * 2.0f;
            b = 25 + (x >= 0.0f ? (x < 6.0f ? (int)x : 6) : (x < 0.0f ? 0 : 6));
            p = 1.0f;
        } else {
            const float thr = __builtin_amdgcn_exp2f(0.2f * (__builtin_amdgcn_logf((float)WARM_TOL) - 3.0f * __builtin_amdgcn_logf(dM)));
            const float g = 1.0f - __builtin_amdgcn_rcpf(thr);
            if (e <= g) { b = 0; p = 0.0f; }
            else {
                const float c = g * __builtin_amdgcn_rcpf(e);      // in (1/2, 1): thr >= 2 here
                // acos(c) = √(1 − c)·(a0 + a1 c + a2 c² + a3 c³) on [0, 1], |error| <= 5e-5 rad (Abramowitz & Stegun 4.4.45); the error enters p through
                // (1 − e cos E0)·δE0, well inside a bucket
                const float E0 = __builtin_amdgcn_sqrtf(1.0f - c) * fmaf(fmaf(fmaf(-0.0187293f, c, 0.0742610f), c, -0.2121144f), c, 1.5707288f);
                const float s0 = __builtin_amdgcn_sqrtf(fmaxf(fmaf(-c, c, 1.0f), 0.0f));
                p = fminf(fmaxf((E0 - e * s0) * 0.318309886f, 0.0f), 1.0f);
                const float x = (__builtin_amdgcn_logf(fmaxf(p, 1e-30f)) + 17.0f) * (24.0f / 17.0f);
                b = 1 + (x > 0.0f ? (x < 23.0f ? (int)x : 23) : 0);
            }
        }
        bk[i] = b;
        lq[i] = p < 1.0f ? __builtin_amdgcn_logf(1.0f - p) : -INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TILE_WPT; ++i)
        if (bk[i] >= 0) cnt[bk[i] * TILE_TPB + t] += 1;      // my own byte: no other thread writes it
    __syncthreads();
    // ---- per bucket: exclusive prefix over the 1 024 thread counts = (prefix over the 64 groups of 16 threads) + (prefix within the group, written
    // back over the counts as bytes: a group holds at most 64 walkers). Wave w takes buckets 2w and 2w + 1; lane l takes group l.
#pragma unroll
    for (int q = 0; q < TILE_K / (TILE_TPB / WAVE); ++q) {
        const int b = wv * (TILE_K / (TILE_TPB / WAVE)) + q;
        uint4* grp = reinterpret_cast<uint4*>(cnt + b * TILE_TPB) + lane;
        uint4 v = *grp;
        uint32_t words[4] = {v.x, v.y, v.z, v.w};
        uint32_t run = 0;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            uint32_t outw = 0;
#pragma unroll
            for (int by = 0; by < 4; ++by) {
                const uint32_t c = (words[wd] >> (8 * by)) & 0xffu;
                outw |= run << (8 * by);
                run += c;
            }
            words[wd] = outw;
        }
        *grp = make_uint4(words[0], words[1], words[2], words[3]);
        float tot;
        const float off = wave_excl_scan((float)run, lane, tot);      // counts <= 4 096: exact in a float
        goff[b * WAVE + lane] = (uint16_t)off;
        if (lane == 0) bbase[b + 1] = (uint16_t)tot;                 // totals for now
    }
    __syncthreads();
    if (wv == 0) {      // bucket bases: one wave's scan of the 32 totals (a lone thread's loop is 32 dependent LDS round trips: 1.5 µs)
        static_assert(TILE_K <= WAVE, "one lane per bucket");
        const float c = lane < TILE_K ? (float)bbase[lane + 1] : 0.0f;
        float tot;
        const float off = wave_excl_scan(c, lane, tot);
        // (every lane has read its total before any lane writes: the scan's shuffles sit in between, and a wave executes in lock step)
        if (lane < TILE_K) bbase[lane] = (uint16_t)off;
    }
    __syncthreads();
    // ---- positions; perm; log2(1 − p) at the sorted positions
#pragma unroll
    for (int i = 0; i < TILE_WPT; ++i) {
        if (bk[i] < 0) continue;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < i; ++j) rank += (bk[j] == bk[i]) ? 1 : 0;
        const int pos = (int)bbase[bk[i]] + (int)goff[bk[i] * WAVE + (t >> 4)] + (int)cnt[bk[i] * TILE_TPB + t] + rank;
        a.perm[seg0 + pos] = (int32_t)(seg0 + TILE_WPT * t + i);
        slq[pos] = lq[i];
    }
    if (!a.stats) return;      // (block-uniform: a kernel argument) not a probe: the permutation is all that was asked for
    __syncthreads();
    // ---- expected cold wave-rows per row: Σ_tiles (1 − Π (1 − p)), as given and sorted. A tile = LPT consecutive threads.
    constexpr int LPT = WAVE / TILE_WPT;
    float s_id = 0.0f, s_so = 0.0f;
#pragma unroll
    for (int i = 0; i < TILE_WPT; ++i) { s_id += lq[i]; s_so += slq[TILE_WPT * t + i]; }
#pragma unroll
    for (int d = 1; d < LPT; d <<= 1) { s_id += __shfl_xor(s_id, d, WAVE); s_so += __shfl_xor(s_so, d, WAVE); }
    const bool head = (lane & (LPT - 1)) == 0 && TILE_WPT * t < n_seg;
    float c_id = head ? 1.0f - __builtin_amdgcn_exp2f(s_id) : 0.0f;
    float c_so = head ? 1.0f - __builtin_amdgcn_exp2f(s_so) : 0.0f;
#pragma unroll
    for (int d = LPT; d < WAVE; d <<= 1) { c_id += __shfl_xor(c_id, d, WAVE); c_so += __shfl_xor(c_so, d, WAVE); }
    if (lane == 0) { red[2 * wv] = c_id; red[2 * wv + 1] = c_so; }
    __syncthreads();
    if (t == 0) {
        float x = 0.0f, y = 0.0f;
        for (int k = 0; k < TILE_TPB / WAVE; ++k) { x += red[2 * k]; y += red[2 * k + 1]; }
        a.stats[2 * blockIdx.x] = x; a.stats[2 * blockIdx.x + 1] = y;
    }
}
#endif      // OCTO_API_TU

}  // namespace octo
