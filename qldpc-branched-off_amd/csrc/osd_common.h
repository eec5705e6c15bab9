// Shared pieces of the OSD-0 kernels (gf2.hip, osd_fwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

// Phase timers of the OSD kernels exist only in the diagnostic build (`make timers` -> libqldpc_hip_timers.so, -DQLDPC_OSD_TIMERS);
// the default build carries no clock reads.  Counters (uint64[16]): [0] shots, [1] chunks, [2] columns taken into blocks, [3] pivots,
// [4] cycles, [5] kill passes, [6] blocks, [8] sort, [9] phase 1 (reduce columns), [10] phase 2 (block pivots), [11] phase 3 (row updates),
// [12] dependent-column tests, [13] back-substitution.
#ifdef QLDPC_OSD_TIMERS
#define OSD_CLOCK() clock64()
#else
#define OSD_CLOCK() 0ll
#endif

namespace qldpc {
unsigned long long *osd_timer_buffer();      // device buffer of the current device, NULL in the default build

__device__ __forceinline__ unsigned long long osd_key(double x) {
    double a = fabs(x);
    if (a != a) a = INFINITY;
    return (unsigned long long)__double_as_longlong(a);          // non-negative doubles order like their bit patterns
}

// Column order of one shot: ascending |llr| (reference src/decoding/osd.py:11-12), ties by ascending index.
// Stable LSD radix sort of the column indices by the 64-bit key, 8 passes of 8 bits (a bitonic network of (key, index) pairs cost
// 0.44 M cycles per shot, all of it compare-exchange instructions).  The keys stay where they are; a pass permutes the index array
// only.  Stability -- which is what makes ties come out in ascending index order, the permutation starting as the identity -- comes
// from (a) every wave owning a contiguous range of positions and walking it in order, (b) a lane's rank among the lanes of its step
// with the same digit (8 ballots), (c) an exclusive scan of the [digit][wave] counters in digit-major order.
// keys [n], pa / pb [n] each, cnt [256][waves] + [waves] are scratch (LDS or global); the order is left in ordw [n].  Whole workgroup.
__device__ __forceinline__ void osd_radix_sort(const double *__restrict__ llr, int n, unsigned long long *keys, uint16_t *pa, uint16_t *pb,
                                               unsigned *cnt, uint16_t *ordw) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int NW = T >> 6, wv = tid >> 6, lane = tid & 63;
    unsigned *wsum = cnt + 256 * NW;
    const int span = (((n + NW - 1) / NW) + 63) & ~63;                                      // positions per wave, a multiple of 64
    const int wbeg = wv * span, wend = min(n, wbeg + span);
    for (int j = tid; j < n; j += T) { keys[j] = osd_key(llr[j]); pa[j] = (uint16_t)j; }
    if (n <= 512) {
        // small matrices (code capacity): one pass -- position of column i = number of columns with a smaller key, or the same key and a
        // smaller index.  n^2 / T uniform (broadcast) key reads per thread instead of 8 passes x 4 barriers: the OSD kernel's latency on a
        // handful of shots is what a Monte-Carlo step of the early-exit pipeline waits for.
        __syncthreads();
        for (int i = tid; i < n; i += T) {
            const unsigned long long ki = keys[i];
            int rank = 0;
            for (int j = 0; j < n; j++) { const unsigned long long kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
            ordw[rank] = (uint16_t)i;
        }
        __syncthreads();
        return;
    }
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 8 * pass;
        for (int e = tid; e < 256 * NW; e += T) cnt[e] = 0u;
        __syncthreads();
        constexpr int kSteps = 10;                                                          // steps of 64 positions kept in registers
        const bool cached = (span <= 64 * kSteps);
        int cj[kSteps];
        unsigned cd[kSteps];
        unsigned long long csame[kSteps];
        if (cached) {                                                                       // all index / key loads of the pass in flight at once
#pragma unroll
            for (int st = 0; st < kSteps; st++) { const int pos = wbeg + 64 * st + lane; cj[st] = (pos < wend) ? (int)pa[pos] : 0; }
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const int pos = wbeg + 64 * st + lane;
                cd[st] = (pos < wend) ? (unsigned)((keys[cj[st]] >> shift) & 255ull) : 0u;
            }
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const bool valid = (wbeg + 64 * st + lane) < wend;
                unsigned long long same = __ballot(valid);                                  // lanes of this step holding the same digit
#pragma unroll
                for (int b2 = 0; b2 < 8; b2++) {
                    const unsigned long long bal = __ballot((cd[st] >> b2) & 1u);
                    same &= ((cd[st] >> b2) & 1u) ? bal : ~bal;
                }
                csame[st] = valid ? same : 0ull;
            }
        }
        for (int round = 0; round < 2; round++) {                                           // 0: count, 1: scatter
            if (cached) {
#pragma unroll
                for (int st = 0; st < kSteps; st++) {
                    if (csame[st] != 0ull) {
                        const int rank = __builtin_popcountll(csame[st] & ((1ull << lane) - 1ull)), tot = __builtin_popcountll(csame[st]);
                        unsigned *slot = cnt + cd[st] * NW + wv;
                        if (round == 0) {
                            if (rank == 0) *slot += (unsigned)tot;                          // one leader per digit; the column [.][wv] is this wave's own
                        } else {
                            const unsigned base = *slot;                                    // read by the whole group before its leader advances it
                            pb[base + rank] = (uint16_t)cj[st];
                            if (rank == 0) *slot = base + (unsigned)tot;
                        }
                    }
                }
            } else
            for (int p0 = wbeg; p0 < wend; p0 += 64) {
                const int pos = p0 + lane;
                const bool valid = pos < wend;
                const int j = valid ? (int)pa[pos] : 0;
                const unsigned d = valid ? (unsigned)((keys[j] >> shift) & 255ull) : 0u;
                unsigned long long same = __ballot(valid);
#pragma unroll
                for (int b2 = 0; b2 < 8; b2++) {
                    const unsigned long long bal = __ballot((d >> b2) & 1u);
                    same &= ((d >> b2) & 1u) ? bal : ~bal;
                }
                if (valid) {
                    const int rank = __builtin_popcountll(same & ((1ull << lane) - 1ull)), tot = __builtin_popcountll(same);
                    unsigned *slot = cnt + d * NW + wv;
                    if (round == 0) {
                        if (rank == 0) *slot += (unsigned)tot;
                    } else {
                        const unsigned base = *slot;
                        pb[base + rank] = (uint16_t)j;
                        if (rank == 0) *slot = base + (unsigned)tot;
                    }
                }
            }
            __syncthreads();
            if (round == 0) {                                                               // exclusive scan over (digit, wave), 4 entries per thread
                unsigned v[4], sum = 0u;
#pragma unroll
                for (int e = 0; e < 4; e++) { v[e] = cnt[4 * tid + e]; sum += v[e]; }
                unsigned inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
                if (lane == 63) wsum[wv] = inc;
                __syncthreads();
                unsigned before = inc - sum;
                for (int w2 = 0; w2 < wv; w2++) before += wsum[w2];
#pragma unroll
                for (int e = 0; e < 4; e++) { cnt[4 * tid + e] = before; before += v[e]; }
                __syncthreads();
            }
        }
        uint16_t *tsw = pa; pa = pb; pb = tsw;
    }
    for (int j = tid; j < n; j += T) ordw[j] = pa[j];
    __syncthreads();
}

}  // namespace qldpc
