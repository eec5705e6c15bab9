// Shared pieces of the OSD-0 kernels (gf2.hip, osd_gj.hip, osd_gjg.hip, osd_small.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

// Phase timers of the OSD kernels exist only in the diagnostic build (`make timers` -> libqldpc_hip_timers.so, -DQLDPC_OSD_TIMERS);
// the default build carries no clock reads.  Counters (uint64[32]; [16..22] belong to the workgroup BP kernel: wave-iterations, check pass, its barrier, freeze, variable pass, its barrier): [0] shots, [1] chunks, [2] columns taken into blocks, [3] pivots,
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
// The passes: orders the index list pa [0 .. len) (initially in the order ties are to come out in) by keys [pa [i]] and leaves it in out [0 .. len).
// KS = steps of 64 positions a wave keeps in registers per pass (lists of up to 1024 * KS entries; longer ones walk their range in a loop; KS = 0: loop only).
template <int KS>
__device__ __forceinline__ void osd_radix_passes(const unsigned long long *keys, int len, uint16_t *pa, uint16_t *pb, unsigned *cnt, uint16_t *out) {
    const int n = len;
    const int tid = threadIdx.x, T = blockDim.x;
    const int NW = T >> 6, wv = tid >> 6, lane = tid & 63;
    unsigned *wsum = cnt + 256 * NW;
    const int span = (((n + NW - 1) / NW) + 63) & ~63;                                      // positions per wave, a multiple of 64
    const int wbeg = wv * span, wend = min(n, wbeg + span);
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 8 * pass;
        for (int e = tid; e < 256 * NW; e += T) cnt[e] = 0u;
        __syncthreads();
        constexpr int kSteps = KS > 0 ? KS : 1;                                             // steps of 64 positions kept in registers
        const bool cached = KS > 0 && (span <= 64 * kSteps);
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
    for (int j = tid; j < n; j += T) out[j] = pa[j];
    __syncthreads();
}



__device__ __forceinline__ void osd_radix_sort(const double *__restrict__ llr, int n, unsigned long long *keys, uint16_t *pa, uint16_t *pb,
                                               unsigned *cnt, uint16_t *ordw) {
    const int tid = threadIdx.x, T = blockDim.x;
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
    osd_radix_passes<10>(keys, n, pa, pb, cnt, ordw);
}

// Order of a short list (len <= 2048 columns in pa, ascending column index) by value buckets: keys are non-negative doubles <= bound (finite).
// Scratch: pb [len] bucket of an entry, pa [2048 .. 2048 + len) the columns in bucket order, cnt [0 .. 2048) cursors, cnt [2048 .. 4096) bucket starts, cnt [4096 ..) wave sums.
// Returns false, with pa untouched, when a bucket holds more than 48 entries (a long run of equal keys, a lopsided distribution): the rank among bucket mates is
// quadratic in the bucket, the caller's radix passes are not.
__device__ __forceinline__ bool osd_bucket_order(const unsigned long long *keys, int len, double bound, uint16_t *pa, uint16_t *pb, unsigned *cnt, uint16_t *out) {
    constexpr int NB = 2048;
    const int tid = threadIdx.x, T = blockDim.x, wv = tid >> 6, lane = tid & 63;
    unsigned *cursor = cnt, *start = cnt + NB, *wsum = cnt + 2 * NB, *heavy = cnt + 2 * NB + 24;
    uint16_t *byb = pa + 2048;
    const double c = (bound > 0.0) ? (double)NB / bound : 0.0;
    for (int e = tid; e < NB; e += T) cursor[e] = 0u;
    if (tid == 0) *heavy = 0u;
    __syncthreads();
    for (int i = tid; i < len; i += T) {
        const double x = __longlong_as_double((long long)keys[pa[i]]);
        const int bk = min(NB - 1, (int)(x * c));
        pb[i] = (uint16_t)bk;
        atomicAdd(&cursor[bk], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the bucket counts: a contiguous share per thread, a wave scan of the shares, the waves' sums
        const int per = (NB + T - 1) / T, b0 = min(NB, tid * per), b1 = min(NB, b0 + per);
        unsigned sum = 0u, most = 0u;
        for (int e = b0; e < b1; e++) { const unsigned v = cursor[e]; sum += v; most = max(most, v); }
        if (most > 48u) *heavy = 1u;
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        if (*heavy) { __syncthreads(); return false; }                                        // (uniform; the barrier keeps the flag from the next caller's reset)
        unsigned run = inc - sum;
        for (int w2 = 0; w2 < wv; w2++) run += wsum[w2];
        for (int e = b0; e < b1; e++) { const unsigned v = cursor[e]; start[e] = run; cursor[e] = run; run += v; }
    }
    __syncthreads();
    for (int i = tid; i < len; i += T) byb[atomicAdd(&cursor[pb[i]], 1u)] = pa[i];            // (columns, in any order inside a bucket)
    __syncthreads();
    for (int i = tid; i < len; i += T) {
        const int bk = pb[i], j = pa[i];
        const unsigned long long k = keys[j];
        const int s0 = (int)start[bk], s1 = (bk + 1 < NB) ? (int)start[bk + 1] : len;
        int rank = 0;
        for (int q = s0; q < s1; q++) {
            const int j2 = byb[q];
            const unsigned long long k2 = keys[j2];
            rank += (k2 < k || (k2 == k && j2 < j)) ? 1 : 0;
        }
        out[s0 + rank] = (uint16_t)j;
    }
    __syncthreads();
    return true;
}

// The first columns of that order only.  The free-pivot kernels stop their sweep when the residual syndrome is gone (osd_gj.hip) -- on the circuit-level
// matrices after ~170 of 8 800 columns -- so a full sort is mostly wasted: this one finds a key bound tau with `want` <= #{key <= tau} (from a sample of the keys,
// checked by the count; a radix select when the sample falls short), splits the columns into S = {key <= tau} and the rest, both
// in index order, and sorts S alone.  Returns Kt = |S|: ordw [0 .. Kt) is the head of the full order, ordw [Kt .. n) the other columns in INDEX order --
// a caller that gets that far sorts them then (osd_sort_rest).  Scratch as for osd_radix_sort; cnt needs 256 * waves + waves + 8 words.
__device__ __forceinline__ int osd_radix_sort_head(const double *__restrict__ llr, int n, int want, unsigned long long *keys, uint16_t *pa, uint16_t *pb,
                                                   unsigned *cnt, uint16_t *ordw) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int NW = T >> 6, wv = tid >> 6, lane = tid & 63;
    if (want <= 0 || n <= 512 || n < 2 * want) { osd_radix_sort(llr, n, keys, pa, pb, cnt, ordw); return n; }
    for (int j = tid; j < n; j += T) keys[j] = osd_key(llr[j]);
    // membership of the head: (key >> shift) <= prefix.  Every wave owns a contiguous range of column indices (the split below is stable).
    unsigned *wsum = cnt + 264;                                                             // [waves] head members per wave
    const int span = (((n + NW - 1) / NW) + 63) & ~63;
    const int wbeg = min(n, wv * span), wend = min(n, wbeg + span);
    auto count_head = [&](int sh, unsigned long long pf) -> int {
        int mine = 0;
        for (int p0 = wbeg; p0 < wend; p0 += 64) {
            const int j = p0 + lane;
            mine += __builtin_popcountll(__ballot((j < wend) && (keys[j] >> sh) <= pf));
        }
        __syncthreads();                                                                    // (wsum may still be read from the previous count)
        if (lane == 0) wsum[wv] = (unsigned)mine;
        __syncthreads();
        int tot = 0;
        for (int w2 = 0; w2 < NW; w2++) tot += (int)wsum[w2];
        return tot;
    };
    // (1) a key bound from a stratified sample of 256 keys: the sample key of rank ~1.15 * 256 * want / n (+ 4).  A per-key radix select costs 20 k cycles per
    //     level of 8 bits -- more than sorting the head; the sample costs ~3 k, and the count that the split needs anyway tells whether it was large enough.
    int shift = 0, Kt;
    unsigned long long prefix;
    {
        unsigned long long *smp = reinterpret_cast<unsigned long long *>(cnt);             // [256] sample keys
        unsigned *srank = cnt + 512;                                                        // [256]
        unsigned long long *tau = reinterpret_cast<unsigned long long *>(cnt + 768);
        __syncthreads();
        if (tid < 256) {
            const int stratum = n / 256;
            smp[tid] = keys[(int)(((long long)tid * n) >> 8) + (int)(((unsigned)tid * 2654435761u >> 16) % (unsigned)stratum)];
            srank[tid] = 0u;
        }
        __syncthreads();
        const int parts = T >> 8, i = tid & 255, q = tid >> 8;                              // sample i against a share of the others (every lane another sample:
        if (q < parts) {                                                                    //  the compared key is a broadcast read)
            const unsigned long long ki = smp[i];
            unsigned less = 0u;
            for (int c = q * 256 / parts; c < (q + 1) * 256 / parts; c++) {
                const unsigned long long kc = smp[c];
                less += (kc < ki || (kc == ki && c < i)) ? 1u : 0u;
            }
            atomicAdd(&srank[i], less);
        }
        __syncthreads();
        const unsigned rstar = min(254u, (unsigned)((256ll * want * 23 / 20 + n - 1) / n) + 4u);
        if (tid < 256 && srank[tid] == rstar) *tau = smp[tid];
        __syncthreads();
        prefix = *tau;
        Kt = count_head(0, prefix);
    }
    // (2) the sample fell short (a few percent of the shots): the exact bound by a radix select -- 8-bit histograms from the top byte down, until the bin that
    //     holds the want-th key is small
    if (Kt < want) {
        unsigned *hist = cnt;                                                               // [256]
        unsigned *sel = cnt + 256;                                                          // [0] bin, [1] keys below the bin, [2] keys in it
        int below = 0;
        prefix = 0ull;
        for (shift = 56;; shift -= 8) {
            __syncthreads();
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            for (int j0 = 0; j0 < n; j0 += T) {
                // keys crowd into a few bins at the top levels (the exponent bytes): the lanes of a wave that share a digit add their count with ONE atomic
                // (64 atomics on one LDS address take 64 passes of the pipe); what is left after four such groups goes lane by lane
                const int j = j0 + tid;
                const unsigned long long k = (j < n) ? keys[j] : 0ull;
                const unsigned d = (unsigned)(k >> shift) & 255u;
                bool take = (j < n) && (shift == 56 || (k >> (shift + 8)) == prefix);
                unsigned long long active = __ballot(take);
                for (int grp = 0; grp < 4 && active != 0ull; grp++) {
                    const int leader = __builtin_ctzll(active);
                    const unsigned dl = (unsigned)__builtin_amdgcn_readlane((int)d, leader);
                    const unsigned long long same = __ballot(take && d == dl);
                    if (lane == leader) atomicAdd(&hist[dl], (unsigned)__builtin_popcountll(same));
                    if (d == dl) take = false;
                    active &= ~same;
                }
                if (take) atomicAdd(&hist[d], 1u);
            }
            __syncthreads();
            if (wv == 0) {                                                                  // lane: bins 4 lane .. 4 lane + 3
                unsigned v[4], sum = 0u;
#pragma unroll
                for (int e = 0; e < 4; e++) { v[e] = hist[4 * lane + e]; sum += v[e]; }
                unsigned inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
                unsigned run = (unsigned)below + inc - sum;
                int found = -1;
                unsigned fbelow = 0u, fin = 0u;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (found < 0 && run + v[e] >= (unsigned)want) { found = 4 * lane + e; fbelow = run; fin = v[e]; }
                    run += v[e];
                }
                const unsigned long long bal = __ballot(found >= 0);                        // (n >= want: some bin reaches it)
                if (bal != 0ull && lane == __builtin_ctzll(bal)) { sel[0] = (unsigned)found; sel[1] = fbelow; sel[2] = fin; }
            }
            __syncthreads();
            prefix = (prefix << 8) | (unsigned long long)sel[0];
            below = (int)sel[1];
            if (below + (int)sel[2] <= want + want / 2 || shift == 0) break;
        }
        Kt = count_head(shift, prefix);
    }
    // (3) the stable split: head members to pa, the others to ordw [Kt ..), both in index order
    int baseS = 0;
    for (int w2 = 0; w2 < wv; w2++) baseS += (int)wsum[w2];
    int baseR = wbeg - baseS;
    for (int p0 = wbeg; p0 < wend; p0 += 64) {
        const int j = p0 + lane;
        const bool valid = j < wend;
        const bool in = valid && (keys[j] >> shift) <= prefix;
        const unsigned long long bal = __ballot(in), rest = __ballot(valid) & ~bal, lt = (1ull << lane) - 1ull;
        if (in) pa[baseS + __builtin_popcountll(bal & lt)] = (uint16_t)j;
        else if (valid) ordw[Kt + baseR + __builtin_popcountll(rest & lt)] = (uint16_t)j;
        baseS += __builtin_popcountll(bal);
        baseR += __builtin_popcountll(rest);
    }
    __syncthreads();
    // (4) the head in order.  Eight radix passes over ~1 300 keys cost 70 k cycles, nearly all of it fixed (a scan of 4 096 counters and five barriers per pass).
    //     The head's keys are |llr| values between 0 and the bound: 2 048 buckets by VALUE (floor (key * 2048 / bound): monotone in the key, equal keys share a
    //     bucket) hold 0.6 keys each, so bucket offsets + a rank among the bucket mates by (key, column) give the exact stable order in five barriers.
    if (shift == 0 && Kt <= 2048 && n >= 2 * 2048 && prefix < 0x7FF0000000000000ull) {
        if (osd_bucket_order(keys, Kt, __longlong_as_double((long long)prefix), pa, pb, cnt, ordw)) return Kt;
    }
    if (Kt <= 2048) osd_radix_passes<2>(keys, Kt, pa, pb, cnt, ordw);
    else osd_radix_passes<10>(keys, Kt, pa, pb, cnt, ordw);
    return Kt;
}

// ... and the other columns, when a sweep does get past the head: ordw [from .. n) holds them in index order; scratch in GLOBAL memory (the caller's LDS is
// in use by then): keys [n], pa / pb [n - from] each, cnt as above.
__device__ __forceinline__ void osd_sort_rest(const double *__restrict__ llr, int n, int from, unsigned long long *keys, uint16_t *pa, uint16_t *pb,
                                              unsigned *cnt, uint16_t *ordw) {
    const int tid = threadIdx.x, T = blockDim.x, rest = n - from;
    for (int j = tid; j < n; j += T) keys[j] = osd_key(llr[j]);
    for (int i = tid; i < rest; i += T) pa[i] = ordw[from + i];
    __threadfence_block();
    __syncthreads();
    osd_radix_passes<0>(keys, rest, pa, pb, cnt, ordw + from);
    __threadfence_block();
    __syncthreads();
}

// U is stored row-major with an XOR swizzle of the word index when rows are 16 words (128 B): conflict-free row-parallel updates
// (rows of 16 words belong to threads as q = 16 * lane + (wave + lane) % 16, see phase 3: the swizzle follows the lane and moves PAIRS of
// words, so that a row can also be read and written 16 bytes at a time)
__device__ __forceinline__ int uswz(int q, int w, int mw) { return q * mw + ((mw == 16) ? (w ^ ((q >> 3) & 14)) : w); }

// ---- phase 2 of the LDS kernel in ONE wave, registers only (rows of <= 16 words) ----
// lane = 4 * w + g holds word w of the four columns t = 4 i + g (i = 0..3) of the block.  A pivot step is: one ballot over all 16 words
// of column t (first live set bit, kernels.py:71-75), two lane reads, the column turned into the elimination mask, the mask handed to
// the other three lanes of every quad by a DPP quad broadcast, and for each register that still holds later columns two ballots (bits
// a and pp of those columns), the swap (kernels.py:79-82) and the XOR (kernels.py:88-92) -- no LDS access and no barrier inside the
// chain; the step index is a template parameter so every register index and DPP pattern is static.
struct QuadPivot {
    unsigned long long X[4];       // the lane's word of columns g, 4 + g, 8 + g, 12 + g
    unsigned long long live;       // positions >= lrow within the lane's word
    int lrow, nops;
    uint32_t depmask;              // columns found dependent
    int oppv, optv;                // lane k: pivot position / column index of operation k
    bool stop;
    unsigned nzw;                  // diagnostic build: non-zero words over the masks
};

template <int G>
__device__ __forceinline__ unsigned long long quad_bcast(unsigned long long x) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, G * 0x55, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), G * 0x55, 0xF, 0xF, false);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ unsigned long long sext64(int f) { return ((unsigned long long)(uint32_t)f << 32) | (uint32_t)f; }

template <int T>
__device__ __forceinline__ void quad_pivot_step(QuadPivot &S, unsigned long long *R, int mw, int lane, int rankH, int m) {
    constexpr int IT = T >> 2, GT = T & 3;
    const int g = lane & 3, w = lane >> 2;
    const unsigned long long owners = 0x1111111111111111ull << GT;
    const unsigned long long mword = S.X[IT] & S.live;
    const unsigned long long bal = __ballot(mword != 0ull) & owners;
    if (bal == 0ull) { S.depmask |= 1u << T; return; }                                      // dependent on the pivots so far
    const int src = __builtin_ctzll(bal);
    const unsigned long long pword = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), src) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, src);
    const int wp = src >> 2, pb = __builtin_ctzll(pword), pp = wp * 64 + pb, a = S.lrow, wa = a >> 6;
    const unsigned long long al = (w == wa) ? (1ull << (a & 63)) : 0ull, pl = (w == wp) ? (1ull << pb) : 0ull;
    // the pivot column becomes the elimination mask: bits a <-> pp swapped (bit pp is 1), then bit a cleared
    const bool olda = (__ballot((S.X[IT] & al) != 0ull) & owners) != 0ull;
    unsigned long long rm = olda ? (S.X[IT] | pl) : (S.X[IT] & ~pl);
    rm &= ~al;                                                                               // (in this order: a == pp must end with bit a clear)
    if (g == GT && w < mw) R[T * mw + w] = rm;
#ifdef QLDPC_OSD_TIMERS
    S.nzw += (unsigned)__builtin_popcountll(__ballot(rm != 0ull) & owners);
#endif
    const unsigned long long rmq = quad_bcast<GT>(rm), sw = al | pl;
#pragma unroll
    for (int i = IT; i < 4; i++) {
        if (4 * i + 3 <= T) continue;                                                        // no later column in this register
        const unsigned long long x = S.X[i];
        const uint32_t na = (uint32_t)(__ballot((x & al) != 0ull) >> (4 * wa)), np = (uint32_t)(__ballot((x & pl) != 0ull) >> (4 * wp));
        int fa = __builtin_amdgcn_sbfe((int)na, g, 1), fp = __builtin_amdgcn_sbfe((int)np, g, 1);          // 0 / -1: bit a, bit pp of the lane's column
        if (i == IT) { const int later = (g > GT) ? -1 : 0; fa &= later; fp &= later; }      // columns <= t of this register are finished
        S.X[i] = x ^ (sext64(fa ^ fp) & sw) ^ (sext64(fp) & rmq);                            // swap, then add the pivot row where bit a is set
    }
    S.live &= ~al;
    S.oppv = (lane == S.nops) ? pp : S.oppv;
    S.optv = (lane == S.nops) ? T : S.optv;
    S.nops++; S.lrow++;
    if (S.lrow >= rankH || S.lrow >= m) S.stop = true;                                       // full rank: the remaining columns cannot pivot
}


// ---- phase 3 for rows of <= 16 words in LDS: the operations of one block applied to the lane's row (gf2.hip explains why it looks like this)
// rowbase[w ^ swz] is word w of the row (swz: the row's pair swizzle, 0 for an unswizzled vector); `act` = the lane has a row; all 64 lanes
// of a wave call this together.  Operation k: positions a_k = row0 + k and pp_k (lane k of every 16 of ppv), elimination mask
// R + ptv_k * mw.  Only 3.5 % of the (row, operation) pairs change a row of the circuit-level transforms: the row reads the 2 + 16 words
// holding the tested positions of the WHOLE block back to back, keeps one bit per operation and kind (ab: bit a_k, pb: bit pp_k), the wave
// visits only operations some lane has a bit for, and a changed row updates its later bits from two ballots instead of reading again.
__device__ __forceinline__ void osd_rows_apply(unsigned long long *rowbase, int swz, bool act, int row0, int nops, int mw, int ppv, int ptv,
                                               const unsigned long long *R, int tid, unsigned long long &d_wops, unsigned long long &d_lops) {
    const int ws = row0 >> 6, sh = row0 & 63;
    const uint32_t valid = (1u << nops) - 1u;
    const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rowbase);
    // lanes 0-15 of every 32 stand for position a_j = row0 + j, lanes 16-31 for pp_j (j = lane % 16): see the bit updates below
    const int j16 = tid & 15, mypos = (j16 < nops) ? ((tid & 16) ? ppv : row0 + j16) : 0;
    uint32_t ab, pb = 0u;
    {
        const unsigned long long A0 = rowbase[ws ^ swz], A1 = (ws + 1 < mw) ? rowbase[(ws + 1) ^ swz] : 0ull;
        uint32_t Pw[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int pk = (k < nops) ? __builtin_amdgcn_readlane(ppv, k) : 0;
            Pw[k] = row32[2 * ((pk >> 6) ^ swz) + ((pk >> 5) & 1)];
        }
        ab = (uint32_t)((A0 >> sh) | (sh ? (A1 << (64 - sh)) : 0ull)) & valid;             // positions row0 .. row0 + nops - 1
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int pk = __builtin_amdgcn_readlane(ppv, k);
            pb |= ((Pw[k] >> (pk & 31)) & 1u) << k;
        }
        pb &= valid;
        if (!act) { ab = 0u; pb = 0u; }
    }
    int kdone = -1;
    for (;;) {
        uint32_t x = ab | pb;                                                               // OR over the wave
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);        // quad_perm [2,3,0,1]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);       // row_half_mirror
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);       // row_mirror
        uint32_t wtb = (uint32_t)__builtin_amdgcn_readlane((int)x, 0) | (uint32_t)__builtin_amdgcn_readlane((int)x, 16) |
                       (uint32_t)__builtin_amdgcn_readlane((int)x, 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
        if (kdone >= 0) wtb &= ~1u << kdone;
        if (wtb == 0u) break;
        const int k = __builtin_ctz(wtb);
        kdone = k;
        const int ppk = __builtin_amdgcn_readlane(ppv, k), a = row0 + k;
        const unsigned long long *mk = R + __builtin_amdgcn_readlane(ptv, k) * mw;
        // what operation k does to the bits the later operations j > k test in a row it changes: the swap puts the row's old bit a_k at
        // position pp_k (xx: bit j = (a_j == pp_k), bit 16 + j = (pp_j == pp_k)); the XOR flips them by the mask's bits at those
        // positions (mm: bit j = mask_k[a_j], bit 16 + j = mask_k[pp_j])
        const bool later = (j16 > k) && (j16 < nops);
        const uint32_t mword = reinterpret_cast<const uint32_t *>(mk)[2 * (mypos >> 6) + ((mypos >> 5) & 1)];
        const uint32_t xx = (uint32_t)__ballot(later && mypos == ppk);
        const uint32_t mm = (uint32_t)__ballot(later && ((mword >> (mypos & 31)) & 1u));
        const bool ba = (ab >> k) & 1u, bp = (pb >> k) & 1u;
#ifdef QLDPC_OSD_TIMERS
        { d_wops++; d_lops += __builtin_popcountll(__ballot(ba || bp)); }
#endif
        if (ba != bp) {                                                                     // kernels.py:79-82: swap bits a <-> pp
            const int wa = a >> 6, wp = ppk >> 6;
            const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (ppk & 63);
            if (wp == wa) { rowbase[wa ^ swz] ^= abit ^ pbit; }
            else { const unsigned long long xa = rowbase[wa ^ swz], xp = rowbase[wp ^ swz]; rowbase[wa ^ swz] = xa ^ abit; rowbase[wp ^ swz] = xp ^ pbit; }
            const uint32_t sa = xx & 0xFFFFu, sp = xx >> 16;
            ab = ba ? (ab | sa) : (ab & ~sa);
            pb = ba ? (pb | sp) : (pb & ~sp);
        }
        if (bp) {                                                                           // bit a after the swap: add the pivot row (kernels.py:88-92)
            if (mw == 16) {                                                                 // all reads in flight before the first XOR
                ulonglong2 u[8], k2[8];
                ulonglong2 *Uq = reinterpret_cast<ulonglong2 *>(rowbase);
                const ulonglong2 *mk2 = reinterpret_cast<const ulonglong2 *>(mk);
                const int sz = swz >> 1;
#pragma unroll
                for (int w = 0; w < 8; w++) u[w] = Uq[w ^ sz];
#pragma unroll
                for (int w = 0; w < 8; w++) k2[w] = mk2[w];
#pragma unroll
                for (int w = 0; w < 8; w++) { u[w].x ^= k2[w].x; u[w].y ^= k2[w].y; }
#pragma unroll
                for (int w = 0; w < 8; w++) Uq[w ^ sz] = u[w];
            } else {
                for (int w = 0; w < mw; w++) rowbase[w ^ swz] ^= mk[w];
            }
            ab ^= mm & 0xFFFFu;
            pb ^= mm >> 16;
        }
    }
}

}  // namespace qldpc
