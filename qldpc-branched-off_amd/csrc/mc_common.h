// Philox4x32-10 counter-based generator and shared declarations of the Monte-Carlo pipeline.
// Stream definition (a pure function of (seed, shot, block, domain), reproducible on any device):
//   key = (seed_lo, seed_hi); counter = (shot_lo, shot_hi, block, domain).
//   domain 0: code-capacity error bits -- bit j of a shot uses word (j & 3) of block (j >> 2); error iff word < thr,
//   thr = floor(p * 2^32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

struct qldpc_graph;

namespace qldpc {

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                       uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// NB independent Philox4x32-10 blocks with the ROUND loop outermost: the ten rounds of one block are a dependent chain of 64-bit multiplies
// (ILP 2); written one block after the other the compiler leaves them that way and a wave that holds several blocks waits for every multiply
// (mc_first.hip: 2.6x off its issue time).  Same outputs as philox4x32_10 per block: block i uses counter (c0[i], c1[i], c2, c3).
template <int NB>
__device__ __forceinline__ void philox4x32_10_batch(const uint32_t (&c0)[NB], const uint32_t (&c1)[NB], uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                    uint32_t (&out)[NB][4]) {
    uint32_t a[NB], b[NB], c[NB], d[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { a[i] = c0[i]; b[i] = c1[i]; c[i] = c2; d[i] = c3; }
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * a[i], p1 = (uint64_t)0xCD9E8D57u * c[i];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ b[i] ^ k0, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ d[i] ^ k1, n3 = (uint32_t)p0;
            a[i] = n0; b[i] = n1; c[i] = n2; d[i] = n3;
        }
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) { out[i][0] = a[i]; out[i][1] = b[i]; out[i][2] = c[i]; out[i][3] = d[i]; }
}

inline uint32_t bernoulli_threshold(double p) {
    if (p <= 0.0) return 0u;
    if (p >= 1.0) return 0xFFFFFFFFu;
    return (uint32_t)std::floor(p * 4294967296.0);
}

int build_alpha_table(int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len, std::vector<double> &tab);
int gf2_spmv_launch(const qldpc_graph *g, int64_t B, const int8_t *d_vec, int8_t *d_out, hipStream_t stream);

// OSD-0 on the shots listed in d_list[0 .. *d_count) (device-resident count: no host sync).  d_ordering may be NULL
// (stable ascending |llr|); otherwise int32[B][n] indexed by shot.  solution may alias hard.  max_listed: an upper bound of *d_count.
// judge: a caller's per-record judge (logical failure / syndrome check of the solution against the true error, tallied) that the launch MAY take over --
// the one-wave kernels of small matrices do (fused = true on return); the caller launches its own judge kernel when fused stays false
struct OsdJudge { const int8_t *err; const uint64_t *Lmask; unsigned long long *tally; int32_t *count; int64_t max_listed; bool fused; };
int osd0_listed_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                       const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, OsdJudge *judge = nullptr);

}  // namespace qldpc
