// Bit-sliced first iteration of the code-capacity Monte-Carlo pipeline under reference semantics (per-shot early exit,
// src/decoding/kernels.py:361-364).
//
// A Monte-Carlo plan decodes against a UNIFORM prior p0 = log((1-p)/p) > 0 (alpha.py:119-120).  Iteration 0 of
// minsum_decoder_full is then a closed form of the syndrome alone:
//   * every variable-to-check message is p0 (kernels.py:263-265), so min1 = min2 = p0 and the message of check i to each of its
//     columns is  (-1)^{s_i} mag,  mag = alpha_0 * p0  (kernels.py:289-314; one rounding, as in the kernels);
//   * values_0[j] = p0 + sum of those messages over the column's checks in ascending row order (kernels.py:316-320).  The partial
//     sums are 0, +-mag, +-2 mag, +-RN(3 mag): whichever order the +1 / -1 terms come in, the same doubles appear, so for column
//     degree <= 3 the posterior is a function of (degree, number u_j of unsatisfied neighbour checks) only -- the host evaluates
//     it literally in every order and refuses the shortcut if they disagree (mc.hip);
//   * hard_0[j] = values_0[j] < 0 (kernels.py:349) = neg[deg_j][u_j], a table of <= 16 booleans;
//   * the decoder stops after this iteration iff H hard_0 == s (kernels.py:352-364): final_iter = 0.
// At BASELINE's error rates ~97 % of the shots end there.  This kernel does exactly that much for EVERY shot of a batch with shots
// in the BIT dimension: a lane owns BITS consecutive shots, word E[j] holds error bit j of those shots, s = H e (a6), the counts
// u_j, hard_0, the test H (e xor hard_0) == 0 and the logical comparison L (e xor hard_0) (engine.py:99-100) are word-wide
// XOR / AND / OR on LDS-resident bit planes.  Shots that do not stop here (and only those) are listed for the full decoder
// (minsum_regular.hip on a shot list), which replays them from their Philox stream: results are identical to running the full
// decoder on every shot -- tests compare the tallies of both pipelines with the CPU checker's.
// Sampling law and stream: mc_common.h (bit j of shot g = word j & 3 of Philox block (g, j >> 2) < thr): 36 Philox calls per shot
// are the floor of this kernel (~85 % of its instructions).
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

namespace qldpc {

struct FirstArgs {
    int m, n, k;
    const int32_t *indptr, *indices, *colptr, *rowidx, *lptr, *lidx;    // H (CSR, CSC), logical rows (CSR)
    int64_t B, shot_begin;
    uint32_t seed_lo, seed_hi, thr;
    unsigned negbits;                  // bit 4 * deg + u: values_0 < 0 for a column of that degree with u unsatisfied checks
    unsigned long long *tally;
    int32_t *cont_list, *cont_count;   // shots that go on to the full decoder
    unsigned long long *clk;
};

// CDEG / VDEG > 0: every row / column has exactly that degree (fixed-trip loops: the index loads of a row are one contiguous scalar load);
// 0: general CSR / CSC loops
template <int BITS, int CDEG, int VDEG>
__global__ __launch_bounds__(64, 2) void mc_first_kernel(FirstArgs A) {
    extern __shared__ unsigned char lds[];
    typedef typename std::conditional<BITS <= 8, uint8_t, typename std::conditional<BITS <= 16, uint16_t, uint32_t>::type>::type word_t;
    const int lane = threadIdx.x, m = A.m, n = A.n;
    // (restrict: the two plane arrays do not overlap, so the reads of one row's columns need not wait for the previous row's store)
    word_t *__restrict__ E = reinterpret_cast<word_t *>(lds);       // [n][64]: error plane j, later d = e xor hard_0
    word_t *__restrict__ S = reinterpret_cast<word_t *>(lds) + (size_t)n * 64;     // [m][64]: syndrome plane i
    const unsigned full = (BITS == 32) ? 0xFFFFFFFFu : ((1u << BITS) - 1u);
    const int64_t per_wave = (int64_t)64 * BITS;
    const ClkStamp clk0 = clk_begin(A.clk);
    unsigned t_trials = 0, t_conv = 0, t_zsyn = 0, t_zerr = 0;
    for (int64_t base = (int64_t)blockIdx.x * per_wave; base < A.B; base += (int64_t)gridDim.x * per_wave) {
        const int64_t first = base + (int64_t)lane * BITS;           // this lane's shots: first .. first + BITS - 1
        const int64_t left = A.B - first;
        const unsigned valid = left <= 0 ? 0u : (left >= BITS ? full : ((1u << (int)left) - 1u));
        // ---- e ~ Bernoulli(p)^n for BITS shots per lane: plane 4 q + w collects word w of block q of every shot ----
        const int nq = (n + 3) >> 2;
        for (int q = 0; q < nq; q++) {
            unsigned acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int sh0 = 0; sh0 < BITS; sh0 += 8) {                 // eight shots' blocks at a time, their rounds interleaved (mc_common.h)
                uint32_t glo[8], ghi[8], o[8][4];
#pragma unroll
                for (int i = 0; i < 8; i++) { const uint64_t g = (uint64_t)(A.shot_begin + first + sh0 + i); glo[i] = (uint32_t)g; ghi[i] = (uint32_t)(g >> 32); }
                philox4x32_10_batch<8>(glo, ghi, (uint32_t)q, 0u, A.seed_lo, A.seed_hi, o);
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int w = 0; w < 4; w++) acc[w] |= (o[i][w] < A.thr ? 1u : 0u) << (sh0 + i);
            }
#pragma unroll
            for (int w = 0; w < 4; w++)
                if (4 * q + w < n) E[(size_t)(4 * q + w) * 64 + lane] = (word_t)(acc[w] & valid);
        }
        // ---- s = H e (a6, kernels.py:222-231) ----
        unsigned anys = 0u;
#pragma unroll 4
        for (int i = 0; i < m; i++) {
            unsigned s = 0u;
            if (CDEG > 0) {
#pragma unroll
                for (int kk = 0; kk < CDEG; kk++) s ^= E[(size_t)A.indices[i * CDEG + kk] * 64 + lane];
            } else {
                for (int e = A.indptr[i]; e < A.indptr[i + 1]; e++) s ^= E[(size_t)A.indices[e] * 64 + lane];
            }
            S[(size_t)i * 64 + lane] = (word_t)s;
            anys |= s;
        }
        // ---- hard_0 from the number of unsatisfied neighbour checks (bit-sliced 2-bit counter), d = e xor hard_0 in place ----
#pragma unroll 4
        for (int j = 0; j < n; j++) {
            const int c0 = (VDEG > 0) ? j * VDEG : A.colptr[j], deg = (VDEG > 0) ? VDEG : A.colptr[j + 1] - c0;          // deg <= 3 (host-checked)
            unsigned u0 = 0u, u1 = 0u;
            if (VDEG > 0) {
#pragma unroll
                for (int d = 0; d < VDEG; d++) {
                    const unsigned s = S[(size_t)A.rowidx[c0 + d] * 64 + lane];
                    u1 |= u0 & s;                                            // (u1 u0) += s, never above 3
                    u0 ^= s;
                }
            } else {
                for (int d = 0; d < deg; d++) {
                    const unsigned s = S[(size_t)A.rowidx[c0 + d] * 64 + lane];
                    u1 |= u0 & s;
                    u0 ^= s;
                }
            }
            const unsigned nb = A.negbits >> (4 * deg);
            unsigned hard = 0u;
            if (nb & 1u) hard |= ~u1 & ~u0;
            if (nb & 2u) hard |= ~u1 & u0;
            if (nb & 4u) hard |= u1 & ~u0;
            if (nb & 8u) hard |= u1 & u0;
            E[(size_t)j * 64 + lane] = (word_t)((E[(size_t)j * 64 + lane] ^ hard) & valid);
        }
        // ---- stop test: H hard_0 == s  <=>  H d == 0 (kernels.py:352-364) ----
        unsigned unsat = 0u;
#pragma unroll 4
        for (int i = 0; i < m; i++) {
            unsigned s = 0u;
            if (CDEG > 0) {
#pragma unroll
                for (int kk = 0; kk < CDEG; kk++) s ^= E[(size_t)A.indices[i * CDEG + kk] * 64 + lane];
            } else {
                for (int e = A.indptr[i]; e < A.indptr[i + 1]; e++) s ^= E[(size_t)A.indices[e] * 64 + lane];
            }
            unsat |= s;
        }
        const unsigned conv = valid & ~unsat, cont = valid & unsat;
        // ---- logical comparison of the shots that stop: L (e xor e_hat) != 0 (engine.py:99-100) ----
        unsigned lerr = 0u;
        for (int r = 0; r < A.k; r++) {
            unsigned s = 0u;
            for (int e = A.lptr[r]; e < A.lptr[r + 1]; e++) s ^= E[(size_t)A.lidx[e] * 64 + lane];
            lerr |= s;
        }
        t_trials += __builtin_popcount(conv);
        t_conv += __builtin_popcount(conv);
        t_zsyn += __builtin_popcount(conv & ~anys);
        t_zerr += __builtin_popcount(conv & lerr);
        // ---- list the shots that go on: one atomic per wave, lane offsets from a wave prefix sum ----
        const int mine = __builtin_popcount(cont);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63, 64);
        if (total > 0) {
            int start = 0;
            if (lane == 63) start = atomicAdd(A.cont_count, total);
            start = __shfl(start, 63, 64) + incl - mine;
            unsigned c = cont;
            while (c) {
                const int sh = __builtin_ctz(c);
                c &= c - 1u;
                A.cont_list[start++] = (int32_t)(first + sh);
            }
        }
    }
    clk_end(A.clk, clk0);
    unsigned v4[4] = {t_trials, t_conv, t_zsyn, t_zerr};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned x = v4[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        v4[i] = x;
    }
    if (lane == 0) {
        if (v4[0]) { atomicAdd(&A.tally[QLDPC_TALLY_TRIALS], (unsigned long long)v4[0]); atomicAdd(&A.tally[QLDPC_TALLY_ITERS_Z], (unsigned long long)v4[0]); }   // final_iter + 1 = 1
        if (v4[1]) atomicAdd(&A.tally[QLDPC_TALLY_BP_CONV_Z], (unsigned long long)v4[1]);
        if (v4[2]) atomicAdd(&A.tally[QLDPC_TALLY_ZERO_SYND_Z], (unsigned long long)v4[2]);
        if (v4[3]) { atomicAdd(&A.tally[QLDPC_TALLY_Z_ERR], (unsigned long long)v4[3]); atomicAdd(&A.tally[QLDPC_TALLY_TOTAL_ERR], (unsigned long long)v4[3]); }
    }
}

// values_0 of a column of degree `deg` with `u` unsatisfied checks, summed literally (kernels.py:279,316,320) in EVERY order of the
// +mag / -mag terms; false when two orders give different doubles (then the shortcut is not exact and must not be used)
static bool first_iteration_value(double p0, double mag, int deg, int u, double &out) {
    std::vector<int> sign(deg, 0);
    for (int i = 0; i < u; i++) sign[i] = 1;
    std::sort(sign.begin(), sign.end());
    bool have = false;
    do {
        double s = 0.0;
        for (int i = 0; i < deg; i++) s += sign[i] ? -mag : mag;
        const double v = s + p0;
        if (have && !(v == out)) return false;
        out = v; have = true;
    } while (std::next_permutation(sign.begin(), sign.end()));
    return true;
}

// negbits for FirstArgs, or false when the closed form does not apply (p0 <= 0, a column heavier than 3, max_iter < 1, order-dependent sums)
bool mc_first_table(const qldpc_graph *g, double p0, double alpha0, double clip, int max_iter, unsigned &negbits) {
    if (!(p0 > 0.0) || !std::isfinite(p0) || !(alpha0 > 0.0) || !std::isfinite(alpha0) || max_iter < 1 || g->max_col_deg > 3 || g->m < 1 || g->n < 1) return false;
    if (!(std::fabs(p0) <= clip)) return false;               // iteration 1 of the full decoder would clip Q; keep the two pipelines on identical ground
    for (int i = 0; i < g->m; i++) if (g->indptr[i + 1] - g->indptr[i] < 2) return false;       // degree-1 checks: min2 = inf
    const double mag = alpha0 * std::fabs(p0);                // the kernels' rounding: (+-alpha) * |p0|
    negbits = 0u;
    for (int deg = 0; deg <= 3; deg++)
        for (int u = 0; u <= deg; u++) {
            double v = 0.0;
            if (!first_iteration_value(p0, mag, deg, u, v)) return false;
            if (v < 0.0) negbits |= 1u << (4 * deg + u);
        }
    return true;
}

static int first_bits_option = 8;
void mc_first_set_bits(int bits) { first_bits_option = bits; }

int mc_first_launch(const qldpc_graph *g, int k, const int32_t *d_lptr, const int32_t *d_lidx, int64_t B, uint64_t seed, int64_t shot_begin, uint32_t thr,
                    unsigned negbits, unsigned long long *d_tally, int32_t *d_cont_list, int32_t *d_cont_count, unsigned long long *d_clk, hipStream_t stream) {
    FirstArgs A{};
    A.m = g->m; A.n = g->n; A.k = k;
    A.indptr = g->d_indptr; A.indices = g->d_indices; A.colptr = g->d_colptr; A.rowidx = g->d_rowidx; A.lptr = d_lptr; A.lidx = d_lidx;
    A.B = B; A.shot_begin = shot_begin; A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.thr = thr; A.negbits = negbits;
    A.tally = d_tally; A.cont_list = d_cont_list; A.cont_count = d_cont_count; A.clk = d_clk;
    int bits = first_bits_option;
    while (bits > 8 && (size_t)(g->m + g->n) * 64 * (bits <= 16 ? 2 : 4) > 64 * 1024) bits /= 2;      // the bit planes of one wave must fit 64 KB
    const size_t lds = (size_t)(g->m + g->n) * 64 * (bits <= 8 ? 1 : (bits <= 16 ? 2 : 4));
    if (lds > 64 * 1024) { set_error("first-iteration kernel: graph too large for the LDS bit planes"); return QLDPC_ERR_UNSUPPORTED; }
    const int64_t per_wave = (int64_t)64 * bits;
    const int64_t tasks = (B + per_wave - 1) / per_wave;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 256)));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tasks, (int64_t)256 * per_cu));
    int rc;
    bool reg63 = (g->max_row_deg == 6 && g->max_col_deg == 3);
    for (int i = 0; i < g->m && reg63; i++) reg63 = (g->indptr[i + 1] - g->indptr[i] == 6);
    for (int j = 0; j < g->n && reg63; j++) reg63 = (g->colptr[j + 1] - g->colptr[j] == 3);
    bool launched = false;
#define QLDPC_FIRST_CASE(BITS, CD, VD)                                                                                               \
    if (!launched && bits == BITS && reg63 == (CD > 0)) {                                                                            \
        if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(mc_first_kernel<BITS, CD, VD>), 64 * 1024)) != QLDPC_OK) return rc; \
        hipLaunchKernelGGL((mc_first_kernel<BITS, CD, VD>), dim3(grid), dim3(64), lds, stream, A);                                  \
        launched = true;                                                                                                             \
    }
    QLDPC_FIRST_CASE(8, 6, 3) QLDPC_FIRST_CASE(16, 6, 3) QLDPC_FIRST_CASE(32, 6, 3) QLDPC_FIRST_CASE(8, 0, 0) QLDPC_FIRST_CASE(16, 0, 0) QLDPC_FIRST_CASE(32, 0, 0)
#undef QLDPC_FIRST_CASE
    if (!launched) { set_error("first-iteration kernel: bits per lane must be 8, 16 or 32"); return QLDPC_ERR_INVALID; }
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

}  // namespace qldpc
