// OSD-0 (a9, reference src/decoding/osd.py:5-29 + gf2_elimination_packed_core, src/decoding/kernels.py:48-96) for matrices with
// m <= 1024 rows: one workgroup per shot, the row transform in LDS, FREE pivot rows -- the round-3 kernel.
//
// What the reference computes: the columns of H in ascending-|llr| order, a Gauss-Jordan elimination that takes the first non-zero row at or
// below the diagonal as the pivot (kernels.py:71-82), and e[pivot column] = reduced rhs at the pivot row (osd.py:19-25).  What that IS:
//   S = the columns that are independent of the columns before them in that order (a property of the column spaces only), and
//   e_S = the solution of H_S e_S = b, unique because H_S has full column rank.
// Neither depends on WHICH row of a column is taken as its pivot as long as b lies in the column space of H -- always the case for a
// syndrome that an error produced.  This kernel therefore takes any unused row (the first set bit of the reduced column among the rows that
// have not pivoted yet) and drops everything the reference's row choice costs a formulation that tracks it: no row swaps (no position
// table, no second tested bit per operation), and -- the larger gain -- the operations of a block of 16 columns become ORDER-FREE:
//   * state: U = T^T for the accumulated row transform T (current rows = T * original rows), m x m bits in LDS; row m is all zero (padding
//     target of short columns), row m + 1 carries the right-hand side b = s + H hard (it transforms like a column);
//   * the reduced form of a sparse column h (<= 6 ones) is XOR_{i in supp h} U[i]; it pivots iff it has a one in an unused row;
//   * a block of 16 columns is resolved in ONE wave on registers (lane = 4 w + g holds word w of columns g, 4 + g, 8 + g, 12 + g) by a full
//     Gauss-Jordan among the 16: a pivot step clears the pivot row pp_k from EVERY other column of the block, the finished ones included.
//     The finished columns then are the columns C_k of the block's composite transform  E = I + sum_k C_k e_{pp_k}^T,  i.e.
//         new row p' of T = old row p' + sum_{k : C_k[p']} old row pp_k      <=>      U[q] ^= XOR_{k : bit pp_k of U[q]} C_k
//     with the bits tested on the OLD U[q]: a thread owning row q reads its 16 bits, and whatever is set selects masks to add, in any order
//     (the reference-order kernel in gf2.hip replays the 16 operations one after the other per row, tracking two bits per operation);
//   * dependent columns are dropped in parallel batches and the sweep stops at rank(H), as in the reference-order kernel.
// A right-hand side outside the column space (only a caller's own syndromes can be) shows as a one of the reduced b in an unused row; such a
// shot is put on a list and solved by the reference-order kernel (gf2.hip) afterwards, so every input still gets the reference's answer.
#include "osd_gj.h"

#include <algorithm>

namespace qldpc {

// b (row `brow` of U) has no one in a row outside `used`: wave-uniform, every thread reads the same LDS words
template <bool W16>
__device__ __forceinline__ bool gj_residual_gone(const unsigned long long *U, const unsigned long long *used, int brow, int mw) {
    unsigned long long z = 0ull;
    for (int w = 0; w < mw; w++) z |= U[W16 ? brow * 16 + (w ^ ((brow >> 3) & 14)) : brow * mw + w] & ~used[w];
    return (__builtin_amdgcn_readfirstlane((int)(uint32_t)z) | __builtin_amdgcn_readfirstlane((int)(uint32_t)(z >> 32))) == 0;
}

template <bool W16>      // rows of 16 words (897 <= m <= 1024, 1024 threads): the circuit-level matrices
__global__ __launch_bounds__(1024) void osd0_gj_kernel(OsdGjArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: branches on it are wave-uniform for the compiler too
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + P.offAlive);        // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] column of pivot t
    uint16_t *pvrow = reinterpret_cast<uint16_t *>(lds + P.offPr);         // [m] row of pivot t
    unsigned long long *R = reinterpret_cast<unsigned long long *>(lds + P.offR);       // [16][mw] reduced columns of the block in flight
    unsigned long long *Cb = R + kGjBlock * mw;                                          // [2][16][mw] composite masks, by block parity
    unsigned long long *usedw = reinterpret_cast<unsigned long long *>(lds + P.offUsed); // [2][16] rows that have pivoted, by block parity
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [1] pivot mask, [2] anydep, [3] next c; [4..] cols[16], opp[16]
    int *bcolb = blk + 4, *oppb = bcolb + 2 * kGjBlock;                   // bcolb [2][16]: the block's columns (positions in the chunk), double-buffered; oppb [2][16]: pivot rows, by block parity
    uint32_t *selb = reinterpret_cast<uint32_t *>(oppb + 2 * kGjBlock);   // [16] pending operations a column of the block still needs
    int *s_item = reinterpret_cast<int *>(selb + kGjBlock), *sbar = s_item + 1, *nbb = s_item + 2;      // sbar: arrivals at the barrier of the waves 1 .. 15; nbb [2]: block sizes
    int *tcnt = s_item + 4;                                               // rows the pending block touches (length of tlist)
    uint32_t *tlist = reinterpret_cast<uint32_t *>(lds + P.offTl);        // [m + 2] row | selector << 16
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;
    auto uix = [&](int q, int w) -> int { return W16 ? q * 16 + (w ^ ((q >> 3) & 14)) : q * mw + w; };      // (uswz, osd_common.h)

    const int total = *P.count;
    const ClkStamp clk0 = clk_begin(P.clk);
    for (;;) {
        if (tid == 0) *s_item = atomicAdd(P.queue, 1);
        __syncthreads();
        const int item = *s_item;
        if (item >= total) break;
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        const long long t_start = OSD_CLOCK();
        int sorted_upto = n;                                                 // ordw [0 .. sorted_upto) is in order, the rest in index order
        if (!P.ordering) {                                                   // column order: ascending |llr| (osd.py:11-12), ties by index
            unsigned long long *keys = reinterpret_cast<unsigned long long *>(lds);
            uint16_t *pa = reinterpret_cast<uint16_t *>(lds + (size_t)n * 8), *pb = pa + n;
            unsigned *cnt = reinterpret_cast<unsigned *>(lds + (((size_t)n * 12 + 15) & ~(size_t)15));
            sorted_upto = osd_radix_sort_head(llr, n, P.presort, keys, pa, pb, cnt, ordw);     // (the head of the order; the sweep rarely gets past it)
        }
        const long long t_head = OSD_CLOCK();
        // ---- init: T = I, b = s + H hard (osd.py:8-9) ----
        for (int t = tid; t < (m + 2) * mw; t += T) U[t] = 0ull;
        if (tid < 32) {                                                      // rows >= m of the last word never pivot
            const int w = tid & 15;
            usedw[tid] = (w >= mw) ? ~0ull : ((w == mw - 1 && (m & 63)) ? (~0ull << (m & 63)) : 0ull);
        }
        if (tid == 0) { *sbar = 0; *tcnt = 0; }
        if (tid < 16) selb[tid] = 0u;
        // the hard decision as LDS bytes (in the chunk's support table, idle until the first chunk): the row sums below gather ~35 of them per row
        uint8_t *hstage = reinterpret_cast<uint8_t *>(colrows);
        const bool staged = n <= K * cd * 2;
        if (staged) for (int j = tid; j < n; j += T) hstage[j] = (uint8_t)hard[j];
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            U[uix(r, r >> 6)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            if (P.ell_col) {                                                 // slot-major row view: eight column loads, then eight loads of hard, in flight
                const int deg = P.deg_of_row[r];
                for (int k0 = 0; k0 < deg; k0 += 8) {
                    int cj[8];
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++) cj[j2] = (k0 + j2 < deg) ? (int)P.ell_col[(size_t)(k0 + j2) * m + r] : -1;
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++) sy ^= (cj[j2] >= 0) ? ((staged ? (int)hstage[cj[j2]] : (int)hard[cj[j2]]) & 1) : 0;
                }
            } else {
                for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= (staged ? (int)hstage[P.indices[e]] : (int)hard[P.indices[e]]) & 1;
            }
            if (sy) atomicOr(&U[uix(brow, r >> 6)], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0, cb = 0, ub = 0, bi = 0;                                         // pivots so far; buffer of the next block's masks / of the used rows in force
        uint32_t pend = 0u;                                                  // pivot mask of the block whose operations U has not seen yet
        bool kill_due = false;
        int sbar_target = 0;                                                 // (the partial barrier counts up for the whole shot)
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_p3 = 0, c_kill = 0, c_own = 0, c_col = 0, c_gat = 0, c_p3own = 0;
        (void)c_own; (void)c_col; (void)c_gat; (void)c_p3own;
        const long long t_sorted = OSD_CLOCK();
        bool finished = (P.rankH == 0) || gj_residual_gone<W16>(U, usedw, brow, mw);       // (the hard decision already reproduces the syndrome)
        for (int base = 0; base < n && !finished; base += K) {
            const int L = min(K, n - base);
            d_chunks++;
            if (base + L > sorted_upto) {                                    // past the sorted head (rare): the other columns' order, scratch in global memory
                unsigned long long *gk = P.sortws + (size_t)blockIdx.x * P.sortws_words;
                uint16_t *gpa = reinterpret_cast<uint16_t *>(gk + n), *gpb = gpa + n;
                osd_sort_rest(llr, n, sorted_upto, gk, gpa, gpb, reinterpret_cast<unsigned *>(gk + n + (n + 3) / 2), ordw);
                sorted_upto = n;
            }
            for (int c = tid; c < L; c += T) {
                sidx[c] = P.ordering ? (uint16_t)P.ordering[shot * n + base + c] : ordw[base + c];
                alive[c] = 1;
            }
            if (tid == 0) blk[3] = 0;
            __syncthreads();
            for (int t = tid; t < L * cd; t += T) {                          // supports of the chunk's columns -> LDS
                const int c = t / cd, d = t - c * cd, j = sidx[c];
                const int k = P.colptr[j] + d;
                colrows[t] = (k < P.colptr[j + 1]) ? (uint16_t)P.rowidx[k] : (uint16_t)m;          // row m of U is all zero
            }
            __syncthreads();
            // drops every still-alive column in [c0, c1) of the chunk that lies in the span of the pivots so far.  Four lanes (a quad) per column,
            // lane g of the quad taking the words g, g + 4, ..: the threads t0 = 0 .. tcount - 1 (tcount a multiple of 4, whole quads) take part.
            // `used` must be the buffer that matches the state of U.
            auto kill_pass = [&](int c0, int c1, int t0, int tcount, const unsigned long long *used) {
                const int g4 = t0 & 3;
                for (int c2 = c0 + (t0 >> 2); c2 < c1; c2 += tcount >> 2) {
                    if (!alive[c2]) continue;                                // (the same for the four lanes of the quad)
                    const uint16_t *cr2 = colrows + c2 * cd;
                    int rr[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr2[d] : m;
                    unsigned long long any = 0ull;
                    for (int w = g4; w < mw; w += 4) {
                        const unsigned long long lv = ~used[w];
                        if (lv == 0ull) continue;                            // every row of this word has pivoted
                        unsigned long long xs[8];
#pragma unroll
                        for (int d = 0; d < 8; d++) xs[d] = (d < cd) ? U[uix(rr[d], w)] : 0ull;
                        unsigned long long x = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                        for (int d = 8; d < cd; d++) x ^= U[uix(cr2[d], w)];
                        any |= x & lv;
                    }
                    int f = (any != 0ull) ? 1 : 0;                           // OR over the quad
                    f |= __builtin_amdgcn_update_dpp(0, f, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
                    f |= __builtin_amdgcn_update_dpp(0, f, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
                    if (!f && g4 == 0) alive[c2] = 0;
                }
            };
            // U lags the pivots by the pending block: the set of used rows that matches it is the one the pending block's chain started from
            if (row > 0) {                                                   // a fresh chunk late in the sweep is mostly dependent columns
                long long tk = OSD_CLOCK();
                d_kills++;
                kill_pass(0, L, tid, T, usedw + 16 * (pend ? (ub ^ 1) : ub));      // (the whole chunk: a window here lets 15 % more dependent columns into the chains)
                __syncthreads();
                c_kill += OSD_CLOCK() - tk;
            }
            // one wave gathers the next (up to 16) alive columns of the chunk from blk[3] on into block buffer `to` (ballot scan)
            auto collect_block = [&](int to) {
                const int lane = tid & 63;
                int *bc = bcolb + 16 * to;
                int c = blk[3], nbc = 0;
                while (c < L && nbc < kGjBlock) {
                    const int cc = c + lane;
                    const bool al = (cc < L) && alive[cc];
                    const unsigned long long bal = __ballot(al);
                    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (al && nbc + before < kGjBlock) bc[nbc + before] = cc;
                    const int got = __builtin_popcountll(bal);
                    if (nbc + got >= kGjBlock) {                              // stop right behind the column that filled the block
                        int need = kGjBlock - nbc;
                        unsigned long long bb = bal;
                        int lastpos = 0;
                        while (need-- > 0) { lastpos = __builtin_ctzll(bb); bb &= bb - 1; }
                        c += lastpos + 1; nbc = kGjBlock;
                    } else { nbc += got; c += 64; }
                }
                if (c > L) c = L;
                if (lane == 0) { nbb[to] = nbc; blk[3] = c; }
            };
            {
                const long long tc = OSD_CLOCK();
                if (wave == 0) collect_block(bi);
                __syncthreads();
                c_col += OSD_CLOCK() - tc;
            }
            while (true) {
                const int nb = nbb[bi];
                const int *bcol = bcolb + 16 * bi;
                if (nb == 0) break;
                d_blocks++; d_cols += nb;
                long long tp = OSD_CLOCK();
                const int cprev = cb ^ 1;                                    // buffers of the pending block (operations known, U not yet updated)
                // ---- (B) reduced columns from the LAGGING U, R[t] = XOR of U rows, plus which pending operations each still needs (bit pp_k of the
                //      column: the pending block's transform on a column is r ^= XOR_{k : r[pp_k]} C_k, tested on the lagging r).  The waves without
                //      a column run a due dependent-column test on the same consistent state. ----
                int ppvPrev = oppb[16 * cprev + (tid & 15)];                 // pending column k's pivot row sits in lane k of every 16
                asm volatile("" : "+v"(ppvPrev));
                if (tid < nb * mw) {
                    const int t = tid / mw, w = tid - t * mw;
                    const uint16_t *cr = colrows + bcol[t] * cd;
                    int rr[8];
                    unsigned long long xs[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;
#pragma unroll
                    for (int d = 0; d < 8; d++) xs[d] = U[uix(rr[d], w)];
                    unsigned long long acc = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                    for (int d = 8; d < cd; d++) acc ^= U[uix(cr[d], w)];
                    R[t * mw + w] = acc;
                    uint32_t mysel = 0u;
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int pk = __builtin_amdgcn_readlane(ppvPrev, k);
                        mysel |= ((pk >> 6) == w && ((acc >> (pk & 63)) & 1ull)) ? (1u << k) : 0u;
                    }
                    mysel &= pend;
                    if (mysel) atomicOr(&selb[t], mysel);
                }
                __syncthreads();
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- (C) wave 0: the pending operations applied to the block's columns, then the block's pivots and composite masks on
                //      registers; the other waves: the pending block's row updates U[q] ^= XOR_{k : bit pp_k of U[q]} C_k (and b) ----
                if (wave == 0) {
                    const int lane = tid, g = lane & 3, w = lane >> 2;
                    __builtin_amdgcn_s_setprio(3);                           // (three row-update waves share this SIMD)
                    GjBlock S;
#pragma unroll
                    for (int i = 0; i < 4; i++) S.X[i] = (4 * i + g < nb && w < mw) ? R[(4 * i + g) * mw + w] : 0ull;
                    if (pend) {
                        uint32_t sl4[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) sl4[i] = (4 * i + g < nb && w < mw) ? selb[4 * i + g] : 0u;
                        if (lane < 16) selb[lane] = 0u;                      // (read above; refilled by the next block's column work)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            uint32_t sl = sl4[i];
                            while (sl != 0u) {
                                const int k = __builtin_ctz(sl);
                                sl &= sl - 1u;
                                S.X[i] ^= Cb[(16 * cprev + k) * mw + w];
                            }
                        }
                    }
                    S.live = ~usedw[16 * ub + w];
                    S.nops = 0; S.maxops = P.rankH - row; S.depmask = 0u; S.pivmask = 0u; S.oppv = 0;
                    {
#define QLDPC_GSTEP(TT) if (TT < nb && S.nops < S.maxops) gj_pivot_step<TT>(S, lane);
                        QLDPC_GSTEP(0) QLDPC_GSTEP(1) QLDPC_GSTEP(2) QLDPC_GSTEP(3) QLDPC_GSTEP(4) QLDPC_GSTEP(5) QLDPC_GSTEP(6) QLDPC_GSTEP(7)
                        QLDPC_GSTEP(8) QLDPC_GSTEP(9) QLDPC_GSTEP(10) QLDPC_GSTEP(11) QLDPC_GSTEP(12) QLDPC_GSTEP(13) QLDPC_GSTEP(14) QLDPC_GSTEP(15)
#undef QLDPC_GSTEP
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) if (4 * i + g < nb && w < mw) Cb[(16 * cb + 4 * i + g) * mw + w] = S.X[i];
                    if (g == 0) usedw[16 * (ub ^ 1) + w] = ~S.live;
                    if (lane < 16) oppb[16 * cb + lane] = S.oppv;
                    if (lane < 16 && ((S.pivmask >> lane) & 1u)) {
                        const int t = row + __builtin_popcount(S.pivmask & ((1u << lane) - 1u));
                        pvcol[t] = sidx[bcol[lane]]; pvrow[t] = (uint16_t)S.oppv;
                    }
                    if (lane < nb && ((S.depmask >> lane) & 1u)) alive[bcol[lane]] = 0;
                    if (lane == 0) { blk[1] = (int)S.pivmask; blk[2] = (S.depmask != 0u) ? 1 : 0; }
                    __builtin_amdgcn_s_setprio(0);
#ifdef QLDPC_OSD_TIMERS
                    c_own += OSD_CLOCK() - tp;
#endif
                } else if (pend) {
                    const int ppv = ppvPrev;
                    const unsigned long long *Cp = Cb + 16 * cprev * mw;
                    if (W16) {
                        // Round 4: a row first only TESTS -- it reads the dwords that hold its 16 tested bits, not its 128 bytes (the round-3 form read every
                        // row into 32 registers and took the tested dwords out with wave-uniform register indices: s_set_gpr_idx mode switches, 1890 of the
                        // 4570 cycles of a pass).  How many rows a block touches swings over a shot: next to none while U is still close to the identity
                        // (3.5 % of all (row, operation) pairs over a shot), about every row near the end, when the reduced columns are dense.  So per wave:
                        //   * few rows of the wave hit (< kGjDenseLanes): they go on a list; behind a barrier of the workers the listed rows are updated by
                        //     16 lanes each (lane = word: XOR of the selected masks, one read-modify-write of the row's word);
                        //   * many hit: every lane updates its own row on the spot (eight 16-byte reads, the visited masks under lane masks, one write-back)
                        //     -- an update touches nobody else's row, so it need not wait for the other rows' tests.
                        const int lane = tid & 63;
                        int pk[16];
#pragma unroll
                        for (int k = 0; k < 16; k++) pk[k] = __builtin_amdgcn_readlane(ppv, k);
                        for (int v = wave - 1; v < 16; v += 15) {            // waves 1 .. 15 stand for the row groups v = 0 .. 14, wave 1 also takes v = 15
                            const int q = (lane << 4) + ((v + lane) & 15);
                            const bool act = (q < m + 2) && (q != m);
                            const int qq = act ? q : m;
                            const uint32_t *row32 = reinterpret_cast<const uint32_t *>(U + qq * 16);
                            const int sz = (qq >> 3) & 14;
                            uint32_t dw[16];
#pragma unroll
                            for (int k = 0; k < 16; k++) {                   // (pivot rows of operations that do not exist are 0: a harmless read)
                                const int d = pk[k] >> 5;
                                dw[k] = row32[2 * ((d >> 1) ^ sz) + (d & 1)];
                            }
                            uint32_t sel = 0u;
#pragma unroll
                            for (int k = 0; k < 16; k++) sel |= ((dw[k] >> (pk[k] & 31)) & 1u) << k;
                            sel &= pend;
                            const bool hit = act && sel != 0u;
                            const unsigned long long hm = __ballot(hit);
#ifdef QLDPC_OSD_TIMERS
                            c_gat += (unsigned long long)__builtin_popcount(hit ? sel : 0u);
#endif
                            if (__builtin_popcountll(hm) >= kGjDenseLanes) {
                                uint4 *Uq = reinterpret_cast<uint4 *>(U + qq * 16);
                                const int sz4 = (qq >> 4) & 7;
                                uint4 row[8];
#pragma unroll
                                for (int w = 0; w < 8; w++) row[w] = Uq[w ^ sz4];
#pragma unroll
                                for (int k = 0; k < 16; k++) {
                                    const bool mine = hit && ((sel >> k) & 1u);
                                    if (__ballot(mine) == 0ull) continue;             // nobody in the wave: scalar skip
                                    if (mine) {
                                        const uint4 *mk4 = reinterpret_cast<const uint4 *>(Cp + k * 16);
#pragma unroll
                                        for (int w = 0; w < 8; w++) { const uint4 t = mk4[w]; row[w].x ^= t.x; row[w].y ^= t.y; row[w].z ^= t.z; row[w].w ^= t.w; }
                                    }
                                }
                                if (hit) {
#pragma unroll
                                    for (int w = 0; w < 8; w++) Uq[w ^ sz4] = row[w];
                                }
                            } else if (hm != 0ull) {
                                const int first = __builtin_ctzll(hm);
                                int base = 0;
                                if (lane == first) base = atomicAdd(tcnt, __builtin_popcountll(hm));
                                base = __builtin_amdgcn_readlane(base, first);
                                if (hit) tlist[base + __builtin_popcountll(hm & ((1ull << lane) - 1ull))] = (uint32_t)q | (sel << 16);
                            }
                            if (wave != 1) break;
                        }
                        // every row has been tested on the OLD state: now the updates may land (a barrier of the waves 1 .. 15)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) atomicAdd(sbar, 1);
                        sbar_target += (T >> 6) - 1;
                        for (int spin = 0; *reinterpret_cast<volatile int *>(sbar) < sbar_target && spin < (1 << 22); spin++) __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int nt = *reinterpret_cast<volatile int *>(tcnt);
                        const int l16 = lane & 15;
                        for (int e = (wave - 1) * 4 + (lane >> 4); e < nt; e += 4 * ((T >> 6) - 1)) {
                            const uint32_t ent = reinterpret_cast<volatile uint32_t *>(tlist)[e];
                            const int q = (int)(ent & 0xFFFFu);
                            uint32_t sel = ent >> 16;
                            unsigned long long acc = 0ull;
                            while (sel != 0u) {
                                const int k = __builtin_ctz(sel);
                                sel &= sel - 1u;
                                acc ^= Cp[k * 16 + l16];
                            }
                            U[uix(q, l16)] ^= acc;
                        }
                    } else {
                        for (int qb = 0; qb < m + 2; qb += T - 64) {
                            const int q = qb + tid - 64;
                            const bool act = (q < m + 2) && (q != m);
                            gj_rows_apply<W16>(U, Cp, act ? q : m, mw, pend, ppv, tid & 63, c_gat);      // idle lanes look at the all-zero row
                        }
                    }
#ifdef QLDPC_OSD_TIMERS
                    c_p3own += OSD_CLOCK() - tp;
#endif
                }
                if (wave != 0 && kill_due) {
                    // a due dependent-column test on what the next blocks will take, once every row update of the pending block has landed (a
                    // barrier of the waves 1 .. 15 only: wave 0 is in its pivot chain).  U then matches the used rows the chain started from.
                    if (pend) {
                        if ((tid & 63) == 0) atomicAdd(sbar, 1);
                        sbar_target += (T >> 6) - 1;
                        for (int spin = 0; *reinterpret_cast<volatile int *>(sbar) < sbar_target && spin < (1 << 22); spin++) __builtin_amdgcn_s_sleep(2);   // (bounded: the waves are resident and always arrive)
                    }
                    kill_pass(blk[3], min(L, blk[3] + kGjKillWindow), tid - 64, T - 64, usedw + 16 * ub);
                }
                if (kill_due) { d_kills++; kill_due = false; }
                if (wave == 1) collect_block(bi ^ 1);                        // the next block's columns, behind the test (collecting before it, while wave 1
                                                                             // waits for the later waves' rows, lets 7 % more dependent columns into the blocks)
                __syncthreads();
                if (tid == 64) *tcnt = 0;                                    // (the next appends come behind the barrier of phase (B))
                bi ^= 1;
                pend = (uint32_t)blk[1];                                     // columns of the block that pivoted: its operations are pending now
                const int anydep = blk[2];
                row += __builtin_popcount(pend);
                cb ^= 1; ub ^= 1;
                c_p2 += OSD_CLOCK() - tp;
                if (row >= P.rankH || row >= m) { finished = true; break; }
                // The residual syndrome is gone: b has no one left in a row that has not pivoted (U, and b with it, stands behind the block BEFORE this one; so do the
                // used rows in the other buffer).  A later pivot row then holds b = 0: its operation adds nothing to b and its column gets e = 0 -- b, and with it
                // every e, is final, this block's pivots included.  The sweep ends here with the reference's answer (osd.py:19-25 reads b at the pivot rows only);
                // on the circuit-level matrices that is after ~150 of the ~970 pivots.
                if (gj_residual_gone<W16>(U, usedw + 16 * (ub ^ 1), brow, mw)) { pend = 0u; finished = true; break; }
                if (anydep && (d_blocks % kGjKillEvery) == 0) kill_due = true;      // done behind the next block's row updates
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        if (pend) {                                                          // the last block's row updates, all waves
            const long long tp = OSD_CLOCK();
            const int cprev = cb ^ 1;
            int ppv = oppb[16 * cprev + (tid & 15)];
            asm volatile("" : "+v"(ppv));
            for (int qb = 0; qb < m + 2; qb += T) {
                const int q = W16 ? qb + ((tid & 63) << 4) + (((tid >> 6) + tid) & 15) : qb + tid;
                const bool act = (q < m + 2) && (q != m);
                gj_rows_apply<W16>(U, Cb + 16 * cprev * mw, act ? q : m, mw, pend, ppv, tid & 63, c_gat);
            }
            pend = 0u;
            __syncthreads();
            c_p3 += OSD_CLOCK() - tp;
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2); atomicAdd(&P.dbg[11], c_p3);
            atomicAdd(&P.dbg[12], c_kill); atomicAdd(&P.dbg[14], (unsigned long long)(t_head - t_start));          // [14] the sort alone ([8] = sort + initialisation)
        }
#ifdef QLDPC_OSD_TIMERS
        if (P.dbg && tid == 0) { atomicAdd(&P.dbg[13], c_own); atomicAdd(&P.dbg[7], c_col); }       // [13] wave 0's pivot chains, [7] block collection
        if (P.dbg && (tid & 63) == 0) atomicAdd(&P.dbg[15], c_gat);                                  // [15] touched (row, operation) pairs (every wave counts its lanes)
        if (P.dbg && (tid & 127) == 0) atomicAdd(&P.dbg[24 + (tid >> 7)], c_p3own);                  // [24..31] row updates of waves 0, 2, .. 14 without the barrier
#endif
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        if (tid == 0) {                                                      // b outside the column space: a one of the reduced b in an unused row
            const unsigned long long *used = usedw + 16 * ub;
            unsigned long long bad = 0ull;
            for (int w = 0; w < mw; w++) bad |= U[uix(brow, w)] & ~used[w];
            if (bad) P.redo_list[atomicAdd(P.redo_count, 1)] = (int32_t)shot;
        }
        __syncthreads();
        for (int t = tid; t < row; t += T) {
            const int j = pvcol[t], pr = pvrow[t];
            const int8_t bbit = (int8_t)((U[uix(brow, pr >> 6)] >> (pr & 63)) & 1ull);
            sol[j] = (int8_t)((hard[j] ^ bbit) & 1);
        }
        __syncthreads();
    }
    clk_end(P.clk, clk0);
}

// handled = true when this kernel took the shots.  The shots it could not answer (right-hand side outside the column space) are left in
// g->ws_redo ([0] count, [4..] shot indices) for the reference-order kernel, which the caller launches behind this one on the same stream.
int osd0_gjq_launch(const qldpc_graph *g, const OsdGjArgs &base, int grid, hipStream_t stream, bool &launched);

int osd0_gj_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                   const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled) {
    handled = false;
    if (g->m > 1024 || g->m < 1 || g->n >= 65535 || g->n < 1) return QLDPC_OK;
    OsdGjArgs P;
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;
    size_t off = std::max((size_t)(g->m + 2) * P.mw * 8, (size_t)g->n * 12 + 16 + sort_cnt);       // U, aliased by the sort scratch
    off = (size_t)round_up((int64_t)off, 16);
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    P.offAlive = (int)off; off += (size_t)P.K;
    P.offRows = (int)off; off += (size_t)round_up((int64_t)P.K * P.cdeg * 2, 8);
    P.offPc = (int)off; off += round_up((int64_t)g->m * 2, 8);
    P.offPr = (int)off; off += round_up((int64_t)g->m * 2, 8);
    P.offR = (int)off; off += (size_t)3 * kGjBlock * P.mw * 8;
    P.offUsed = (int)off; off += 32 * 8;
    P.offBlk = (int)off; off += (4 + 5 * kGjBlock + 8) * 4;
    P.offTl = (int)off; off += (size_t)round_up((int64_t)(g->m + 2) * 4, 16);
    const size_t lds = off + 16;
    if (lds > 160 * 1024) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    const int grid = 512;
    // per workgroup: the column order in flight, and global scratch for ordering the columns behind the sorted head (keys, two index arrays, counters)
    const size_t ord_bytes = (size_t)round_up((int64_t)grid * g->n * 2 + 64, 16);
    P.sortws_words = (size_t)g->n + (size_t)(g->n + 3) / 2 + (256 * 16 + 64) / 2 + 8;
    int rc = g->ws_misc.ensure(ord_bytes + (size_t)grid * P.sortws_words * 8);
    if (rc != QLDPC_OK) return rc;
    P.sortws = reinterpret_cast<unsigned long long *>(g->ws_misc.as<unsigned char>() + ord_bytes);
    P.presort = osd_presort_choice() < 0 ? (int)round_up(std::max(g->m, 1024), 1024) : osd_presort_choice();      // (automatic: about m columns, whole chunks)
    if ((rc = g->ws_redo.ensure((size_t)(max_listed + 4) * 4)) != QLDPC_OK) return rc;
    if ((rc = g->ws_queue.ensure(16)) != QLDPC_OK) return rc;
    P.ordws = g->ws_misc.as<uint16_t>();
    P.colptr = g->d_colptr; P.rowidx = g->d_rowidx; P.indptr = g->d_indptr; P.indices = g->d_indices;
    P.ell_col = (g->d_ell_col && g->d_deg_of_row) ? g->d_ell_col : nullptr; P.deg_of_row = g->d_deg_of_row;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = g->clk_probe;
    P.dbg = osd_timer_buffer();
    P.queue = g->ws_queue.as<int>() + 3;
    P.redo_count = g->ws_redo.as<int32_t>(); P.redo_list = P.redo_count + 4;
    QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    QLDPC_HIP_TRY(hipMemsetAsync(P.redo_count, 0, 4, stream));
    const int block = (int)std::min<int64_t>(1024, round_up(std::max(g->m + 2, 256), 64));
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_gj_kernel<true>), 160 * 1024)) != QLDPC_OK) return rc;
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_gj_kernel<false>), 160 * 1024)) != QLDPC_OK) return rc;
#ifdef QLDPC_EXPERIMENTS
    if (P.mw == 16 && block == 1024 && (flags & QLDPC_FLAG_OSD_QUEUE)) {      // the look-ahead-queue form (osd_gjq.hip): measured 25 % slower (profiles/r03_experiments.txt item 12)
        bool launched = false;
        if ((rc = osd0_gjq_launch(g, P, grid, stream, launched)) != QLDPC_OK) return rc;
        if (launched) { handled = true; return QLDPC_OK; }
    }
#endif
    if (P.mw == 16 && block == 1024) hipLaunchKernelGGL(osd0_gj_kernel<true>, dim3(grid), dim3(block), lds, stream, P);
    else hipLaunchKernelGGL(osd0_gj_kernel<false>, dim3(grid), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
