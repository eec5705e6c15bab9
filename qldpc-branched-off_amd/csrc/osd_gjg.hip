// OSD-0 with free pivot rows (osd_gj.hip has the algorithm and why the reference's answer does not depend on the pivot rows) for 1024 < m <= 4096:
// the row transform U = T^T (1 MB per shot at m = 2880) lives in HBM / L2, one slab per workgroup, WORD-major (word w of row q at w * (m + 2) + q:
// a thread that owns a row walks its words, the lanes of a wave then touch neighbouring addresses).  What changes against the kernel it replaces for
// these sizes (osd0_lds_kernel<UG = true>, gf2.hip, which replays a block's 16 operations one after the other per row and reads / writes a touched row
// through L2 per OPERATION -- 71 % of its 51 M cycles per shot):
//   * the operations of a block are order-free (U[q] ^= XOR_{k : bit pp_k of U[q]} C_k, bits tested on the old row): a row owner reads its 16 tested
//     words in one round trip, accumulates the selected masks from LDS, and reads / writes the row ONCE per block, 16 words in flight;
//   * the block's pivots and composite masks come from ONE wave on registers, lane = word (rows of up to 64 words), sixteen columns in sixteen
//     register pairs: a pivot row is cleared from every other column of the block by a scalar test per column (only the pivot word's lane can hold the
//     bit) and a wave-wide XOR -- no barrier inside the chain (the replaced kernel: sixteen waves and a workgroup barrier per pivot);
//   * dependent-column tests on a window of the next columns only, four lanes per column, only words with unused rows.
// Right-hand sides outside the column space go on the list the reference-order kernel solves afterwards, as for m <= 1024.
#include "osd_gj.h"

#include <algorithm>

namespace qldpc {

struct OsdGjgArgs {
    OsdGjArgs A;
    unsigned long long *ug;        // [grid][(m + 2) * mw] row transform, word-major
    unsigned long long *ugkeys;    // [grid][n + (n + 1) / 2] sort keys + two index arrays
    int offAlive, offSort;
};

struct GjgBlock {
    unsigned long long X[16];      // the lane's word of the block's sixteen columns
    unsigned long long live;       // rows of the lane's word that have not pivoted
    int nops, maxops;
    uint32_t depmask, pivmask;
    int oppv;                      // lane t: pivot row of column t
};

template <int T>
__device__ __forceinline__ void gjg_pivot_step(GjgBlock &S, int lane) {
    const unsigned long long mword = S.X[T] & S.live;
    const unsigned long long bal = __ballot(mword != 0ull);
    if (bal == 0ull) { S.depmask |= 1u << T; return; }                                      // in the span of the pivots so far
    const int wp = __builtin_ctzll(bal);
    const unsigned long long pword = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), wp) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, wp);
    const int pb = __builtin_ctzll(pword), pp = wp * 64 + pb;
    const unsigned long long pl = (lane == wp) ? (1ull << pb) : 0ull;
    const unsigned long long rm = S.X[T] & ~pl;                                             // the column without its pivot bit
#pragma unroll
    for (int c = 0; c < 16; c++) {
        if (c == T) continue;
        if (__ballot((S.X[c] & pl) != 0ull) != 0ull) S.X[c] ^= rm;                          // (scalar test: only lane wp can hold the bit)
    }
    S.X[T] = rm;
    S.live &= ~pl;
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(S.oppv) : "s"(pp), "n"(T));
    S.pivmask |= 1u << T;
    S.nops++;
}

// b (row `brow` of the word-major U, in global memory) has no one in a row outside `used` (LDS): every wave evaluates it for itself, lane = word
__device__ __forceinline__ bool gjg_residual_gone(const unsigned long long *U, const unsigned long long *used, int brow, int ms, int mw, int lane) {
    const unsigned long long z = (lane < mw) ? (U[(size_t)lane * ms + brow] & ~used[lane]) : 0ull;
    return __ballot(z != 0ull) == 0ull;
}

__global__ __launch_bounds__(1024) void osd0_gjg_kernel(OsdGjgArgs PP) {
    extern __shared__ unsigned char lds[];
    const OsdGjArgs &P = PP.A;
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ms = m + 2;                                                  // stride of a word plane
    unsigned long long *U = PP.ug + (size_t)blockIdx.x * (size_t)ms * mw;
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + PP.offAlive);       // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] column of pivot t
    uint16_t *pvrow = reinterpret_cast<uint16_t *>(lds + P.offPr);         // [m] row of pivot t
    unsigned long long *R = reinterpret_cast<unsigned long long *>(lds + P.offR);        // [16][mw] reduced columns -> composite masks
    unsigned long long *usedw = reinterpret_cast<unsigned long long *>(lds + P.offUsed); // [2][64] rows that have pivoted, by block parity
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [1] pivot mask, [2] anydep, [3] next c; [4..] cols[16], opp[16]
    int *bcol = blk + 4, *opp = bcol + kGjBlock;
    int *s_item = opp + kGjBlock;
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;
    auto uix = [&](int q, int w) -> size_t { return (size_t)w * ms + q; };

    const int total = *P.count;
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
        unsigned long long *skeys = PP.ugkeys + (size_t)blockIdx.x * (size_t)(n + (n + 1) / 2);       // sort scratch (global): keys, two index arrays
        uint16_t *spa = reinterpret_cast<uint16_t *>(skeys + n), *spb = spa + n;
        int sorted_upto = n;                                                 // ordw [0 .. sorted_upto) is in order, the rest in index order
        if (!P.ordering)                                                     // column order: ascending |llr| (osd.py:11-12), ties by index -- its head (osd_common.h)
            sorted_upto = osd_radix_sort_head(llr, n, P.presort, skeys, spa, spb, reinterpret_cast<unsigned *>(lds + PP.offSort), ordw);
        // ---- init: T = I, b = s + H hard (osd.py:8-9) ----
        for (size_t t = tid; t < (size_t)ms * mw; t += T) U[t] = 0ull;
        if (tid < 128) {                                                     // rows >= m never pivot
            const int w = tid & 63;
            usedw[tid] = (w >= mw) ? ~0ull : ((w == mw - 1 && (m & 63)) ? (~0ull << (m & 63)) : 0ull);
        }
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            U[uix(r, r >> 6)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            if (sy) atomicOr(&U[uix(brow, r >> 6)], 1ull << (r & 63));
        }
        __threadfence_block();
        __syncthreads();
        int row = 0, par = 0;                                                // pivots so far; parity of the block count (usedw buffer in force)
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_p3 = 0, c_kill = 0;
        const long long t_sorted = OSD_CLOCK();
        bool finished = (P.rankH == 0) || gjg_residual_gone(U, usedw + 64 * par, brow, ms, mw, lane);
        for (int base = 0; base < n && !finished; base += K) {
            const int L = min(K, n - base);
            d_chunks++;
            if (base + L > sorted_upto) {                                    // past the sorted head: the other columns' order now
                osd_sort_rest(llr, n, sorted_upto, skeys, spa, spb, reinterpret_cast<unsigned *>(lds + PP.offSort), ordw);
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
            // drops every still-alive column in [c0, c1) that lies in the span of the pivots so far: four lanes (a quad) per column, lane g of the quad
            // takes the words g, g + 4, .., four words (24 loads) in flight; the threads t0 = 0 .. tcount - 1 (whole quads) take part
            auto kill_pass = [&](int c0, int c1, int t0, int tcount, const unsigned long long *used) {
                const int g4 = t0 & 3;
                for (int c2 = c0 + (t0 >> 2); c2 < c1; c2 += tcount >> 2) {
                    if (!alive[c2]) continue;
                    const uint16_t *cr2 = colrows + c2 * cd;
                    int rr[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr2[d] : m;
                    unsigned long long any = 0ull;
                    for (int w0 = g4; w0 < mw; w0 += 16) {
                        unsigned long long lv[4], xs[4][8];
#pragma unroll
                        for (int j = 0; j < 4; j++) lv[j] = (w0 + 4 * j < mw) ? ~used[w0 + 4 * j] : 0ull;
#pragma unroll
                        for (int j = 0; j < 4; j++)
#pragma unroll
                            for (int d = 0; d < 8; d++) xs[j][d] = (d < cd && lv[j] != 0ull) ? U[uix(rr[d], w0 + 4 * j)] : 0ull;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            unsigned long long x = ((xs[j][0] ^ xs[j][1]) ^ (xs[j][2] ^ xs[j][3])) ^ ((xs[j][4] ^ xs[j][5]) ^ (xs[j][6] ^ xs[j][7]));
                            if (lv[j] != 0ull) for (int d = 8; d < cd; d++) x ^= U[uix(cr2[d], w0 + 4 * j)];
                            any |= x & lv[j];
                        }
                    }
                    int f = (any != 0ull) ? 1 : 0;                           // OR over the quad
                    f |= __builtin_amdgcn_update_dpp(0, f, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
                    f |= __builtin_amdgcn_update_dpp(0, f, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
                    if (!f && g4 == 0) alive[c2] = 0;
                }
            };
            bool kill_due = false;
            if (row > 0) {                                                   // a fresh chunk late in the sweep is mostly dependent columns
                long long tk = OSD_CLOCK();
                d_kills++;
                kill_pass(0, L, tid, T, usedw + 64 * par);                  // (the whole chunk: a window here lets 25 % more dependent columns into the chains)
                __syncthreads();
                c_kill += OSD_CLOCK() - tk;
            }
            while (true) {
                if (wave == 0) {                                             // wave 0 collects the next alive columns of the chunk (ballot scan)
                    int c = blk[3], nbc = 0;
                    while (c < L && nbc < kGjBlock) {
                        const int cc = c + lane;
                        const bool al = (cc < L) && alive[cc];
                        const unsigned long long bal = __ballot(al);
                        const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                        if (al && nbc + before < kGjBlock) bcol[nbc + before] = cc;
                        const int got = __builtin_popcountll(bal);
                        if (nbc + got >= kGjBlock) {                          // stop right behind the column that filled the block
                            int need = kGjBlock - nbc;
                            unsigned long long bb = bal;
                            int lastpos = 0;
                            while (need-- > 0) { lastpos = __builtin_ctzll(bb); bb &= bb - 1; }
                            c += lastpos + 1; nbc = kGjBlock;
                        } else { nbc += got; c += 64; }
                    }
                    if (c > L) c = L;
                    if (lane == 0) { blk[0] = nbc; blk[1] = 0; blk[2] = 0; blk[3] = c; }
                }
                __syncthreads();
                const int nb = blk[0];
                if (nb == 0) break;
                d_blocks++; d_cols += nb;
                long long tp = OSD_CLOCK();
                // ---- phase 1: reduced columns R[t] = XOR of U rows ----
                for (int x = tid; x < nb * mw; x += T) {
                    const int t = x / mw, w = x - t * mw;
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
                }
                __syncthreads();
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 2: the block's pivots and composite masks, wave 0 on registers (lane = word); the other waves run a due dependent-column test ----
                const unsigned long long *used_now = usedw + 64 * par;
                unsigned long long *used_next = usedw + 64 * (par ^ 1);
                if (wave == 0) {
                    GjgBlock S;
#pragma unroll
                    for (int t = 0; t < 16; t++) S.X[t] = (t < nb && lane < mw) ? R[t * mw + lane] : 0ull;
                    S.live = ~used_now[lane];
                    S.nops = 0; S.maxops = P.rankH - row; S.depmask = 0u; S.pivmask = 0u; S.oppv = 0;
#define QLDPC_GSTEP(TT) if (TT < nb && S.nops < S.maxops) gjg_pivot_step<TT>(S, lane);
                    QLDPC_GSTEP(0) QLDPC_GSTEP(1) QLDPC_GSTEP(2) QLDPC_GSTEP(3) QLDPC_GSTEP(4) QLDPC_GSTEP(5) QLDPC_GSTEP(6) QLDPC_GSTEP(7)
                    QLDPC_GSTEP(8) QLDPC_GSTEP(9) QLDPC_GSTEP(10) QLDPC_GSTEP(11) QLDPC_GSTEP(12) QLDPC_GSTEP(13) QLDPC_GSTEP(14) QLDPC_GSTEP(15)
#undef QLDPC_GSTEP
#pragma unroll
                    for (int t = 0; t < 16; t++) if (t < nb && lane < mw) R[t * mw + lane] = S.X[t];
                    used_next[lane] = ~S.live;
                    if (lane < 16) opp[lane] = S.oppv;
                    if (lane < 16 && ((S.pivmask >> lane) & 1u)) {
                        const int t = row + __builtin_popcount(S.pivmask & ((1u << lane) - 1u));
                        pvcol[t] = sidx[bcol[lane]]; pvrow[t] = (uint16_t)S.oppv;
                    }
                    if (lane < nb && ((S.depmask >> lane) & 1u)) alive[bcol[lane]] = 0;
                    if (lane == 0) { blk[1] = (int)S.pivmask; blk[2] = (S.depmask != 0u) ? 1 : 0; }
                } else if (kill_due) {
                    kill_pass(blk[3], min(L, blk[3] + kGjKillWindow), tid - 64, T - 64, used_now);   // (what the next blocks will take)
                }
                if (kill_due) { d_kills++; kill_due = false; }
                __syncthreads();
                const uint32_t valid = (uint32_t)blk[1];                     // columns of the block that pivoted
                const int nops = __builtin_popcount(valid), anydep = blk[2];
                par ^= 1;
                c_p2 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 3: U[q] ^= XOR_{k : bit pp_k of U[q]} C_k for every row q (and for b): the 16 tested words in one round trip, the
                //      selected masks accumulated from LDS, the row read and written once, 16 words in flight ----
                if (nops > 0) {
                    int ppv = opp[tid & 15];
                    asm volatile("" : "+v"(ppv));
                    int pk[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) pk[k] = __builtin_amdgcn_readlane(ppv, k);
                    for (int qb = 0; qb < ms; qb += T) {
                        const int q = qb + tid;
                        const bool act = (q < ms) && (q != m);
                        const int qq = act ? q : m;                          // idle lanes look at the all-zero row: none of their bits is set
                        uint32_t Pw[16];                                     // the dword that holds the tested bit
                        unsigned long long mk[16], touched = 0ull;
                        const uint32_t *U32 = reinterpret_cast<const uint32_t *>(U);
#pragma unroll
                        for (int k = 0; k < 16; k++) Pw[k] = ((valid >> k) & 1u) ? U32[2 * uix(qq, pk[k] >> 6) + ((pk[k] >> 5) & 1)] : 0u;
#pragma unroll
                        for (int k = 0; k < 16; k++) { mk[k] = __ballot(((Pw[k] >> (pk[k] & 31)) & 1u) != 0u); touched |= mk[k]; }
                        if (touched == 0ull) continue;
                        const bool mine = (touched >> lane) & 1ull;
                        for (int w0 = 0; w0 < mw; w0 += 16) {
                            unsigned long long acc[16];
#pragma unroll
                            for (int j = 0; j < 16; j++) acc[j] = 0ull;
#pragma unroll
                            for (int k = 0; k < 16; k++) {
                                if (mk[k] == 0ull) continue;                  // nobody in the wave: scalar skip
                                if ((mk[k] >> lane) & 1ull) {
#pragma unroll
                                    for (int j = 0; j < 16; j++) if (w0 + j < mw) acc[j] ^= R[k * mw + w0 + j];
                                }
                            }
                            if (mine) {
                                unsigned long long u[16];
#pragma unroll
                                for (int j = 0; j < 16; j++) u[j] = (w0 + j < mw) ? U[uix(qq, w0 + j)] : 0ull;
#pragma unroll
                                for (int j = 0; j < 16; j++) if (w0 + j < mw) U[uix(qq, w0 + j)] = u[j] ^ acc[j];
                            }
                        }
                    }
                }
                row += nops;
                __threadfence_block();
                __syncthreads();
                c_p3 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                if (row >= P.rankH || row >= m) { finished = true; break; }
                // the residual syndrome is gone (osd_gj.hip: b has no one left in an unused row, so no later pivot changes b and every later column gets e = 0)
                if (gjg_residual_gone(U, usedw + 64 * par, brow, ms, mw, lane)) { finished = true; break; }
                if (anydep && (d_blocks % kGjKillEvery) == 0) kill_due = true;       // done by the idle waves beside the next block's pivot chain
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2); atomicAdd(&P.dbg[11], c_p3);
            atomicAdd(&P.dbg[12], c_kill);
        }
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        if (tid == 0) {                                                      // b outside the column space: a one of the reduced b in an unused row
            const unsigned long long *used = usedw + 64 * par;
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
}

// handled = true when this kernel took the shots; the ones it lists in g->ws_redo go through the reference-order kernel (osd0_lds_kernel<UG = true>)
int osd0_gjg_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream, size_t ws_offset, bool &handled) {
    handled = false;
    if (g->m > 4096 || g->m < 1 || g->n >= 65535 || g->n < 1) return QLDPC_OK;
    OsdGjgArgs PP;
    OsdGjArgs &P = PP.A;
    P = OsdGjArgs{};
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    P.presort = osd_presort_choice() < 0 ? (int)round_up(std::max(g->m, 1024), 1024) : osd_presort_choice();      // (automatic: about m columns, whole chunks)
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;
    size_t off = 0;
    PP.offSort = (int)off; off += sort_cnt;
    off = (size_t)round_up((int64_t)off, 16);
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    PP.offAlive = (int)off; off += (size_t)P.K;
    P.offRows = (int)off; off += (size_t)round_up((int64_t)P.K * P.cdeg * 2, 8);
    P.offPc = (int)off; off += round_up((int64_t)g->m * 2, 8);
    P.offPr = (int)off; off += round_up((int64_t)g->m * 2, 8);
    P.offR = (int)off; off += (size_t)kGjBlock * P.mw * 8;
    P.offUsed = (int)off; off += 128 * 8;
    P.offBlk = (int)off; off += (4 + 2 * kGjBlock + 4) * 4;
    const size_t lds = off + 16;
    if (lds > 160 * 1024) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    const int grid = 512;
    const size_t sz_ord = (size_t)round_up((int64_t)grid * g->n * 2 + 64, 16);
    const size_t sz_u = (size_t)grid * (size_t)(g->m + 2) * P.mw * 8;
    const size_t per_keys = (size_t)g->n + (size_t)(g->n + 1) / 2;
    const size_t sz_k = (size_t)grid * per_keys * 8;
    int rc = g->ws_misc.ensure(ws_offset + sz_ord + sz_u + sz_k);
    if (rc != QLDPC_OK) return rc;
    if ((rc = g->ws_redo.ensure((size_t)(max_listed + 4) * 4)) != QLDPC_OK) return rc;
    if ((rc = g->ws_queue.ensure(16)) != QLDPC_OK) return rc;
    unsigned char *base = g->ws_misc.as<unsigned char>() + ws_offset;
    P.ordws = reinterpret_cast<uint16_t *>(base);
    PP.ug = reinterpret_cast<unsigned long long *>(base + sz_ord);
    PP.ugkeys = reinterpret_cast<unsigned long long *>(base + sz_ord + sz_u);
    P.colptr = g->d_colptr; P.rowidx = g->d_rowidx; P.indptr = g->d_indptr; P.indices = g->d_indices;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = nullptr;
    P.dbg = osd_timer_buffer();
    P.queue = g->ws_queue.as<int>() + 3;
    P.redo_count = g->ws_redo.as<int32_t>(); P.redo_list = P.redo_count + 4;
    QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    QLDPC_HIP_TRY(hipMemsetAsync(P.redo_count, 0, 4, stream));
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_gjg_kernel), 160 * 1024)) != QLDPC_OK) return rc;
    hipLaunchKernelGGL(osd0_gjg_kernel, dim3(grid), dim3(1024), lds, stream, PP);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
