// OSD-0 (a9, reference src/decoding/osd.py:5-29 + gf2_elimination_packed_core, src/decoding/kernels.py:48-96) for matrices with
// 512 <= m <= 1024 rows: the round-1 LDS-resident Gauss-Jordan kernel (gf2.hip, osd0_lds_kernel) with its two expensive phases OVERLAPPED.
//
// Per block of 16 columns the round-1 kernel runs, one after the other,
//   phase 2: the block's pivots -- a serial chain (ballot -> scalar -> lane read -> barrier per pivot) on four "holder" waves, the other
//            twelve idle:                                                                             1.37 M cycles per shot
//   phase 3: the block's 16 operations on every row of the transform U -- all waves, LDS bound:    1.29 M cycles per shot
// and these use disjoint resources.  Here block i's row updates run on the twelve non-holder waves WHILE the holders resolve block
// i + 1, in lockstep at the holders' own per-pivot barrier: between barrier t-1 and barrier t the holders do pivot step t of block i + 1
// and everybody else applies operation t of block i to its rows.  What makes that legal:
//   * the columns of block i + 1 are reduced through U BEFORE block i's operations reach U (phase 1 sits between two barriers while U
//     is stable), so they miss those 16 operations; the holders apply them to their register copies first (the same swap-and-add they
//     apply for the pivots of their own block, kernels.py:79-92 restricted to the block) -- 16 x ~12 instructions, no barrier;
//   * dependent-column tests (kill passes) read U between blocks and therefore see the pivots of all blocks but the pending one: they
//     test against the number of pivots actually applied to U, which only makes them slightly less eager;
//   * a chunk that runs out of columns, and the end of the sweep, drain the pending block with an iteration that has no columns.
// Everything else (position-space transform U = T^T in LDS, radix sort of the reliability order, parallel dependent-column tests,
// stop at rank(H), back-fill) is the round-1 algorithm; results are identical (same pivot rows, same solution on every input).
//
// STATUS (round 2, measured on 1 x MI355X, profiles/r02_osd_experiments.txt): correct, but SLOWER than running the phases one after the
// other: 3.31 M cycles per shot for the fused loop against 1.37 M + 1.29 M.  The two phases do not use disjoint resources after all: the
// pivot chain is a chain of LDS round trips (publish the pivot, read it back, read the mask), and those queue behind the row updates,
// which keep the LDS pipeline full -- the chain's latency inflates to the row-update time instead of hiding under it; on top of that the
// pending block costs the dependent-column tests one block of eagerness (79 instead of 71 blocks per shot).  Selected only by
// QLDPC_FLAG_OSD_PIPED; the default stays the round-1 kernel.
#include "common.h"
#include "mc_common.h"
#include "osd_common.h"

#include <algorithm>

namespace qldpc {

constexpr int kPipeBlock = 16;      // columns per block: 4 holder waves x 4 columns (16 lanes = 16 words per column)
constexpr int kPipeHolders = 4;     // waves 0..3 resolve pivots, waves 4.. update rows

struct OsdPipeArgs {
    int m, n, mw, rankH, K, cdeg, nokill;
    const int32_t *indptr, *indices;
    const uint16_t *colrows_g;         // [n][cdeg] rows of every column in ascending order, padded with the zero row m
    const int32_t *list, *count;
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    uint16_t *ordws;                   // [grid][n] sorted column order of the shot in flight (global, L2-resident)
    int *queue;
    unsigned long long *clk, *dbg;
    int offIdx, offAlive, offRows, offPc, offR, offBlk;
};

__device__ __forceinline__ int pswz(int q, int w, int mw) { return q * mw + ((mw == 16) ? (w ^ (q & 15)) : w); }

// one operation (kernels.py:79-92) on one row of U: swap bits a <-> pp, then add the elimination mask if bit a is set
__device__ __forceinline__ void pipe_apply_op(unsigned long long *U, int q, int mw, int a, int pp, const unsigned long long *mk) {
    const int wa = a >> 6, wp = pp >> 6;
    const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
    const unsigned long long xa = U[pswz(q, wa, mw)];
    const unsigned long long xp = (wp == wa) ? xa : U[pswz(q, wp, mw)];
    const bool ba = (xa & abit) != 0ull, bp = (xp & pbit) != 0ull;
    if (ba != bp) {
        if (wp == wa) { U[pswz(q, wa, mw)] = xa ^ abit ^ pbit; }
        else { U[pswz(q, wa, mw)] = xa ^ abit; U[pswz(q, wp, mw)] = xp ^ pbit; }
    }
    if (bp) {                                            // bit a after the swap: add the pivot row (kernels.py:88-92)
        if (mw == 16) {                                  // all 32 reads in flight before the first XOR (a rolled loop waits per word)
            unsigned long long u[16];
#pragma unroll
            for (int w = 0; w < 16; w++) u[w] = U[q * 16 + (w ^ (q & 15))];
#pragma unroll
            for (int w = 0; w < 16; w++) u[w] ^= mk[w];
#pragma unroll
            for (int w = 0; w < 16; w++) U[q * 16 + (w ^ (q & 15))] = u[w];
        } else {
            for (int w = 0; w < mw; w++) U[pswz(q, w, mw)] ^= mk[w];
        }
    }
}

__global__ __launch_bounds__(1024) void osd0_pipe_kernel(OsdPipeArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x;
    const int wv = tid >> 6, lane = tid & 63;
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + P.offAlive);        // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] pivot t sits at position t
    unsigned long long *Rb = reinterpret_cast<unsigned long long *>(lds + P.offR);      // [2][kPipeBlock][mw] reduced columns -> masks, two blocks in flight
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [3] next c, [4] work item
    int *bcol = blk + 8;                                                   // [kPipeBlock] chunk positions of the block being pivoted
    int *ops = bcol + kPipeBlock;                                          // [2][3][kPipeBlock]: a, pp, column-in-block of each operation
    int2 *stp = reinterpret_cast<int2 *>(ops + 6 * kPipeBlock);            // [kPipeBlock] (a | -1 dependent | -2 no column, pp) of step t
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;                                                // U row that carries b
    const bool holder = wv < kPipeHolders;
    const int HT = kPipeHolders * 64;                                      // threads that do not own rows while a block is being pivoted
    // holders: lane = (grp, w): word w of column 4 * wave + grp
    const int w16 = lane & 15, grp = lane >> 4, sc = 4 * wv + grp;

    const int total = *P.count;
    const ClkStamp clk0 = clk_begin(P.clk);
    for (;;) {
        if (tid == 0) blk[4] = atomicAdd(P.queue, 1);
        __syncthreads();
        const int item = blk[4];
        if (item >= total) break;
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        const long long t_start = OSD_CLOCK();
        if (!P.ordering) {
            unsigned long long *keys = reinterpret_cast<unsigned long long *>(lds);                 // the sort scratch aliases U
            uint16_t *pa = reinterpret_cast<uint16_t *>(lds + (size_t)n * 8), *pb = pa + n;
            unsigned *cnt = reinterpret_cast<unsigned *>(lds + (((size_t)n * 12 + 15) & ~(size_t)15));
            osd_radix_sort(llr, n, keys, pa, pb, cnt, ordw);
        }
        // ---- init: T = I (positions = original rows), b = s + H hard (osd.py:8-9) ----
        for (int t = tid; t < (m + 2) * mw; t += T) U[t] = 0ull;
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            U[pswz(r, r >> 6, mw)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            if (sy) atomicOr(&U[pswz(brow, r >> 6, mw)], 1ull << (r & 63));
        }
        __syncthreads();
        int row_applied = 0;      // pivots whose operations have reached U
        int npend = 0, pbuf = 0;  // operations of the block resolved last, not yet in U (buffer pbuf); pivots found = row_applied + npend
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_kill = 0;
        const long long t_sorted = OSD_CLOCK();
        bool finished = (P.rankH == 0);
        for (int base = 0; base < n && !finished; base += K) {
            const int L = min(K, n - base);
            d_chunks++;
            for (int c = tid; c < L; c += T) {
                sidx[c] = P.ordering ? (uint16_t)P.ordering[shot * n + base + c] : ordw[base + c];
                alive[c] = 1;
            }
            if (tid == 0) blk[3] = 0;
            __syncthreads();
            for (int t = tid; t < L * cd; t += T) {                          // supports of the chunk's columns -> LDS
                const int c = t / cd, d = t - c * cd;
                colrows[t] = P.colrows_g[(size_t)sidx[c] * cd + d];          // short columns are padded with m: row m of U is all zero
            }
            __syncthreads();
            // drops every still-alive column of the chunk from c0 on that is dependent on the pivots applied to U so far
            auto kill_pass = [&](int c0) {
                const int row = row_applied, wq = row >> 6;
                for (int c2 = c0 + tid; c2 < L; c2 += T) {
                    if (!alive[c2]) continue;
                    const uint16_t *cr2 = colrows + c2 * cd;
                    int rr[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr2[d] : m;
                    unsigned long long any = 0ull;
                    for (int w = wq; w < mw; w++) {
                        unsigned long long xs[8];
#pragma unroll
                        for (int d = 0; d < 8; d++) xs[d] = U[pswz(rr[d], w, mw)];
                        unsigned long long x = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                        for (int d = 8; d < cd; d++) x ^= U[pswz(cr2[d], w, mw)];
                        any |= (w == wq) ? (x & (~0ull << (row & 63))) : x;
                    }
                    if (!any) alive[c2] = 0;
                }
            };
            if (row_applied > 0 && !P.nokill) {
                const long long tk = OSD_CLOCK();
                d_kills++;
                kill_pass(0);
                __syncthreads();
                c_kill += OSD_CLOCK() - tk;
            }
            while (true) {
                if (tid < 64) {                                              // wave 0 collects the next alive columns of the chunk (ballot scan)
                    int c = blk[3], nbc = 0;
                    while (c < L && nbc < kPipeBlock) {
                        const int cc = c + tid;
                        const bool al = (cc < L) && alive[cc];
                        const unsigned long long bal = __ballot(al);
                        const int before = __builtin_popcountll(bal & ((1ull << tid) - 1ull));
                        if (al && nbc + before < kPipeBlock) bcol[nbc + before] = cc;
                        const int got = __builtin_popcountll(bal);
                        if (nbc + got >= kPipeBlock) {
                            int need = kPipeBlock - nbc;
                            unsigned long long bb = bal;
                            int lastpos = 0;
                            while (need-- > 0) { lastpos = __builtin_ctzll(bb); bb &= bb - 1; }
                            c += lastpos + 1; nbc = kPipeBlock;
                        } else { nbc += got; c += 64; }
                    }
                    if (c > L) c = L;
                    if (tid == 0) { blk[0] = nbc; blk[3] = c; }
                }
                __syncthreads();
                const int nb = blk[0];
                if (nb == 0 && npend == 0) break;                            // chunk exhausted, nothing pending
                const int cbuf = pbuf ^ 1;
                unsigned long long *R = Rb + (size_t)cbuf * kPipeBlock * mw;             // the block being pivoted
                const unsigned long long *Rp = Rb + (size_t)pbuf * kPipeBlock * mw;      // masks of the pending block
                int *opa = ops + cbuf * 3 * kPipeBlock, *opp = opa + kPipeBlock, *opt = opp + kPipeBlock;
                const int *ppa = ops + pbuf * 3 * kPipeBlock, *ppp = ppa + kPipeBlock, *ppt = ppp + kPipeBlock;
                if (nb > 0) { d_blocks++; d_cols += nb; }
                long long tp = OSD_CLOCK();
                // ---- phase 1: reduced columns through U (which has every operation but the pending block's) ----
                for (int x = tid; x < nb * mw; x += T) {
                    const int t = x / mw, w = x - t * mw;
                    const uint16_t *cr = colrows + bcol[t] * cd;
                    int rr[8];
                    unsigned long long xs[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;
#pragma unroll
                    for (int d = 0; d < 8; d++) xs[d] = U[pswz(rr[d], w, mw)];
                    unsigned long long acc = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                    for (int d = 8; d < cd; d++) acc ^= U[pswz(cr[d], w, mw)];
                    R[t * mw + w] = acc;
                }
                __syncthreads();
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- holders: column registers, brought up to date with the pending block's operations ----
                unsigned long long X = 0ull;
                int colid = 0;
                if (holder) {
                    X = (sc < nb && w16 < mw) ? R[sc * mw + w16] : 0ull;
                    colid = (sc < nb) ? (int)sidx[bcol[sc]] : 0;
                    for (int k = 0; k < npend; k++) {
                        const int a = ppa[k], pp = ppp[k], wa = a >> 6, wp = pp >> 6;
                        const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                        const unsigned long long rmw = (w16 < mw) ? Rp[ppt[k] * mw + w16] : 0ull;
                        const unsigned long long balA = __ballot(w16 == wa && (X & abit) != 0ull), balP = __ballot(w16 == wp && (X & pbit) != 0ull);
                        const bool ba = (balA >> (grp * 16 + wa)) & 1ull, bp = (balP >> (grp * 16 + wp)) & 1ull;
                        if (ba != bp) { if (w16 == wa) X ^= abit; if (w16 == wp) X ^= pbit; }
                        if (bp) X ^= rmw;
                    }
                }
                // ---- the fused step loop: pivot step t of this block (holders) || operation t of the pending block on the rows (the rest) ----
                int lrow = row_applied + npend, nops = 0, anydep = 0;
                for (int t = 0; t < kPipeBlock; t++) {
                    if (holder) {
                        if (wv == (t >> 2)) {
                            const int gt = t & 3;
                            if (t < nb && lrow < P.rankH && lrow < m) {
                                const bool ing = (grp == gt);
                                const int wq = lrow >> 6;
                                const unsigned long long mword = (!ing || w16 < wq) ? 0ull : ((w16 == wq) ? (X & (~0ull << (lrow & 63))) : X);
                                const unsigned long long bal = (__ballot(mword != 0ull) >> (gt * 16)) & 0xFFFFull;
                                if (bal == 0ull) {                                           // dependent on the pivots so far
                                    if (lane == gt * 16) { stp[t] = make_int2(-1, 0); alive[bcol[t]] = 0; }
                                } else {
                                    const int pw = __builtin_amdgcn_readfirstlane(__builtin_ctzll(bal)), src = gt * 16 + pw;
                                    const unsigned long long pword = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mword >> 32), src) << 32) |
                                                                     (unsigned)__builtin_amdgcn_readlane((int)mword, src);
                                    const int pp = pw * 64 + __builtin_ctzll(pword), a = lrow, wa = a >> 6, wp = pp >> 6;
                                    const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                                    const bool olda = __ballot(ing && w16 == wa && (X & abit) != 0ull) != 0ull;
                                    // swap bits a <-> pp of the pivot column itself (bit pp is 1), then clear bit a: that is the elimination mask
                                    unsigned long long rm = X;
                                    if (w16 == wp) rm = olda ? (rm | pbit) : (rm & ~pbit);
                                    if (w16 == wa) rm &= ~abit;
                                    if (ing) { X = rm; if (w16 < mw) R[t * mw + w16] = rm; }
                                    if (lane == gt * 16) { stp[t] = make_int2(a, pp); opa[nops] = a; opp[nops] = pp; opt[nops] = t; pvcol[a] = (uint16_t)colid; }
                                }
                            } else if (lane == gt * 16) {
                                stp[t] = make_int2(-2, 0);                                   // no column / full rank
                            }
                        }
                    } else if (t < npend) {
                        // 768 row threads for m + 2 rows: the first 242 take two.  (Splitting those second rows by word ranges over adjacent
                        // lanes to even the load was measured SLOWER, 4.0 M vs 3.3 M cycles per shot: the row updates are LDS-throughput bound
                        // and the split adds test reads.)
                        const int a = ppa[t], pp = ppp[t];
                        const unsigned long long *mk = Rp + ppt[t] * mw;
                        for (int q = tid - HT; q < m + 2; q += T - HT)
                            if (q != m) pipe_apply_op(U, q, mw, a, pp, mk);
                    }
                    __syncthreads();
                    const int2 st = stp[t];
                    if (st.x == -1) { anydep = 1; continue; }
                    if (st.x < 0) continue;
                    if (holder && 4 * wv + 3 > t) {                                          // wave-uniform: this wave still holds a later column
                        const unsigned long long rmw = (w16 < mw) ? R[t * mw + w16] : 0ull;
                        const int a = st.x, pp = st.y, wa = a >> 6, wp = pp >> 6;
                        const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                        unsigned long long x = X;
                        const unsigned long long balA = __ballot(w16 == wa && (x & abit) != 0ull), balP = __ballot(w16 == wp && (x & pbit) != 0ull);
                        const bool ba = (balA >> (grp * 16 + wa)) & 1ull, bp = (balP >> (grp * 16 + wp)) & 1ull;
                        if (ba != bp) { if (w16 == wa) x ^= abit; if (w16 == wp) x ^= pbit; }
                        if (bp) x ^= rmw;                                                    // after the swap, bit a of the column is bp
                        if (sc > t && sc < nb) X = x;
                    }
                    nops++; lrow++;
                }
                __syncthreads();          // stp[] / ops of this block are complete; the pending block is in U
                row_applied += npend;
                npend = nops; pbuf = cbuf;
                c_p2 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                if (row_applied + npend >= P.rankH || row_applied + npend >= m) { finished = true; break; }
                // ---- dependent columns were met: drop every column of the chunk that is dependent on what U holds by now ----
                if (anydep && !P.nokill && row_applied > 0) {
                    d_kills++;
                    kill_pass(blk[3]);
                    __syncthreads();
                    c_kill += OSD_CLOCK() - tp;
                }
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        // ---- drain: the last block's operations, all threads ----
        if (npend > 0) {
            const long long tp = OSD_CLOCK();
            const int *ppa = ops + pbuf * 3 * kPipeBlock, *ppp = ppa + kPipeBlock, *ppt = ppp + kPipeBlock;
            const unsigned long long *Rp = Rb + (size_t)pbuf * kPipeBlock * mw;
            for (int q = tid; q < m + 2; q += T) {
                if (q == m) continue;
                for (int k = 0; k < npend; k++) pipe_apply_op(U, q, mw, ppa[k], ppp[k], Rp + ppt[k] * mw);
            }
            row_applied += npend; npend = 0;
            __syncthreads();
            c_p2 += OSD_CLOCK() - tp;
        }
        const int row = row_applied;
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2);
            atomicAdd(&P.dbg[12], c_kill);
        }
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < row; t += T) {
            const int j = pvcol[t];
            const int8_t bbit = (int8_t)((U[pswz(brow, t >> 6, mw)] >> (t & 63)) & 1ull);
            sol[j] = (int8_t)((hard[j] ^ bbit) & 1);
        }
        __syncthreads();
    }
    clk_end(P.clk, clk0);
}

int host_gf2_rank(const qldpc_graph *g);
int ensure_col_rows(const qldpc_graph *g);

static bool plan_osd_pipe(const qldpc_graph *g, OsdPipeArgs &P, size_t &lds) {
    if (g->m > 1024 || g->m + 2 <= 512 || g->n >= 65535) return false;       // needs the 1024-thread block: 4 holder waves + 12 row waves
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;
    size_t off = std::max((size_t)(g->m + 2) * P.mw * 8, (size_t)g->n * 12 + 16 + sort_cnt);
    off = (size_t)round_up((int64_t)off, 16);
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    P.offAlive = (int)off; off += (size_t)P.K;
    P.offRows = (int)off; off += (size_t)round_up((int64_t)P.K * P.cdeg * 2, 16);
    P.offPc = (int)off; off += (size_t)round_up((int64_t)g->m * 2, 16);
    P.offR = (int)off; off += (size_t)2 * kPipeBlock * P.mw * 8;
    P.offBlk = (int)off; off += (8 + kPipeBlock + 6 * kPipeBlock + 2 * kPipeBlock) * 4;
    lds = off + 16;
    return lds <= 160 * 1024;
}

int osd0_pipe_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                     const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled) {
    OsdPipeArgs P;
    size_t lds = 0;
    handled = false;
    if (!plan_osd_pipe(g, P, lds)) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    int rc = ensure_col_rows(g);
    if (rc != QLDPC_OK) return rc;
    P.colrows_g = g->d_col_rows;
    const int grid = 512;
    const size_t sz_ord = (size_t)round_up((int64_t)grid * g->n * 2 + 64, 256);
    if ((rc = g->ws_misc.ensure(sz_ord)) != QLDPC_OK) return rc;
    P.ordws = g->ws_misc.as<uint16_t>();
    P.indptr = g->d_indptr; P.indices = g->d_indices;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = g->clk_probe;
    P.dbg = osd_timer_buffer();
    P.nokill = (flags & QLDPC_FLAG_OSD_NOKILL) ? 1 : 0;
    if ((rc = g->ws_queue.ensure(16)) != QLDPC_OK) return rc;
    P.queue = g->ws_queue.as<int>() + 2;
    QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_pipe_kernel), 160 * 1024)) != QLDPC_OK) return rc;
    hipLaunchKernelGGL(osd0_pipe_kernel, dim3(grid), dim3(1024), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
