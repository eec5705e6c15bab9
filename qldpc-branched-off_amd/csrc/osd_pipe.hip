// OSD-0 (a9, reference src/decoding/osd.py:5-29 + gf2_elimination_packed_core, src/decoding/kernels.py:48-96) for matrices with
// m <= 1024 rows: the LDS-resident Gauss-Jordan kernel of gf2.hip (osd0_lds_kernel) with its two expensive phases OVERLAPPED.
//
// Per block of 16 columns the gf2.hip kernel runs, one after the other,
//   phase 2: the block's pivots -- a serial chain in ONE wave on registers (osd_common.h: quad_pivot_step), the other waves idle;
//   phase 3: the block's <= 16 operations on every row of the transform U -- all waves, instruction-issue bound (osd_rows_apply).
// Here the row updates of block i run WHILE wave 0 resolves block i + 1 (and wave 0 does its own share of the rows afterwards).  What makes
// that legal:
//   * the columns of block i + 1 are reduced through U (phase 1) while U is stable, BEFORE block i's operations reach it, so they miss those
//     <= 16 operations; wave 0 applies them to the 16 column vectors first -- a column vector transforms exactly like a row of U = T^T, so
//     this is osd_rows_apply on 16 "rows" that live in the block buffer -- and only then starts the chain.  The chain itself touches no LDS
//     besides its own block buffer, so it does not queue behind the row updates (the round-2 attempt with the round-1 chain did);
//   * dependent-column tests (kill passes) read U in the same stable window and therefore see the pivots of all blocks but the pending
//     one: they test against the number of pivots actually applied to U, which only makes them slightly less eager;
//   * a chunk that runs out of columns, and the end of the sweep, drain the pending block.
// Everything else (position-space transform U = T^T in LDS, radix sort of the reliability order, stop at rank(H), back-fill) is the gf2.hip
// algorithm; results are identical (same pivot rows, same solution on every input).
//
// STATUS (round 2, measured on 1 x MI355X, profiles/r02_osd_experiments.txt): correct, and exactly as fast as the serial phases (2.80 vs 2.78 M
// cycles per shot on the circuit-level matrices), so it is selected only by QLDPC_FLAG_OSD_PIPED.  Per block the fork costs 20.6 k cycles where the
// serial kernel spends 11.9 k (chain) + 15.0 k (rows): wave 0's path is 6.7 k (pending block onto the 16 new columns) + 9.8 k (chain), its own
// 64 rows cost another full pass whoever takes them (a pass is ~9.5 k cycles of instruction issue almost independent of the number of rows in it),
// the dependent-column tests need their own window again (0.31 M) and see the transform two blocks late (92 blocks per shot instead of 79).
#include "common.h"
#include "mc_common.h"
#include "osd_common.h"

#include <algorithm>

namespace qldpc {

constexpr int kPipeBlock = 16;      // columns per block (4 per lane of the resolving wave)

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

__global__ __launch_bounds__(1024) void osd0_pipe_kernel(OsdPipeArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + P.offAlive);        // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] pivot t sits at position t
    unsigned long long *Rb = reinterpret_cast<unsigned long long *>(lds + P.offR);      // [2][kPipeBlock][mw] reduced columns -> masks, two blocks in flight
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [1] pivots of the block just resolved, [2] it met dependent columns, [3] next c, [4] work item
    int *bcol = blk + 8;                                                   // [kPipeBlock] chunk positions of the block being pivoted
    int *ops = bcol + kPipeBlock;                                          // [2][2][kPipeBlock]: pp, column-in-block of each operation
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;                                                // U row that carries b
    const bool strided = (mw == 16);                                       // rows to threads as in gf2.hip (mw == 16 implies 1024 threads)
    auto rowq = [&](int qb) { return strided ? qb + ((tid & 63) << 4) + (((tid >> 6) + tid) & 15) : qb + tid; };
    unsigned long long d_wops = 0, d_lops = 0;
    (void)d_wops; (void)d_lops;

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
            U[uswz(r, r >> 6, mw)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            if (sy) atomicOr(&U[uswz(brow, r >> 6, mw)], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0;              // pivots whose operations have reached U
        int pend = 0, pbuf = 0;   // operations of the block resolved last, not yet in U (buffer pbuf); pivots found = row + pend
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_kill = 0, c_po = 0, c_chain = 0, c_own = 0;
        (void)c_po; (void)c_chain; (void)c_own;
        const long long t_sorted = OSD_CLOCK();
        bool finished = (P.rankH == 0);
        // the pending block's operations on the rows of U this thread owns.  beside_chain: wave 0 is busy with the next block's pivots, its rows
        // are taken by waves 1-3 in a second pass (a third each; dealing them to all fifteen waves was measured slower: a pass costs nearly the
        // same whatever the number of rows in it)
        auto apply_pending = [&](bool beside_chain) {
            const int *pp_p = ops + pbuf * 2 * kPipeBlock, *pt_p = pp_p + kPipeBlock;
            int ppv = pp_p[tid & 15], ptv = pt_p[tid & 15];                 // operation k's (pp, column) sit in lane k of every 16
            asm volatile("" : "+v"(ppv), "+v"(ptv));
            const unsigned long long *Rp = Rb + (size_t)pbuf * kPipeBlock * mw;
            const int wave = tid >> 6;
            if (beside_chain && wave == 0) return;
            for (int qb = 0; qb < m + 2; qb += T) {
                const int q = rowq(qb);
                const bool act = (q < m + 2) && (q != m);
                const int qq = act ? q : m;                                  // idle lanes look at the all-zero row
                osd_rows_apply(U + qq * mw, (mw == 16) ? ((qq >> 3) & 14) : 0, act, row, pend, mw, ppv, ptv, Rp, tid, d_wops, d_lops);
                if (beside_chain && wave <= 3) {                             // wave 0's rows of this round: lanes 22 (wave - 1) .. 22 wave - 1 of it
                    const int l0 = 22 * (wave - 1) + lane, t0 = l0;           // thread index the row would have had in wave 0
                    const bool mine = (lane < 22) && (l0 < 64);
                    const int q0 = strided ? qb + (t0 << 4) + (t0 & 15) : qb + t0;
                    const bool act0 = mine && (q0 < m + 2) && (q0 != m);
                    const int qq0 = act0 ? q0 : m;
                    osd_rows_apply(U + qq0 * mw, (mw == 16) ? ((qq0 >> 3) & 14) : 0, act0, row, pend, mw, ppv, ptv, Rp, tid, d_wops, d_lops);
                }
            }
        };
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
                const int wq = row >> 6;
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
                        for (int d = 0; d < 8; d++) xs[d] = U[uswz(rr[d], w, mw)];
                        unsigned long long x = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                        for (int d = 8; d < cd; d++) x ^= U[uswz(cr2[d], w, mw)];
                        any |= (w == wq) ? (x & (~0ull << (row & 63))) : x;
                    }
                    if (!any) alive[c2] = 0;
                }
            };
            bool kill_due = (row > 0) && !P.nokill;                          // a fresh chunk late in the sweep is mostly dependent columns
            int kill_from = 0;
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
                if (nb == 0 && pend == 0) break;                             // chunk exhausted, nothing pending
                const int cbuf = pbuf ^ 1;
                unsigned long long *Rn = Rb + (size_t)cbuf * kPipeBlock * mw;            // the block being pivoted
                int *pp_n = ops + cbuf * 2 * kPipeBlock, *pt_n = pp_n + kPipeBlock;
                if (nb > 0) { d_blocks++; d_cols += nb; }
                long long tp = OSD_CLOCK();
                // ---- U is stable here: phase 1 (reduced columns of the new block through U, which has every operation but the pending
                // block's) and the dependent-column tests an earlier block asked for ----
                for (int x = tid; x < nb * mw; x += T) {
                    const int t = x / mw, w = x - t * mw;
                    const uint16_t *cr = colrows + bcol[t] * cd;
                    int rr[8];
                    unsigned long long xs[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;
#pragma unroll
                    for (int d = 0; d < 8; d++) xs[d] = U[uswz(rr[d], w, mw)];
                    unsigned long long acc = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                    for (int d = 8; d < cd; d++) acc ^= U[uswz(cr[d], w, mw)];
                    Rn[t * mw + w] = acc;
                }
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                if (kill_due) {
                    d_kills++;
                    kill_pass(kill_from ? blk[3] : 0);
                    kill_due = false; kill_from = 1;
                    c_kill += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                }
                __syncthreads();
                // ---- wave 0: bring the new columns up to date with the pending block, then resolve their pivots; everybody: the pending
                // block's operations on the rows of U (wave 0 after its chain) ----
                if (tid < 64) {
                    int nops_new = 0, anydep_new = 0;
                    __builtin_amdgcn_s_setprio(3);                          // the chain is the critical path: ahead of the row waves sharing this SIMD
                    if (nb > 0) {
                        const long long tq0 = OSD_CLOCK();
                        if (pend > 0) {
                            const int *pp_p = ops + pbuf * 2 * kPipeBlock, *pt_p = pp_p + kPipeBlock;
                            int ppv = pp_p[lane & 15], ptv = pt_p[lane & 15];
                            asm volatile("" : "+v"(ppv), "+v"(ptv));
                            const int tcol = (lane < nb) ? lane : 0;
                            osd_rows_apply(Rn + (size_t)tcol * mw, 0, lane < nb, row, pend, mw, ppv, ptv, Rb + (size_t)pbuf * kPipeBlock * mw, lane, d_wops, d_lops);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        c_po += OSD_CLOCK() - tq0;
                        const long long tq1 = OSD_CLOCK();
                        const int g = lane & 3, w = lane >> 2, lrow0 = row + pend, wq = lrow0 >> 6;
                        QuadPivot S;
#pragma unroll
                        for (int i = 0; i < 4; i++) S.X[i] = (4 * i + g < nb && w < mw) ? Rn[(4 * i + g) * mw + w] : 0ull;
                        S.live = (w > wq) ? ~0ull : ((w == wq) ? (~0ull << (lrow0 & 63)) : 0ull);
                        S.lrow = lrow0; S.nops = 0; S.depmask = 0u; S.oppv = 0; S.optv = 0; S.stop = (lrow0 >= P.rankH || lrow0 >= m); S.nzw = 0u;
#define QLDPC_QSTEP(TT) if (TT < nb && !S.stop) quad_pivot_step<TT>(S, Rn, mw, lane, P.rankH, m);
                        QLDPC_QSTEP(0) QLDPC_QSTEP(1) QLDPC_QSTEP(2) QLDPC_QSTEP(3) QLDPC_QSTEP(4) QLDPC_QSTEP(5) QLDPC_QSTEP(6) QLDPC_QSTEP(7)
                        QLDPC_QSTEP(8) QLDPC_QSTEP(9) QLDPC_QSTEP(10) QLDPC_QSTEP(11) QLDPC_QSTEP(12) QLDPC_QSTEP(13) QLDPC_QSTEP(14) QLDPC_QSTEP(15)
#undef QLDPC_QSTEP
                        if (lane < S.nops) { pp_n[lane] = S.oppv; pt_n[lane] = S.optv; pvcol[lrow0 + lane] = sidx[bcol[S.optv]]; }
                        if (lane < nb && ((S.depmask >> lane) & 1u)) alive[bcol[lane]] = 0;
                        nops_new = S.nops; anydep_new = (S.depmask != 0u) ? 1 : 0;
                        c_chain += OSD_CLOCK() - tq1;
                    }
                    if (lane == 0) { blk[1] = nops_new; blk[2] = anydep_new; }
                    __builtin_amdgcn_s_setprio(0);
                }
                { const long long tq2 = OSD_CLOCK(); if (pend > 0) apply_pending(true); c_own += OSD_CLOCK() - tq2; }
                __syncthreads();
                row += pend;
                pend = blk[1]; pbuf = cbuf;
                c_p2 += OSD_CLOCK() - tp;
                if (row + pend >= P.rankH || row + pend >= m) { finished = true; break; }
                if (blk[2] && !P.nokill) kill_due = true;
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        // ---- drain: the last block's operations ----
        if (pend > 0) {
            const long long tp = OSD_CLOCK();
            apply_pending(false);
            row += pend; pend = 0;
            __syncthreads();
            c_p2 += OSD_CLOCK() - tp;
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2);
            atomicAdd(&P.dbg[12], c_kill); atomicAdd(&P.dbg[11], c_po); atomicAdd(&P.dbg[13], c_chain); atomicAdd(&P.dbg[7], c_own);
        }
#ifdef QLDPC_OSD_TIMERS
        if (P.dbg && (tid & 63) == 0 && tid > 0) atomicAdd(&P.dbg[14], c_own / ((blockDim.x >> 6) - 1));
#endif
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < row; t += T) {
            const int j = pvcol[t];
            const int8_t bbit = (int8_t)((U[uswz(brow, t >> 6, mw)] >> (t & 63)) & 1ull);
            sol[j] = (int8_t)((hard[j] ^ bbit) & 1);
        }
        __syncthreads();
    }
    clk_end(P.clk, clk0);
}

int host_gf2_rank(const qldpc_graph *g);
int ensure_col_rows(const qldpc_graph *g);

static bool plan_osd_pipe(const qldpc_graph *g, OsdPipeArgs &P, size_t &lds) {
    if (g->m > 1024 || g->m < 1 || g->n >= 65535) return false;
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;
    size_t off = std::max((size_t)(g->m + 2) * P.mw * 8, (size_t)g->n * 12 + 16 + sort_cnt);
    off = (size_t)round_up((int64_t)off, 16);
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    P.offAlive = (int)off; off += (size_t)P.K;
    P.offRows = (int)off; off += (size_t)round_up((int64_t)P.K * P.cdeg * 2, 16);
    P.offPc = (int)off; off += (size_t)round_up((int64_t)g->m * 2, 16);
    P.offR = (int)off; off += (size_t)2 * kPipeBlock * P.mw * 8;
    P.offBlk = (int)off; off += (8 + kPipeBlock + 4 * kPipeBlock) * 4;
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
    const int block = (int)std::min<int64_t>(1024, round_up(std::max(g->m + 2, 256), 64));
    hipLaunchKernelGGL(osd0_pipe_kernel, dim3(grid), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
