// OSD-0 with free pivot rows (osd_gj.hip has the algorithm), rows of 16 words (897 <= m <= 1024: the circuit-level matrices): the LOOK-AHEAD QUEUE form.
//
// The pipelined kernel of osd_gj.hip still has, between two pivot chains, the reduction of the next block's columns through U, and it re-tests the
// columns that wait in the chunk for dependence again and again (each test = the column's reduced form from scratch: 6 rows x the live words).
// Here a column is reduced through U exactly ONCE, when it enters a queue of up to 64 reduced columns; from then on its reduced form is kept current
// by the same order-free rule that updates the rows (r ^= XOR_{k : r[pp_k]} C_k for the operations C_k of a finished block).  Consequences:
//   * a block's columns are simply the first 16 of the queue: nothing to compute between two chains but the 16-bit selectors of the pending block;
//   * the dependent-column test is "reduced form has no unused row", evaluated once at entry (most of the ~5,400 dependent columns of a shot never
//     enter) -- no separate test passes, no per-chunk sweep;
//   * filling the queue (reduce the next columns of the chunk, keep the independent ones in order) runs on the fifteen waves beside the pivot
//     chain, behind the previous block's row updates.
// Levels: at the start of iteration c the chains 0 .. c-1 are done, U has seen the blocks 0 .. c-2 ("level c-1", the block c-1 is pending), and so
// has every queue entry.  The selectors lift the entries to level c; the chain of block c starts; the others apply block c-1 to U and then add
// new entries at level c.
#include "osd_gj.h"

#include <algorithm>

namespace qldpc {

constexpr int kNQ = 64;             // queue capacity (reduced columns)
constexpr int kGjqBatches = 3;      // batches of new columns the fifteen waves try per block (a batch = up to 60 columns)

struct OsdGjqArgs {
    OsdGjArgs A;
    int offQ, offC, offInts;
};

__global__ __launch_bounds__(1024) void osd0_gjq_kernel(OsdGjqArgs PP) {
    extern __shared__ unsigned char lds[];
    const OsdGjArgs &P = PP.A;
    const int m = P.m, n = P.n, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = 1024, lane = tid & 63;
    constexpr int mw = 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] column of pivot t
    uint16_t *pvrow = reinterpret_cast<uint16_t *>(lds + P.offPr);         // [m] row of pivot t
    unsigned long long *Qr = reinterpret_cast<unsigned long long *>(lds + PP.offQ);      // [kNQ][16] reduced columns, a ring
    unsigned long long *Cb = reinterpret_cast<unsigned long long *>(lds + PP.offC);      // [2][16][16] composite masks, by block parity
    unsigned long long *usedw = reinterpret_cast<unsigned long long *>(lds + P.offUsed); // [2][16] rows that have pivoted, by block parity
    int *ints = reinterpret_cast<int *>(lds + PP.offInts);
    int *oppb = ints;                                                      // [2][16] pivot rows, by block parity
    uint32_t *selq = reinterpret_cast<uint32_t *>(ints + 32);              // [kNQ] pending operations an entry still needs
    int *qcol = ints + 32 + kNQ;                                           // [kNQ] column id of an entry
    int *wcnt = qcol + kNQ;                                                // [16] independent columns a wave found in the batch in flight
    int *st = wcnt + 16;                                                   // [0] item, [1] pivot mask, [2] queue length, [3] cursor, [4] partial-barrier count
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;
    auto uix = [&](int q, int w) -> int { return q * 16 + (w ^ ((q >> 3) & 14)); };      // (uswz, osd_common.h)

    const int total = *P.count;
    const ClkStamp clk0 = clk_begin(P.clk);
    for (;;) {
        if (tid == 0) st[0] = atomicAdd(P.queue, 1);
        __syncthreads();
        const int item = st[0];
        if (item >= total) break;
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        const long long t_start = OSD_CLOCK();
        if (!P.ordering) {                                                   // column order: ascending |llr| (osd.py:11-12), ties by index
            unsigned long long *keys = reinterpret_cast<unsigned long long *>(lds);
            uint16_t *pa = reinterpret_cast<uint16_t *>(lds + (size_t)n * 8), *pb = pa + n;
            unsigned *cnt = reinterpret_cast<unsigned *>(lds + (((size_t)n * 12 + 15) & ~(size_t)15));
            osd_radix_sort(llr, n, keys, pa, pb, cnt, ordw);
        }
        // ---- init: T = I, b = s + H hard (osd.py:8-9) ----
        for (int t = tid; t < (m + 2) * mw; t += T) U[t] = 0ull;
        if (tid < 32) {                                                      // rows >= m of the last word never pivot
            const int w = tid & 15;
            usedw[tid] = (w == mw - 1 && (m & 63)) ? (~0ull << (m & 63)) : 0ull;
        }
        if (tid < kNQ) selq[tid] = 0u;
        if (tid == 0) st[4] = 0;
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
                    for (int j2 = 0; j2 < 8; j2++) sy ^= (cj[j2] >= 0) ? (hard[cj[j2]] & 1) : 0;
                }
            } else {
                for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            }
            if (sy) atomicOr(&U[uix(brow, r >> 6)], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0, cb = 0, ub = 0;                                         // pivots so far; buffer of the next block's masks / of the used rows in force
        uint32_t pend = 0u;                                                  // pivot mask of the block whose operations U has not seen yet
        int qhead = 0, qn = 0;                                               // the queue: ring start, length
        int cbase = 0, L = 0, cpos = 0;                                      // the chunk: first column (position in the order), length, cursor
        int sbar_target = 0;                                                 // (the partial barrier counts up for the whole shot)
        unsigned long long d_cols = 0, d_chunks = 0, d_batches = 0, d_blocks = 0, c_sel = 0, c_chain = 0, c_pro = 0, c_own = 0, c_gat = 0, c_p3own = 0, c_last = 0;
        (void)c_own; (void)c_gat; (void)c_p3own;
        const long long t_sorted = OSD_CLOCK();
        bool finished = (P.rankH == 0);

        // One batch of the chunk's next columns reduced through U and appended, in order, to the queue if they have an unused row.  Called by `np`
        // threads (a multiple of 64: whole waves) with indices i = 0 .. np - 1; sync() separates the steps for exactly those threads.  16 threads per
        // column.  `used`: the set of used rows that matches the level of U.  Every caller tracks qn / cpos itself (uniform values).
        auto ingest_batch = [&](int i, int np, const unsigned long long *used, auto sync) {
            const int room = kNQ - qn;
            const int nbat = min(min(room, np >> 4), L - cpos);
            const int t = i >> 4, w = i & 15, wi = i >> 6;
            const bool act = t < nbat;
            unsigned long long r = 0ull;
            int colid = 0;
            if (act) {
                const int pos = cpos + t;
                const uint16_t *cr = colrows + pos * cd;
                colid = sidx[pos];
                int rr[8];
                unsigned long long xs[8];
#pragma unroll
                for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;
#pragma unroll
                for (int d = 0; d < 8; d++) xs[d] = U[uix(rr[d], w)];
                r = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                for (int d = 8; d < cd; d++) r ^= U[uix(cr[d], w)];
            }
            const unsigned long long bal = __ballot(act && (r & ~used[w]) != 0ull);
            const uint32_t g16 = (uint32_t)(bal >> (lane & 48)) & 0xFFFFu;   // my column's 16 lanes
            const bool indep = g16 != 0u;
            // the wave's (up to four) columns: bit j = column j of the wave is independent
            const uint32_t im = ((bal & 0xFFFFull) ? 1u : 0u) | ((bal & 0xFFFF0000ull) ? 2u : 0u) | ((bal & 0xFFFF00000000ull) ? 4u : 0u) | ((bal >> 48) ? 8u : 0u);
            if (lane == 0) wcnt[wi] = __builtin_popcount(im);
            sync();
            int before = 0, tot = 0;
            const int nwv = np >> 6;
            for (int w2 = 0; w2 < nwv; w2++) { const int c2 = wcnt[w2]; tot += c2; if (w2 < wi) before += c2; }
            if (indep) {
                const int slot = (qhead + qn + before + __builtin_popcount(im & ((1u << (lane >> 4)) - 1u))) & (kNQ - 1);
                Qr[slot * 16 + w] = r;
                if (w == 0) qcol[slot] = colid;
            }
            qn += tot; cpos += nbat;
            sync();                                                          // (wcnt is reused by the next batch)
        };
        auto sync_all = [&]() { __syncthreads(); };
        auto sync_others = [&]() {                                           // the fifteen waves beside the chain: an LDS counter
            if (lane == 0) atomicAdd(&st[4], 1);
            sbar_target += 15;
            for (int spin = 0; *reinterpret_cast<volatile int *>(&st[4]) < sbar_target && spin < (1 << 22); spin++) __builtin_amdgcn_s_sleep(1);
        };

        while (!finished) {
            // ---- next chunk of the column order when the cursor has run out ----
            if (cpos >= L && cbase + L < n) {
                {
                    cbase += L; L = min(K, n - cbase); cpos = 0; d_chunks++;
                    for (int c = tid; c < L; c += T) sidx[c] = P.ordering ? (uint16_t)P.ordering[shot * n + cbase + c] : ordw[cbase + c];
                    __syncthreads();
                    for (int t = tid; t < L * cd; t += T) {                  // supports of the chunk's columns -> LDS
                        const int c = t / cd, d = t - c * cd, j = sidx[c];
                        const int k = P.colptr[j] + d;
                        colrows[t] = (k < P.colptr[j + 1]) ? (uint16_t)P.rowidx[k] : (uint16_t)m;      // row m of U is all zero
                    }
                    __syncthreads();
                }
            }
            // ---- the queue is short and nobody is filling it: all sixteen waves do (start of a shot, chunk boundaries, the dependent-heavy end) ----
            {
                const long long tq = OSD_CLOCK();
                const unsigned long long *usedU = usedw + 16 * (pend ? (ub ^ 1) : ub);      // U lags by the pending block
                while (qn < kGjBlock && cpos < L) { ingest_batch(tid, T, usedU, sync_all); d_batches++; }
                c_pro += OSD_CLOCK() - tq;
            }
            if (qn < kGjBlock && cpos >= L && cbase + L < n) continue;       // still short: the next chunk first
            if (qn == 0) break;                                              // no column left
            const int nb = min(qn, kGjBlock);
            d_blocks++; d_cols += nb;
            long long tp = OSD_CLOCK();
            const int cprev = cb ^ 1;
            int ppvPrev = oppb[16 * cprev + (tid & 15)];                     // pending column k's pivot row sits in lane k of every 16
            asm volatile("" : "+v"(ppvPrev));
            // ---- selectors: which pending operations an entry still needs (bit pp_k of its reduced form; 16 threads per entry) ----
            if (pend) {
                const int t = tid >> 4, w = tid & 15;
                if (t < qn) {
                    const int slot = (qhead + t) & (kNQ - 1);
                    const unsigned long long r = Qr[slot * 16 + w];
                    uint32_t mysel = 0u;
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int pk = __builtin_amdgcn_readlane(ppvPrev, k);
                        mysel |= ((pk >> 6) == w && ((r >> (pk & 63)) & 1ull)) ? (1u << k) : 0u;
                    }
                    mysel &= pend;
                    if (mysel) atomicOr(&selq[slot], mysel);
                }
                __syncthreads();
            }
            c_sel += OSD_CLOCK() - tp; tp = OSD_CLOCK();
            // ---- wave 0: the block's pivots and composite masks on registers; the others: the rest of the queue lifted, the pending block's row
            //      updates, then new entries ----
            if (wave == 0) {
                const int g = lane & 3, w = lane >> 2;
                __builtin_amdgcn_s_setprio(3);
                GjBlock S;
                uint32_t sl4[4];
                int slot4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    slot4[i] = (qhead + 4 * i + g) & (kNQ - 1);
                    S.X[i] = (4 * i + g < nb) ? Qr[slot4[i] * 16 + w] : 0ull;
                    sl4[i] = (pend && 4 * i + g < nb) ? selq[slot4[i]] : 0u;
                }
                if (pend) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        uint32_t sl = sl4[i];
                        while (sl != 0u) {
                            const int k = __builtin_ctz(sl);
                            sl &= sl - 1u;
                            S.X[i] ^= Cb[(16 * cprev + k) * 16 + w];
                        }
                    }
                    if (lane < 16) selq[(qhead + lane) & (kNQ - 1)] = 0u;    // (read above)
                }
                S.live = ~usedw[16 * ub + w];
                S.nops = 0; S.maxops = P.rankH - row; S.depmask = 0u; S.pivmask = 0u; S.oppv = 0;
#define QLDPC_GSTEP(TT) if (TT < nb && S.nops < S.maxops) gj_pivot_step<TT>(S, lane);
                QLDPC_GSTEP(0) QLDPC_GSTEP(1) QLDPC_GSTEP(2) QLDPC_GSTEP(3) QLDPC_GSTEP(4) QLDPC_GSTEP(5) QLDPC_GSTEP(6) QLDPC_GSTEP(7)
                QLDPC_GSTEP(8) QLDPC_GSTEP(9) QLDPC_GSTEP(10) QLDPC_GSTEP(11) QLDPC_GSTEP(12) QLDPC_GSTEP(13) QLDPC_GSTEP(14) QLDPC_GSTEP(15)
#undef QLDPC_GSTEP
#pragma unroll
                for (int i = 0; i < 4; i++) if (4 * i + g < nb) Cb[(16 * cb + 4 * i + g) * 16 + w] = S.X[i];
                if (g == 0) usedw[16 * (ub ^ 1) + w] = ~S.live;
                if (lane < 16) oppb[16 * cb + lane] = S.oppv;
                if (lane < 16 && ((S.pivmask >> lane) & 1u)) {
                    const int t = row + __builtin_popcount(S.pivmask & ((1u << lane) - 1u));
                    pvcol[t] = (uint16_t)qcol[(qhead + lane) & (kNQ - 1)]; pvrow[t] = (uint16_t)S.oppv;
                }
                if (lane == 0) st[1] = (int)S.pivmask;
                __builtin_amdgcn_s_setprio(0);
#ifdef QLDPC_OSD_TIMERS
                c_own += OSD_CLOCK() - tp;
#endif
            } else {
                const int i = tid - 64;                                      // 0 .. 959
                if (pend) {
                    // the entries behind the block: lifted to the level of the chain that is running (16 threads per entry)
                    for (int t = kGjBlock + (i >> 4); t < qn; t += 60) {
                        const int slot = (qhead + t) & (kNQ - 1), w = i & 15;
                        uint32_t sl = selq[slot];                            // (the entry's sixteen threads sit in one wave: all have read before lane w = 0 clears)
                        if (sl) {
                            unsigned long long r = Qr[slot * 16 + w];
                            if (w == 0) selq[slot] = 0u;
                            while (sl != 0u) { const int k = __builtin_ctz(sl); sl &= sl - 1u; r ^= Cb[(16 * cprev + k) * 16 + w]; }
                            Qr[slot * 16 + w] = r;
                        }
                    }
                    // the pending block's row updates: rows go to threads as q = 16 * lane + (v + lane) % 16, v = 0 .. 15 (conflict-free 16-byte row
                    // accesses under the pair swizzle); waves 1 .. 15 stand for v = 0 .. 14, wave 1 then takes v = 15
                    const unsigned long long *Cp = Cb + 16 * cprev * 16;
                    for (int v = wave - 1; v < 16; v += 15) {
                        const int q = (lane << 4) + ((v + lane) & 15);
                        const bool act = (q < m + 2) && (q != m);
                        gj_rows_apply<true>(U, Cp, act ? q : m, 16, pend, ppvPrev, lane, c_gat);
                        if (wave != 1) break;
                    }
#ifdef QLDPC_OSD_TIMERS
                    c_p3own += OSD_CLOCK() - tp;
#endif
                    sync_others();                                           // every row update has landed: U is at the level of the running chain
                }
                // new entries at the level U has now (the chain's own starting level)
                for (int b = 0; b < kGjqBatches && qn < kNQ - kGjBlock && cpos < L; b++) { ingest_batch(i, 960, usedw + 16 * ub, sync_others); d_batches++; }
                if (i == 0) { st[2] = qn; st[3] = cpos; }
            }
            __syncthreads();
            if (wave == 0) { qn = st[2]; cpos = st[3]; }                    // (the others tracked both while they filled the queue)
            pend = (uint32_t)st[1];                                          // columns of the block that pivoted: its operations are pending now
            row += __builtin_popcount(pend);
            qhead = (qhead + nb) & (kNQ - 1); qn -= nb;
            cb ^= 1; ub ^= 1;
            c_chain += OSD_CLOCK() - tp;
            if (row >= P.rankH || row >= m) finished = true;
        }
        if (pend) {                                                          // the last block's row updates, all waves
            const long long tp = OSD_CLOCK();
            const int cprev = cb ^ 1;
            int ppv = oppb[16 * cprev + (tid & 15)];
            asm volatile("" : "+v"(ppv));
            const int q = (lane << 4) + (((tid >> 6) + tid) & 15);
            const bool act = (q < m + 2) && (q != m);
            gj_rows_apply<true>(U, Cb + 16 * cprev * 16, act ? q : m, 16, pend, ppv, lane, c_gat);
            pend = 0u;
            __syncthreads();
            c_last += OSD_CLOCK() - tp;
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_batches); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_sel); atomicAdd(&P.dbg[10], c_chain); atomicAdd(&P.dbg[11], c_last);
            atomicAdd(&P.dbg[12], c_pro);
        }
#ifdef QLDPC_OSD_TIMERS
        if (P.dbg && tid == 0) atomicAdd(&P.dbg[13], c_own);
        if (P.dbg && lane == 0) atomicAdd(&P.dbg[15], c_gat);
        if (P.dbg && (tid & 127) == 0) atomicAdd(&P.dbg[24 + (tid >> 7)], c_p3own);
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


// tried first for rows of 16 words; P comes filled by osd0_gj_launch (pointers, sizes, rank); launched = false when the LDS carve does not fit
int osd0_gjq_launch(const qldpc_graph *g, const OsdGjArgs &base, int grid, hipStream_t stream, bool &launched) {
    launched = false;
    if (base.mw != 16) return QLDPC_OK;
    OsdGjqArgs PP;
    PP.A = base;
    OsdGjArgs &P = PP.A;
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;
    for (int K = 1024; K >= 256 && !launched; K >>= 1) {
        P.K = K;
        size_t off = std::max((size_t)(g->m + 2) * 16 * 8, (size_t)g->n * 12 + 16 + sort_cnt);       // U, aliased by the sort scratch
        off = (size_t)round_up((int64_t)off, 16);
        P.offIdx = (int)off; off += (size_t)K * 2;
        P.offRows = (int)off; off += (size_t)round_up((int64_t)K * P.cdeg * 2, 8);
        P.offPc = (int)off; off += round_up((int64_t)g->m * 2, 8);
        P.offPr = (int)off; off += round_up((int64_t)g->m * 2, 8);
        PP.offQ = (int)off; off += (size_t)kNQ * 16 * 8;
        PP.offC = (int)off; off += (size_t)2 * kGjBlock * 16 * 8;
        P.offUsed = (int)off; off += 32 * 8;
        PP.offInts = (int)off; off += (32 + 2 * kNQ + 16 + 8) * 4;
        const size_t lds = off + 16;
        if (lds > 160 * 1024) continue;
        int rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_gjq_kernel), 160 * 1024);
        if (rc != QLDPC_OK) return rc;
        hipLaunchKernelGGL(osd0_gjq_kernel, dim3(grid), dim3(1024), lds, stream, PP);
        QLDPC_HIP_TRY(hipGetLastError());
        launched = true;
    }
    return QLDPC_OK;
}

}  // namespace qldpc
