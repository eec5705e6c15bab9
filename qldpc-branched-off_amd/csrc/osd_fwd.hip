// OSD-0 (a9, reference src/decoding/osd.py:5-29 + gf2_elimination_packed_core, src/decoding/kernels.py:48-96) for matrices with
// m <= 1024 rows, one workgroup per shot, row transform resident in LDS -- the round-2 kernel.
//
// What the reference computes: the columns of H taken in ascending-|llr| order, a full Gauss-Jordan elimination with first-nonzero
// pivoting, and e[pivot column] = reduced rhs at the pivot row.  What is needed of it: WHICH columns pivot (the greedy basis in that
// order, with the reference's row choice) and the solution on them.  This kernel gets both from a FORWARD elimination plus one
// back-substitution, which is the same linear algebra (proof in DESIGN.md 4.2: Jordan's upward eliminations only combine pivot rows,
// so e = ((T H_S)_P)^-1 (T s')_P for the forward transform T as well), and never touches the permuted m x n matrix:
//   * state = the accumulated row transform T (current rows = T * original rows) kept as U = T^T, m x m bits in LDS, in POSITION space:
//     bit p of U[q] = T[p][q], p = current physical row (the reference's swaps kernels.py:79-82 are bit swaps here); row m is all zero
//     (padding target), row m+1 carries the right-hand side b = s + H hard (it transforms like a column);
//   * the next column h (<= 6 ones) in reduced form is XOR_{i in supp h} U[i]; it pivots iff it has a one at a position >= row, and the
//     pivot is the FIRST such position (kernels.py:71-75);
//   * columns are taken 16 at a time.  Phase 1 reduces the 16 through U.  Phase 2 -- the only serial part -- runs in ONE wave on a
//     lane-major copy of the 16 columns (lane l, register w holds the 16 column bits of position 64 w + l): a pivot step is a ballot,
//     two lane reads and three VALU instructions per register, with no barrier and no LDS round trip (the round-1 kernel spent 1.2 k
//     cycles per pivot on a ballot -> scalar -> lane-read -> barrier chain over four waves).  Phase 3 applies the block's <= 16
//     operations to every row of U with the row held in registers and the elimination masks arriving through SCALAR loads (one
//     LDS read + write of the row per block instead of per pivot; the masks cost no LDS or vector-register traffic at all);
//   * forward only: an operation adds the pivot row to the rows BELOW it, so masks are zero up to the pivot position and every
//     register / LDS word left of the block's first pivot is skipped -- half the row on average;
//   * dependent columns are dropped in parallel batches as in the round-1 kernel; the sweep stops at rank(H);
//   * the solution: for the pivots t = r-1 .. 0, e_t = b'[t], and if e_t the column's reduced form above the diagonal is added to b'.
//     Those upper parts are recomputed from the final U (row t of T is frozen once t has pivoted), 64 columns at a time, and
//     consumed by one wave.
// Results are identical to the reference's on every input (also for syndromes outside the column space); the tests compare solutions
// with a literal Gauss-Jordan on the CPU and with the reference's own outputs.
//
// STATUS (round 2, measured on 1 x MI355X, profiles/r02_osd_experiments.txt): correct, but NOT faster than the round-1 Gauss-Jordan
// kernel on the circuit-level matrices (4.3 M vs 3.5 M cycles per shot), so it is selected only by QLDPC_FLAG_OSD_FWD.  Phase 3 here is
// cheaper (1.0 M vs 1.29 M cycles per shot) but phase 2 is not: a lone wave issues one instruction per four cycles and every
// VALU -> SALU -> VALU hop of the pivot chain costs more, 1.4 k cycles per pivot in every formulation tried (all 16 words per step; only
// the two words next to the pivot row with the rest caught up in parallel -- useless here: the reduced columns of these matrices are
// sparse, 60 of 70 blocks per shot have a pivot further away).  The round-1 kernel pays 1.2 k per pivot with four waves and a barrier.
#include "common.h"
#include "mc_common.h"
#include "osd_common.h"

#include <algorithm>

namespace qldpc {

constexpr int kFwdBlock = 16;       // columns per block = bits of a lane-major register
constexpr int kFwdBack = 32;        // pivots per back-substitution batch (two batches in flight)

struct OsdFwdArgs {
    int m, n, mw, rankH, K, cdeg, nokill;
    const int32_t *indptr, *indices, *colptr, *rowidx;
    const uint16_t *colrows_g;         // [n][cdeg] rows of every column in ascending order, padded with the zero row m
    const int32_t *list, *count;       // shots to process (device-resident count)
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    uint16_t *ordws;                   // [grid][n] sorted column order of the shot in flight (global, L2-resident)
    uint32_t *maskg;                   // [grid][kFwdBlock][32] elimination masks of the block in flight (read back through the scalar cache)
    int *queue;                        // next list entry to process (zeroed before the launch)
    unsigned long long *clk, *dbg;
    int offIdx, offAlive, offRows, offPc, offR, offRL, offBlk;
};

#define LDSP __attribute__((address_space(3)))
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// U is stored row-major with an XOR swizzle of the word index when rows are 16 words (128 B): conflict-free row-parallel access
__device__ __forceinline__ int fswz(int q, int w, int mw) { return q * mw + ((mw == 16) ? (w ^ (q & 15)) : w); }

// wave-uniform values that arrive as arguments of a non-inlined function live in vector registers: hand them back to the scalar unit
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class T> __device__ __forceinline__ T *uni_ptr(T *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

// 64 bytes from a wave-uniform global address into scalar registers (the masks of phase 3: no LDS, no VGPRs, free broadcast)
// (load and wait are one statement: between a bare s_load and its s_waitcnt the compiler could copy the not-yet-written registers)
__device__ __forceinline__ u32x16 sload16(const uint32_t *p) {
    u32x16 r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ void sload32(const uint32_t *p, u32x16 &lo, u32x16 &hi) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
}

__device__ __forceinline__ void sprefetch_block(const uint32_t *p) {
    uint32_t d0, d1, d2, d3;
    // the whole 2 KB staging block (32 lines), every miss in flight at once; the loaded dwords are not used
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(p) : "memory");
}

// Phase 3 for one row of U: the block's operations on the row held in registers.  W0 = first word the block can change (even).
// Operation k (kernels.py:79-92 restricted to the rows below the pivot): swap bits a <-> pp, then add the mask if bit a is set.  With
// ba / bp = bits a / pp before the operation:  bit a' = bp;  every other bit:  row' = row ^ (bp ? mask ^ E_pp : 0) ^ (ba ? E_pp : 0).
// Position a_k = row0 + k is final once operation k is done (later operations of a forward elimination only touch positions below
// it), so the new bits a are collected in `abits` and written as one bit-field after the loop; until then bit a_k of the registers is
// dead.  maskg holds A_k = mask ^ E_pp (mask itself when a == pp); lane k of ppv holds pp_k.  The row is a 32-dword vector indexed through s_set_gpr_idx; the
// one dynamic write per operation is unconditional (a uniform branch around it makes the compiler copy all 32 registers).
template <int W0>
__device__ __forceinline__ void fwd_apply_row(LDSP unsigned long long *U, int q, int mw, int row0, int nops, int ppv, const uint32_t *maskg) {
    u32x32 u;
#pragma unroll
    for (int d = 0; d < 32; d++) u[d] = 0u;
#pragma unroll
    for (int w = W0; w < 16; w++) {
        if (w < mw) {
            const unsigned long long x = U[fswz(q, w, mw)];
            u[2 * w] = (uint32_t)x; u[2 * w + 1] = (uint32_t)(x >> 32);
        }
    }
    uint32_t abits = 0u;
    for (int k = 0; k < nops; k++) {
        const int a = row0 + k, pp = __builtin_amdgcn_readlane(ppv, k);
        const uint32_t *mk = maskg + k * 32;
        u32x16 Alo, Ahi;
        if (W0 < 8) sload32(mk, Alo, Ahi); else Ahi = sload16(mk + 16);
        const int da = a >> 5, dp = pp >> 5;
        const uint32_t am = 1u << (a & 31), pm = 1u << (pp & 31);
        const bool ba = (u[da] & am) != 0u, bp = (u[dp] & pm) != 0u;
        abits |= (bp ? 1u : 0u) << k;
        u[dp] ^= (ba && a != pp) ? pm : 0u;          // branch-free: a uniform branch around a dynamic insert makes the compiler copy the row
        if (bp) {
#pragma unroll
            for (int d = 2 * W0; d < 32; d++) u[d] ^= (d < 16) ? Alo[d & 15] : Ahi[d & 15];
        }
    }
    // positions row0 .. row0 + nops - 1 take their final bits; the field lies in words W0 .. W0 + 2
    {
        const unsigned long long field = (nops >= 32) ? 0xFFFFFFFFull : ((1ull << nops) - 1ull);
        const int ws = row0 >> 6, sh = row0 & 63;
        const unsigned long long clr_lo = field << sh, set_lo = (unsigned long long)abits << sh;
        const unsigned long long clr_hi = sh ? (field >> (64 - sh)) : 0ull, set_hi = sh ? ((unsigned long long)abits >> (64 - sh)) : 0ull;
#pragma unroll
        for (int w = W0; w < W0 + 3 && w < 16; w++) {
            if (w == ws) {
                u[2 * w] = (u[2 * w] & ~(uint32_t)clr_lo) | (uint32_t)set_lo;
                u[2 * w + 1] = (u[2 * w + 1] & ~(uint32_t)(clr_lo >> 32)) | (uint32_t)(set_lo >> 32);
            }
            if (w == ws + 1) {
                u[2 * w] = (u[2 * w] & ~(uint32_t)clr_hi) | (uint32_t)set_hi;
                u[2 * w + 1] = (u[2 * w + 1] & ~(uint32_t)(clr_hi >> 32)) | (uint32_t)(set_hi >> 32);
            }
        }
    }
#pragma unroll
    for (int w = W0; w < 16; w++)
        if (w < mw) U[fswz(q, w, mw)] = ((unsigned long long)u[2 * w + 1] << 32) | u[2 * w];
}

// Phase 2: the pivots of one block, executed by ONE wave on the lane-major copy (register i, lane l = the 16 column bits of position
// 64 (WQ0 + i) + l; words left of the block's first pivot are final and never loaded).  Column t pivots at the first position >= lrow
// holding a one (kernels.py:71-75); the rows at positions lrow and pp swap (kernels.py:79-82) in column t and the later columns; the
// pivot row is added to every row BELOW it that has a one in column t (kernels.py:88-92, forward part), later columns only.
// A lone wave issues one instruction per four cycles whatever it is, so the serial chain is kept to the TWO words the block's pivots
// live in (positions lrow .. lrow + 15 and, once the transform has filled in, the pivot a few positions behind lrow): per step two
// ballots, two lane reads, branch-free selects on those two registers.  Every other word only ever receives "row ^= pivot row where
// bit t" -- a per-position linear map that does not feed back into the chain -- and is brought up to date afterwards by the other
// waves in parallel (fwd_far_words below) from the per-step pivot rows left in ophi[].  A column whose pivot lies further away (the
// first blocks of a shot, T still near the identity) switches the rest of the block to the full form: the far words are loaded,
// caught up and carried along here (blk[5] = 1 tells the parallel stage not to transform them again).
// (not inlined: inside the kernel its scalar registers compete with ~70 live kernel-wide values and the step loop fills with spill code)
template <int WQ0>
__device__ __noinline__ void fwd_pivot_block(LDSP uint32_t *RL, int mw, int nb, int row, int rankH, int m, LDSP uint8_t *alive, LDSP const int *bcol,
                                             LDSP const uint16_t *sidx, LDSP int *opa, LDSP int *opp, LDSP int *opt, LDSP int *ophi,
                                             LDSP uint16_t *pvcol, LDSP int *blk, int lane) {
    constexpr int NW = 16 - WQ0, NN = NW < 2 ? NW : 2;
    mw = uni(mw); nb = uni(nb); row = uni(row); rankH = uni(rankH); m = uni(m);
    uint32_t r[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) r[i] = (i < NN && WQ0 + i < mw) ? RL[(WQ0 + i) * 64 + lane] : 0u;
    bool full = false;
    int lrow = row, nops = 0, anydep = 0;
    for (int t = 0; t < nb; t++) {
        const int alq = lrow - 64 * WQ0;                                                         // 0 .. 127: position lrow relative to word WQ0
        unsigned long long c0 = __ballot(((r[0] >> t) & 1u) != 0u), c1 = 0ull;
        if (NN > 1) c1 = __ballot(((r[NN > 1 ? 1 : 0] >> t) & 1u) != 0u);
        if (alq < 64) c0 &= ~0ull << alq; else { c0 = 0ull; c1 &= ~0ull << (alq - 64); }
        int app = -1;                                                                            // pivot position relative to word WQ0
        uint32_t prow = 0u;
        if (c0 != 0ull) { app = __builtin_ctzll(c0); prow = (uint32_t)__builtin_amdgcn_readlane((int)r[0], app); }
        else if (c1 != 0ull) { const int lp = __builtin_ctzll(c1); app = 64 + lp; prow = (uint32_t)__builtin_amdgcn_readlane((int)r[NN > 1 ? 1 : 0], lp); }
        else if (NW > 2) {
            if (!full) {                                                                         // bring the far words up to date, keep them from here on
#pragma unroll
                for (int i = 2; i < NW; i++) r[i] = (WQ0 + i < mw) ? RL[(WQ0 + i) * 64 + lane] : 0u;
                for (int j = 0; j < nops; j++) {
                    const int tj = __builtin_amdgcn_readfirstlane(opt[j]);
                    const uint32_t ph = (uint32_t)__builtin_amdgcn_readfirstlane(ophi[j]);
#pragma unroll
                    for (int i = 2; i < NW; i++) r[i] ^= (uint32_t)__builtin_amdgcn_sbfe((int)r[i], tj, 1) & ph;
                }
                full = true;
            }
#pragma unroll
            for (int i = 2; i < NW; i++) {
                if (app < 0) {
                    const unsigned long long bal = __ballot(((r[i] >> t) & 1u) != 0u);
                    if (bal != 0ull) { const int lp = __builtin_ctzll(bal); app = i * 64 + lp; prow = (uint32_t)__builtin_amdgcn_readlane((int)r[i], lp); }
                }
            }
        }
        if (app < 0) {                                                                           // dependent on the pivots so far
            if (lane == 0) alive[bcol[t]] = 0;
            anydep = 1;
            continue;
        }
        const uint32_t arow = (alq < 64) ? (uint32_t)__builtin_amdgcn_readlane((int)r[0], alq)
                                         : (uint32_t)__builtin_amdgcn_readlane((int)r[NN > 1 ? 1 : 0], alq - 64);
        const uint32_t low = (1u << t) - 1u;                                                     // finished columns keep their bits
        const uint32_t newA = (arow & low) | (prow & ~low), newP = (prow & low) | (arow & ~low);
        const uint32_t phi = prow & ~low & ~(1u << t);                                           // the pivot row on the later columns
#pragma unroll
        for (int i = 0; i < NN; i++) {                                                           // branch-free on relative positions; a word left of lrow
            const int posn = i * 64 + lane;                                                      // has no position == app, == alq or > alq: it stays as it is
            uint32_t x = r[i];
            x = (posn == app) ? newP : x;
            x = (posn == alq) ? newA : x;                                                        // (app == alq: no swap, the row keeps prow on the later columns)
            r[i] = x ^ ((uint32_t)__builtin_amdgcn_sbfe((int)x, t, 1) & ((posn > alq) ? phi : 0u));
        }
        if (NW > 2 && full) {
#pragma unroll
            for (int i = 2; i < NW; i++) {
                uint32_t x = r[i];
                if ((app >> 6) == i) x = (lane == (app & 63)) ? newP : x;
                r[i] = x ^ ((uint32_t)__builtin_amdgcn_sbfe((int)x, t, 1) & phi);
            }
        }
        if (lane == 0) { opa[nops] = lrow; opp[nops] = 64 * WQ0 + app; opt[nops] = t; ophi[nops] = (int)phi; pvcol[lrow] = sidx[bcol[t]]; }
        nops++; lrow++;
        if (lrow >= rankH || lrow >= m) break;                                                   // full rank: the remaining columns cannot pivot
    }
#pragma unroll
    for (int i = 0; i < NW; i++)
        if ((i < NN || full) && WQ0 + i < mw) RL[(WQ0 + i) * 64 + lane] = r[i];
    if (lane == 0) { blk[1] = nops; blk[2] = anydep; blk[5] = full ? 1 : 0; }
}

template <int W0>
__device__ __noinline__ void fwd_apply_block(LDSP unsigned long long *U, int m, int mw, int row0, int nops, int ppv, const uint32_t *maskg) {
    m = uni(m); mw = uni(mw); row0 = uni(row0); nops = uni(nops); maskg = uni_ptr(maskg);
    for (int q = threadIdx.x; q < m + 2; q += blockDim.x) {
        if (q == m) continue;
        fwd_apply_row<W0>(U, q, mw, row0, nops, ppv, maskg);
    }
}

__global__ __launch_bounds__(1024) void osd0_fwd_kernel(OsdFwdArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x;
    const int wv = tid >> 6, lane = tid & 63;
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + P.offAlive);        // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports; reused by the back-substitution
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] pivot t sits at position t
    unsigned long long *R = reinterpret_cast<unsigned long long *>(lds + P.offR);       // [kFwdBlock][mw] reduced columns, column-major
    uint32_t *RL = reinterpret_cast<uint32_t *>(lds + P.offRL);            // [16][64] the same block, lane-major
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [1] nops, [2] anydep, [3] next c, [4] work item, [5] far words carried by phase 2
    int *bcol = blk + 8, *opa = bcol + kFwdBlock, *opp = opa + kFwdBlock, *opt = opp + kFwdBlock, *ophi = opt + kFwdBlock;
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    uint32_t *maskg = P.maskg + (size_t)blockIdx.x * (kFwdBlock * 32);
    const int brow = m + 1;                                                // U row that carries b

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
        // ---- column order: ascending |llr| (osd.py:11-12), ties by ascending index ----
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
            U[fswz(r, r >> 6, mw)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            if (sy) atomicOr(&U[fswz(brow, r >> 6, mw)], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0;
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_p3 = 0, c_kill = 0, c_back = 0, c_ser = 0, d_full = 0;
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
            // drops every still-alive column of the chunk from c0 on that is dependent on the pivots found so far (one thread per column)
            auto kill_pass = [&](int c0) {
                const int wq = row >> 6;
                for (int c2 = c0 + tid; c2 < L; c2 += T) {
                    if (!alive[c2]) continue;
                    const uint16_t *cr2 = colrows + c2 * cd;
                    int rr[8];                                           // the column's support once (short columns point at the zero row m)
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr2[d] : m;
                    unsigned long long any = 0ull;
                    for (int w = wq; w < mw; w++) {
                        unsigned long long xs[8];
#pragma unroll
                        for (int d = 0; d < 8; d++) xs[d] = U[fswz(rr[d], w, mw)];
                        unsigned long long x = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                        for (int d = 8; d < cd; d++) x ^= U[fswz(cr2[d], w, mw)];     // columns heavier than 8 (not the circuit-level matrices)
                        any |= (w == wq) ? (x & (~0ull << (row & 63))) : x;
                    }
                    if (!any) alive[c2] = 0;
                }
            };
            if (row > 0 && !P.nokill) {                                      // a fresh chunk late in the sweep is mostly dependent columns
                const long long tk = OSD_CLOCK();
                d_kills++;
                kill_pass(0);
                __syncthreads();
                c_kill += OSD_CLOCK() - tk;
            }
            // ================= blocks of up to kFwdBlock alive columns =================
            while (true) {
                if (tid < 64) {                                              // wave 0 collects the next alive columns of the chunk (ballot scan)
                    int c = blk[3], nbc = 0;
                    while (c < L && nbc < kFwdBlock) {
                        const int cc = c + tid;
                        const bool al = (cc < L) && alive[cc];
                        const unsigned long long bal = __ballot(al);
                        const int before = __builtin_popcountll(bal & ((1ull << tid) - 1ull));
                        if (al && nbc + before < kFwdBlock) bcol[nbc + before] = cc;
                        const int got = __builtin_popcountll(bal);
                        if (nbc + got >= kFwdBlock) {                         // stop right behind the column that filled the block
                            int need = kFwdBlock - nbc;
                            unsigned long long bb = bal;
                            int lastpos = 0;
                            while (need-- > 0) { lastpos = __builtin_ctzll(bb); bb &= bb - 1; }
                            c += lastpos + 1; nbc = kFwdBlock;
                        } else { nbc += got; c += 64; }
                    }
                    if (c > L) c = L;
                    if (tid == 0) { blk[0] = nbc; blk[1] = 0; blk[2] = 0; blk[3] = c; }
                }
                __syncthreads();
                const int nb = blk[0];
                if (nb == 0) break;
                d_blocks++; d_cols += nb;
                long long tp = OSD_CLOCK();
                // ---- phase 1: reduced columns R[t] = XOR of the U rows of the column's support (column-major words) ----
                for (int x = tid; x < nb * mw; x += T) {
                    const int t = x / mw, w = x - t * mw;
                    const uint16_t *cr = colrows + bcol[t] * cd;
                    int rr[8];
                    unsigned long long xs[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;           // short columns point at the zero row m
#pragma unroll
                    for (int d = 0; d < 8; d++) xs[d] = U[fswz(rr[d], w, mw)];
                    unsigned long long acc = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                    for (int d = 8; d < cd; d++) acc ^= U[fswz(cr[d], w, mw)];
                    R[t * mw + w] = acc;
                }
                __syncthreads();
                // lane-major copy: wave w turns word w of the nb columns into one register per lane (bit t = column t at position 64 w + lane)
                if (wv < mw) {
                    const uint32_t *R32 = reinterpret_cast<const uint32_t *>(R);
                    uint32_t acc = 0u;
                    for (int t = 0; t < nb; t++) acc |= ((R32[(t * mw + wv) * 2 + (lane >> 5)] >> (lane & 31)) & 1u) << t;
                    RL[wv * 64 + lane] = acc;
                }
                __syncthreads();
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 2: the block's pivots, one wave, no barrier (fwd_pivot_block) ----
                if (wv == 0) {
#define QLDPC_PIVOT_CASE(W) case W: fwd_pivot_block<W>((LDSP uint32_t *)RL, mw, nb, row, P.rankH, m, (LDSP uint8_t *)alive, (LDSP const int *)bcol, (LDSP const uint16_t *)sidx, (LDSP int *)opa, (LDSP int *)opp, (LDSP int *)opt, (LDSP int *)ophi, (LDSP uint16_t *)pvcol, (LDSP int *)blk, lane); break;
                    switch (row >> 6) {
                        QLDPC_PIVOT_CASE(0) QLDPC_PIVOT_CASE(1) QLDPC_PIVOT_CASE(2) QLDPC_PIVOT_CASE(3) QLDPC_PIVOT_CASE(4) QLDPC_PIVOT_CASE(5)
                        QLDPC_PIVOT_CASE(6) QLDPC_PIVOT_CASE(7) QLDPC_PIVOT_CASE(8) QLDPC_PIVOT_CASE(9) QLDPC_PIVOT_CASE(10) QLDPC_PIVOT_CASE(11)
                        QLDPC_PIVOT_CASE(12) QLDPC_PIVOT_CASE(13) QLDPC_PIVOT_CASE(14)
                        default: fwd_pivot_block<15>((LDSP uint32_t *)RL, mw, nb, row, P.rankH, m, (LDSP uint8_t *)alive, (LDSP const int *)bcol, (LDSP const uint16_t *)sidx, (LDSP int *)opa, (LDSP int *)opp, (LDSP int *)opt, (LDSP int *)ophi, (LDSP uint16_t *)pvcol, (LDSP int *)blk, lane); break;
                    }
#undef QLDPC_PIVOT_CASE
                }
                __syncthreads();
                const long long tpp = OSD_CLOCK();
                c_ser += tpp - tp;
                const int nops = blk[1], anydep = blk[2];
                d_full += blk[5];
                // ---- all waves: wave w owns word w of the block.  Far words first receive the block's per-position map (see fwd_pivot_block);
                // then every finished column goes back to a column-major word by ballot -- the elimination mask of its operation,
                // A_k = (column t_k at the positions below a_k) ^ E_pp -- collected one operation per lane and stored with one instruction
                // into the global staging block the scalar loads of phase 3 read.
                if (wv < mw && nops > 0) {
                    uint32_t x = RL[wv * 64 + lane];
                    const int v_t = opt[lane & 15], v_a = opa[lane & 15], v_p = opp[lane & 15], v_h = ophi[lane & 15];
                    if (!blk[5] && wv >= (row >> 6) + 2) {
                        for (int k = 0; k < nops; k++)
                            x ^= (uint32_t)__builtin_amdgcn_sbfe((int)x, __builtin_amdgcn_readlane(v_t, k), 1) & (uint32_t)__builtin_amdgcn_readlane(v_h, k);
                    }
                    int mine_lo = 0, mine_hi = 0;
                    for (int k = 0; k < nops; k++) {
                        const int t = __builtin_amdgcn_readlane(v_t, k), a = __builtin_amdgcn_readlane(v_a, k), pp = __builtin_amdgcn_readlane(v_p, k);
                        const int wa = a >> 6;
                        unsigned long long bal = __ballot(((x >> t) & 1u) != 0u);
                        if (wv < wa) bal = 0ull;
                        else if (wv == wa) bal &= ((~0ull << (a & 63)) << 1);
                        if (a != pp && wv == (pp >> 6)) bal ^= 1ull << (pp & 63);
                        if (lane == k) { mine_lo = (int)(uint32_t)bal; mine_hi = (int)(uint32_t)(bal >> 32); }
                    }
                    if (lane < nops) *reinterpret_cast<uint2 *>(maskg + lane * 32 + 2 * wv) = make_uint2((uint32_t)mine_lo, (uint32_t)mine_hi);
                }
                // the stores only have to reach this XCD's L2 (the vector L1 is write-through and the scalar cache reads from the same L2): wait for
                // them, no agent-scope release -- __threadfence() writes the whole L2 back on this part (58 us per block, measured)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                c_p2 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 3: the block's operations on every row of U (and on b), rows in registers, masks in scalar registers ----
                if (nops > 0) {
                    // the staging block is rewritten every block: drop stale scalar-cache lines, then pull the block's 2 x nops lines into the
                    // scalar cache with all misses in flight at once (one L2 round trip per block instead of one per operation)
                    sprefetch_block(maskg);
                    const int w0 = (row >> 6) & ~1;
                    // the block's pivot positions, one per lane (v_readlane per operation instead of an LDS round trip); loaded by EVERY lane:
                    // v_readlane also reads lanes that sit out the row loop below
                    int ppv = opp[tid & 15];
                    asm volatile("" : "+v"(ppv));                          // pins the load here: the compiler may otherwise sink it into the divergent loop
                    LDSP unsigned long long *Ul = (LDSP unsigned long long *)U;
                    switch (w0) {
                        case 0: fwd_apply_block<0>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 2: fwd_apply_block<2>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 4: fwd_apply_block<4>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 6: fwd_apply_block<6>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 8: fwd_apply_block<8>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 10: fwd_apply_block<10>(Ul, m, mw, row, nops, ppv, maskg); break;
                        case 12: fwd_apply_block<12>(Ul, m, mw, row, nops, ppv, maskg); break;
                        default: fwd_apply_block<14>(Ul, m, mw, row, nops, ppv, maskg); break;
                    }
                }
                row += nops;
                __syncthreads();
                c_p3 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                if (row >= P.rankH || row >= m) { finished = true; break; }
                // ---- dependent columns were met: drop every column of the chunk that is dependent by now ----
                if (anydep && !P.nokill) {
                    d_kills++;
                    kill_pass(blk[3]);
                    __syncthreads();
                    c_kill += OSD_CLOCK() - tp;
                }
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        // ---- back-substitution: e_t = b'[t]; if e_t, add the upper part of pivot column t's reduced form to b' (t = row-1 .. 0).
        // The reduced forms come from the FINAL U (row t of T is frozen once t has pivoted), kBack columns at a time: waves 1.. compute
        // batch i + 1 into one buffer while wave 0 consumes batch i from the other (lane w < mw of wave 0 holds word w of b').
        {
            const long long tb = OSD_CLOCK();
            unsigned long long *RB = reinterpret_cast<unsigned long long *>(colrows);             // [2][kFwdBack][mw]
            const int nbatch = (row + kFwdBack - 1) / kFwdBack;
            auto fill = [&](int bi, int first_thread, int nthreads) {                             // batch bi = pivots [lo, hi), hi = row - bi * kFwdBack
                const int hi = row - bi * kFwdBack, lo = max(0, hi - kFwdBack), cnt = hi - lo;
                unsigned long long *dst = RB + (size_t)(bi & 1) * kFwdBack * mw;
                for (int x = tid - first_thread; x < cnt * mw; x += nthreads) {
                    const int tl = x / mw, w = x - tl * mw, t = lo + tl;
                    const uint16_t *cr = P.colrows_g + (size_t)pvcol[t] * cd;
                    unsigned long long acc = 0ull;
                    for (int d = 0; d < cd; d++) acc ^= U[fswz(cr[d], w, mw)];                    // short columns are padded with the zero row m
                    if (w == (t >> 6)) acc &= ~(1ull << (t & 63));                                // the diagonal stays in b' as e_t
                    dst[tl * mw + w] = acc;
                }
            };
            if (nbatch > 0) fill(0, 0, T);
            __syncthreads();
            unsigned long long bw = (wv == 0 && lane < mw) ? U[fswz(brow, lane, mw)] : 0ull;
            for (int bi = 0; bi < nbatch; bi++) {
                if (wv == 0) {
                    const int hi = row - bi * kFwdBack, lo = max(0, hi - kFwdBack), cnt = hi - lo;
                    const unsigned long long *src = RB + (size_t)(bi & 1) * kFwdBack * mw;
                    for (int t1 = cnt; t1 > 0; t1 -= 8) {
                        unsigned long long col[8];
#pragma unroll
                        for (int j2 = 0; j2 < 8; j2++) col[j2] = (lane < mw && t1 - 1 - j2 >= 0) ? src[(t1 - 1 - j2) * mw + lane] : 0ull;
#pragma unroll
                        for (int j2 = 0; j2 < 8; j2++) {
                            const int tl = t1 - 1 - j2;
                            if (tl >= 0) {
                                const int t = lo + tl;
                                const uint32_t half = (t & 32) ? (uint32_t)(bw >> 32) : (uint32_t)bw;
                                const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)half, t >> 6);
                                if ((word >> (t & 31)) & 1u) bw ^= col[j2];
                            }
                        }
                    }
                } else if (bi + 1 < nbatch) {
                    fill(bi + 1, 64, T - 64);
                }
                __syncthreads();
            }
            if (wv == 0 && lane < mw) U[fswz(brow, lane, mw)] = bw;
            __syncthreads();
            c_back += OSD_CLOCK() - tb;
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2); atomicAdd(&P.dbg[11], c_p3);
            atomicAdd(&P.dbg[12], c_kill); atomicAdd(&P.dbg[13], c_back); atomicAdd(&P.dbg[7], c_ser); atomicAdd(&P.dbg[14], d_full);
        }
        // ---- back-fill (osd.py:19-25): e[pivot col] = solution bit; solution = (hard + e) % 2 ----
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < row; t += T) {
            const int j = pvcol[t];
            const int8_t bbit = (int8_t)((U[fswz(brow, t >> 6, mw)] >> (t & 63)) & 1ull);
            sol[j] = (int8_t)((hard[j] ^ bbit) & 1);
        }
        __syncthreads();
    }
    clk_end(P.clk, clk0);
}

int host_gf2_rank(const qldpc_graph *g);
int ensure_col_rows(const qldpc_graph *g);

// false: this kernel does not take the graph (m > 1024 or the state does not fit one CU's LDS) -- the caller falls back
static bool plan_osd_fwd(const qldpc_graph *g, OsdFwdArgs &P, size_t &lds) {
    if (g->m > 1024 || g->n >= 65535 || g->m < 1) return false;
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;                         // [256][waves] radix counters + per-wave sums
    size_t off = std::max((size_t)(g->m + 2) * P.mw * 8, (size_t)g->n * 12 + 16 + sort_cnt);     // U, aliased by the sort scratch
    off = (size_t)round_up((int64_t)off, 16);
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    P.offAlive = (int)off; off += (size_t)P.K;
    P.offRows = (int)off; off += (size_t)round_up(std::max<int64_t>((int64_t)P.K * P.cdeg * 2, (int64_t)2 * kFwdBack * P.mw * 8), 16);
    P.offPc = (int)off; off += (size_t)round_up((int64_t)g->m * 2, 16);
    P.offR = (int)off; off += (size_t)kFwdBlock * P.mw * 8;
    P.offRL = (int)off; off += (size_t)16 * 64 * 4;
    P.offBlk = (int)off; off += (8 + 5 * kFwdBlock) * 4;
    lds = off + 16;
    return lds <= 160 * 1024;
}

int osd0_fwd_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled) {
    OsdFwdArgs P;
    size_t lds = 0;
    handled = false;
    if (!plan_osd_fwd(g, P, lds)) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    const int grid = 512;
    const size_t sz_ord = (size_t)round_up((int64_t)grid * g->n * 2 + 64, 256);
    const size_t sz_mask = (size_t)grid * kFwdBlock * 32 * 4;
    int rc = g->ws_misc.ensure(sz_ord + sz_mask);
    if (rc != QLDPC_OK) return rc;
    P.ordws = g->ws_misc.as<uint16_t>();
    P.maskg = reinterpret_cast<uint32_t *>(g->ws_misc.as<unsigned char>() + sz_ord);
    QLDPC_HIP_TRY(hipMemsetAsync(P.maskg, 0, sz_mask, stream));          // words beyond the row length are never written: they must read as zero
    if ((rc = ensure_col_rows(g)) != QLDPC_OK) return rc;
    P.colrows_g = g->d_col_rows;
    P.indptr = g->d_indptr; P.indices = g->d_indices; P.colptr = g->d_colptr; P.rowidx = g->d_rowidx;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = g->clk_probe;
    P.dbg = osd_timer_buffer();
    P.nokill = (flags & QLDPC_FLAG_OSD_NOKILL) ? 1 : 0;
    const int block = (int)std::min<int64_t>(1024, round_up(std::max(g->m + 2, 256), 64));
    if ((rc = g->ws_queue.ensure(16)) != QLDPC_OK) return rc;
    P.queue = g->ws_queue.as<int>() + 2;
    QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_fwd_kernel), 160 * 1024)) != QLDPC_OK) return rc;
    hipLaunchKernelGGL(osd0_fwd_kernel, dim3(grid), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
