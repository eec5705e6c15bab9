// Shared device pieces of the free-pivot OSD-0 kernels (osd_gj.hip: the pipelined kernel for any m <= 1024; osd_gjq.hip: its look-ahead-queue form for
// rows of 16 words).  See osd_gj.hip for the algorithm.
#pragma once
#include "common.h"
#include "mc_common.h"
#include "osd_common.h"

namespace qldpc {

struct OsdGjArgs {
    int m, n, mw, rankH, K, cdeg;
    const int32_t *colptr, *rowidx, *indptr, *indices;
    const uint16_t *ell_col; const uint8_t *deg_of_row;   // slot-major row view ([slot][m] columns, [m] degrees) or NULL
    const int32_t *list, *count;
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    uint16_t *ordws;               // [grid][n] sorted column order of the shot in flight (global, L2-resident)
    int *queue;                    // next list entry to process (zeroed before the launch)
    int32_t *redo_list, *redo_count;   // shots whose right-hand side is outside the column space (count zeroed before the launch)
    unsigned long long *clk, *dbg;
    int offIdx, offAlive, offRows, offPc, offPr, offR, offBlk, offUsed, offTl;
    int presort;                   // columns of the sorted head (osd_radix_sort_head; 0 = sort everything up front)
    unsigned long long *sortws;    // [grid][sortws_words] global scratch for the order of the columns behind the head
    size_t sortws_words;
};

int osd_presort_choice();          // option "osd_presort" (options.hip)

int host_gf2_rank(const qldpc_graph *g);

constexpr int kGjBlock = 16;
#ifndef QLDPC_GJ_DENSELANES
#define QLDPC_GJ_DENSELANES 40
#endif
constexpr int kGjDenseLanes = QLDPC_GJ_DENSELANES;    // rows of a wave that must change for the wave to update them in place instead of listing them
#ifndef QLDPC_GJ_KILLWINDOW
#define QLDPC_GJ_KILLWINDOW 240
#endif
#ifndef QLDPC_GJ_KILLEVERY
#define QLDPC_GJ_KILLEVERY 2
#endif
constexpr int kGjKillEvery = QLDPC_GJ_KILLEVERY;        // a block that met dependent columns asks for a test on every that-many-th block (the test sits on the
                                                        // critical path of its block, a dependent column in a chain costs less than a pivot)
constexpr int kGjKillWindow = QLDPC_GJ_KILLWINDOW;      // columns behind a block that its dependent-column test covers

struct GjBlock {
    unsigned long long X[4];       // the lane's word of columns g, 4 + g, 8 + g, 12 + g
    unsigned long long live;       // rows of the lane's word that have not pivoted
    int nops, maxops;
    uint32_t depmask, pivmask;     // columns found dependent / columns that pivoted
    int oppv;                      // lane t: pivot row of column t
};

// one pivot step of the block (column T): any unused row with a one, then that row cleared from every other column of the block
// (the caller's control flow is wave-uniform: every condition below is a scalar branch)
template <int T>
__device__ __forceinline__ void gj_pivot_step(GjBlock &S, int lane) {
    constexpr int IT = T >> 2, GT = T & 3;
    const int g = lane & 3, w = lane >> 2;
    const unsigned long long owners = 0x1111111111111111ull << GT;
    const unsigned long long mword = S.X[IT] & S.live;
    const unsigned long long bal = __ballot(mword != 0ull) & owners;
    if (bal == 0ull) { S.depmask |= 1u << T; return; }                                      // in the span of the pivots so far
    const int src = __builtin_ctzll(bal);
#ifdef QLDPC_GJ_FFBL
    const int pbv = __builtin_ctzll(mword | (1ull << 63));                                  // every lane's own first live one: off the scalar chain
    const int wp = src >> 2, pb = __builtin_amdgcn_readlane(pbv, src), pp = wp * 64 + pb;
#else
    const unsigned long long pword = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), src) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, src);
    const int wp = src >> 2, pb = __builtin_ctzll(pword), pp = wp * 64 + pb;
#endif
    const unsigned long long pl = (w == wp) ? (1ull << pb) : 0ull;
    const unsigned long long rm = S.X[IT] & ~pl;                                            // lanes g == GT: the column without its pivot bit
    const unsigned long long rmq = quad_bcast<GT>(rm);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned long long x = S.X[i];
        const uint32_t np = (uint32_t)(__ballot((x & pl) != 0ull) >> (4 * wp));             // bit g: row pp of column 4 i + g
        const int fp = __builtin_amdgcn_sbfe((int)np, g, 1);                                // 0 / -1
        const unsigned long long add = sext64(fp) & rmq;
        S.X[i] = x ^ ((i == IT && g == GT) ? pl : add);                                     // the pivot column itself only loses its pivot bit
    }
    S.live &= ~pl;
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(S.oppv) : "s"(pp), "n"(T));           // lane t of oppv: pivot row of column t
    S.pivmask |= 1u << T;
    S.nops++;
}

// phase 3 for one row per lane: U[q] ^= XOR_{k : bit pp_k of U[q]} C_k.  All 64 lanes of a wave call this together.
// Rows of 16 words: the lane reads its whole row (eight 16-byte reads, conflict-free under the pair swizzle; sixteen 4-byte gathers of the tested
// dwords, every lane another row, are 8-way bank conflicts each), takes the tested dwords out of the registers with a wave-uniform index, and a
// tested bit becomes a 64-lane mask (one compare writing a scalar pair): "nobody in the wave" is a scalar test, a visited operation runs
// under its mask, and a touched row is written back once.
typedef uint32_t gj_row32 __attribute__((ext_vector_type(32)));
template <bool W16>
__device__ __forceinline__ void gj_rows_apply(unsigned long long *U, const unsigned long long *R, int qq, int mw, uint32_t valid, int ppv, int lane, unsigned long long &c_gather) {
    const long long tg = OSD_CLOCK();
    int pk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pk[k] = __builtin_amdgcn_readlane(ppv, k);
    if (W16) {
        uint4 *Uq = reinterpret_cast<uint4 *>(U + qq * 16);
        const int sz = (qq >> 4) & 7;
        gj_row32 row;
#pragma unroll
        for (int w = 0; w < 8; w++) {                                                       // logical order in the registers
            const uint4 t = Uq[w ^ sz];
            row[4 * w] = t.x; row[4 * w + 1] = t.y; row[4 * w + 2] = t.z; row[4 * w + 3] = t.w;
        }
        unsigned long long mk[16], touched = 0ull;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t dw = row[(pk[k] >> 5) & 31];                                     // (wave-uniform register index)
            mk[k] = ((valid >> k) & 1u) ? __ballot(((dw >> (pk[k] & 31)) & 1u) != 0u) : 0ull;
            touched |= mk[k];
#ifdef QLDPC_OSD_TIMERS
            c_gather += (unsigned long long)__builtin_popcountll(mk[k]);                    // (diagnostic build: the count of touched (row, operation) pairs)
#endif
        }
        (void)tg;
        if (touched == 0ull) return;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (mk[k] == 0ull) continue;                                                    // nobody in the wave: scalar skip
            if ((mk[k] >> lane) & 1ull) {
                const uint4 *mk4 = reinterpret_cast<const uint4 *>(R + k * 16);
                uint4 k4[8];
#pragma unroll
                for (int w = 0; w < 8; w++) k4[w] = mk4[w];
#pragma unroll
                for (int w = 0; w < 8; w++) { row[4 * w] ^= k4[w].x; row[4 * w + 1] ^= k4[w].y; row[4 * w + 2] ^= k4[w].z; row[4 * w + 3] ^= k4[w].w; }
            }
        }
        if ((touched >> lane) & 1ull) {
#pragma unroll
            for (int w = 0; w < 8; w++) Uq[w ^ sz] = make_uint4(row[4 * w], row[4 * w + 1], row[4 * w + 2], row[4 * w + 3]);
        }
    } else {
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(U + qq * mw);
        uint32_t Pw[16];
#pragma unroll
        for (int k = 0; k < 16; k++) Pw[k] = row32[pk[k] >> 5];
        unsigned long long mk[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            mk[k] = ((valid >> k) & 1u) ? __ballot(((Pw[k] >> (pk[k] & 31)) & 1u) != 0u) : 0ull;
#ifdef QLDPC_OSD_TIMERS
            c_gather += (unsigned long long)__builtin_popcountll(mk[k]);
#endif
        }
        (void)tg;
        unsigned long long *rowbase = U + qq * mw;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (mk[k] == 0ull) continue;
            if ((mk[k] >> lane) & 1ull) for (int w = 0; w < mw; w++) rowbase[w] ^= R[k * mw + w];
        }
    }
}

}  // namespace qldpc
