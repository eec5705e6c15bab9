// OSD-0 (a9, reference src/decoding/osd.py:5-29 + gf2_elimination_packed_core, src/decoding/kernels.py:48-96) for SMALL matrices
// (m <= 128 rows, n <= 1024 columns: the code-capacity parity checks): ONE WAVE per shot, the matrix itself in LDS.
//
// The transform kernel of gf2.hip is built for 1008 x 8785 matrices (never materialise the permuted matrix, blocks of 16 pivots, a
// workgroup per shot); on a 72 x 144 matrix its fixed phases and barriers cost 267 k cycles per shot -- and a Monte-Carlo step of the
// early-exit pipeline waits for exactly that latency on the ~80 BP failures of a batch.  Here the reference's algorithm runs literally:
// rows of H (n bits, ORIGINAL column order) + the right-hand side in LDS, columns visited in reliability order, the pivot is the first row
// at or below the current rank with a one (kernels.py:71-75), rows are physically swapped (kernels.py:79-82) and every other row with a one
// is added to (kernels.py:88-92).  A lane owns rows lane and lane + 64; one wave, so no barrier: LDS operations of a wave execute in order.
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"
#include "osd_common.h"

#include <algorithm>

namespace qldpc {

struct OsdSmallArgs {
    int m, n, nw, rankH;
    const int32_t *indptr, *indices;
    const int32_t *list, *count;
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    int *queue;
    unsigned long long *clk;
    // fused judge (a Monte-Carlo piece on a lane of its own, csrc/mc.hip): the wave that solved a record also compares it with the true error
    // (logical failure, engine.py:99-100; syndrome check) and the last workgroup zeroes the piece's counters -- one launch instead of two
    const int8_t *jerr; const uint64_t *jLmask; unsigned long long *jtally; int32_t *jcount;
};

// judge of one solved record by the wave that solved it: dsol [n] (LDS) holds the solution bits
__device__ __forceinline__ void osd_small_judge(const OsdSmallArgs &P, int64_t shot, const uint8_t *dsol, int lane) {
    const int m = P.m, n = P.n;
    const int8_t *err = P.jerr + shot * n, *synd = P.synd + shot * m;
    unsigned long long lm = 0ull;
    for (int j = lane; j < n; j += 64)
        if ((err[j] ^ dsol[j]) & 1) lm ^= P.jLmask[j];
    int bad = 0;
    for (int i = lane; i < m; i += 64) {
        int par = 0;
        for (int e = P.indptr[i]; e < P.indptr[i + 1]; e++) par ^= dsol[P.indices[e]];
        bad |= (par ^ synd[i]) & 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lm ^= __shfl_xor(lm, off, 64); bad |= __shfl_xor(bad, off, 64); }
    if (lane == 0) {
        if (lm) { atomicAdd(&P.jtally[QLDPC_TALLY_Z_ERR], 1ull); atomicAdd(&P.jtally[QLDPC_TALLY_TOTAL_ERR], 1ull); }
        if (bad) atomicAdd(&P.jtally[QLDPC_TALLY_UNSAT_Z], 1ull);
    }
}

// end of a fused launch: the OSD count of the piece, and the LAST workgroup to get here zeroes the piece's counters and the ticket counter (every
// workgroup read `total` before it can be counted)
__device__ __forceinline__ void osd_small_judge_finish(const OsdSmallArgs &P, int total, int lane) {
    if (lane != 0) return;
    if (blockIdx.x == 0 && total) atomicAdd(&P.jtally[QLDPC_TALLY_OSD_Z], (unsigned long long)total);
    __threadfence();
    if (atomicAdd(&P.jcount[3], 1) == (int)gridDim.x - 1) { P.jcount[0] = 0; P.jcount[2] = 0; P.jcount[3] = 0; *P.queue = 0; __threadfence(); }
}

// Work hand-out: workgroup b takes list entry b without asking, further entries come from a ticket counter (tickets start at gridDim.x).  A launch over
// a handful of failures -- what a Monte-Carlo piece of a few thousand shots produces -- then issues no atomic at all: 2048 workgroups drawing a ticket
// each from one address took ~60 us per launch, more than the piece's decode.
__global__ __launch_bounds__(64) void osd0_small_kernel(OsdSmallArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, nw = P.nw, lane = threadIdx.x, rs = nw + 1;              // row stride in words: nw matrix words + the rhs word
    unsigned long long *A = reinterpret_cast<unsigned long long *>(lds);                   // [m][rs]
    unsigned long long *keys = A + (size_t)m * rs;                                          // [n]
    uint16_t *ord = reinterpret_cast<uint16_t *>(keys + n);                                 // [n] columns in reliability order
    uint16_t *pivcol = ord + n;                                                             // [m] column of the pivot at position t
    uint8_t *dsol = reinterpret_cast<uint8_t *>(pivcol + m);                                // [n] solution bits (fused judge only)
    const ClkStamp clk0 = clk_begin(P.clk);
    const int total = *P.count;
    for (int item = blockIdx.x; item < total;) {
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        // ---- column order: ascending |llr| (osd.py:11-12), ties by ascending index ----
        if (P.ordering) {
            for (int c = lane; c < n; c += 64) ord[c] = (uint16_t)P.ordering[shot * n + c];
        } else {
            for (int j = lane; j < n; j += 64) keys[j] = osd_key(llr[j]);
            for (int i = lane; i < n; i += 64) {
                const unsigned long long ki = keys[i];
                int rank = 0;
                for (int j = 0; j < n; j++) { const unsigned long long kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
                ord[rank] = (uint16_t)i;
            }
        }
        // ---- rows of H and b = s + H hard (osd.py:8-9) ----
        for (int r = lane; r < m; r += 64) {
            unsigned long long *row = A + (size_t)r * rs;
            for (int w = 0; w < rs; w++) row[w] = 0ull;
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) {
                const int j = P.indices[e];
                row[j >> 6] |= 1ull << (j & 63);
                sy ^= hard[j] & 1;
            }
            row[nw] = (unsigned long long)sy;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- Gauss-Jordan over the columns in reliability order ----
        int rank = 0;
        const int r0 = lane, r1 = lane + 64;
        // (the sweep also ends when no row at or below the diagonal holds a one of the reduced rhs: see the register kernel below)
        auto residual_gone = [&]() -> bool {
            return __ballot((r0 < m && r0 >= rank && (A[(size_t)r0 * rs + nw] & 1ull)) || (r1 < m && r1 >= rank && (A[(size_t)r1 * rs + nw] & 1ull))) == 0ull;
        };
        bool gone = residual_gone();
        for (int c = 0; c < n && rank < P.rankH && rank < m && !gone; c++) {
            const int j = ord[c], w = j >> 6;
            const unsigned long long bit = 1ull << (j & 63);
            const bool c0 = (r0 < m) && (r0 >= rank) && (A[(size_t)r0 * rs + w] & bit) != 0ull;
            const bool c1 = (r1 < m) && (r1 >= rank) && (A[(size_t)r1 * rs + w] & bit) != 0ull;
            const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1);
            if ((b0 | b1) == 0ull) continue;                                                // dependent on the pivots so far
            const int p = b0 ? __builtin_ctzll(b0) : 64 + __builtin_ctzll(b1);              // first candidate in physical order (kernels.py:71-75)
            if (p != rank && lane < rs) {                                                   // swap rows p <-> rank (kernels.py:79-82), one word per lane
                const unsigned long long xp = A[(size_t)p * rs + lane], xr = A[(size_t)rank * rs + lane];
                A[(size_t)p * rs + lane] = xr; A[(size_t)rank * rs + lane] = xp;
            }
            __builtin_amdgcn_wave_barrier();                                                // (no code motion across: other lanes wrote the rows read next)
            // every other row with a one in this column gets the pivot row added (kernels.py:88-92)
            const bool e0 = (r0 < m) && (r0 != rank) && (A[(size_t)r0 * rs + w] & bit) != 0ull;
            const bool e1 = (r1 < m) && (r1 != rank) && (A[(size_t)r1 * rs + w] & bit) != 0ull;
            if (__any(e0 || e1)) {
                for (int k = 0; k < rs; k++) {
                    const unsigned long long pr = A[(size_t)rank * rs + k];
                    if (pr == 0ull) continue;                                               // (uniform: the pivot row's word)
                    if (e0) A[(size_t)r0 * rs + k] ^= pr;
                    if (e1) A[(size_t)r1 * rs + k] ^= pr;
                }
            }
            if (lane == 0) pivcol[rank] = (uint16_t)j;
            rank++;
            __builtin_amdgcn_wave_barrier();
            gone = residual_gone();
        }
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        if (P.jerr) for (int j = lane; j < n; j += 64) dsol[j] = (uint8_t)(hard[j] & 1);
        if (sol != hard) for (int j = lane; j < n; j += 64) sol[j] = hard[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int t = lane; t < rank; t += 64) {
            const int j = pivcol[t];
            const int8_t v = (int8_t)((hard[j] ^ (int8_t)(A[(size_t)t * rs + nw] & 1ull)) & 1);
            sol[j] = v;
            if (P.jerr) dsol[j] = (uint8_t)v;
        }
        if (P.jerr) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); osd_small_judge(P, shot, dsol, lane); }
        {   // next entry: a ticket (the first gridDim.x entries were handed out by workgroup index)
            int t = 0;
            if (lane == 0) t = atomicAdd(P.queue, 1);
            item = (int)gridDim.x + __builtin_amdgcn_readfirstlane(t);
        }
    }
    clk_end(P.clk, clk0);
    if (P.jerr) osd_small_judge_finish(P, total, lane);
}

// The same with the matrix in REGISTERS (n <= 64 NW columns, NW <= 4): lane r holds rows r and r + 64 (NW matrix words + the rhs each).  The LDS
// form above spends ~2 k cycles per pivot in dependent LDS round trips of a lone wave; here a pivot step is ballots, a DPP minimum, lane
// reads and ~60 VALU instructions.  Rows are not moved: the reference's physical order after its swaps (which decides the pivot row, and
// with it the answer for a syndrome outside the column space) is a position per row, and "swap" exchanges two positions.
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xF, 0xF, false));         // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xF, 0xF, false));         // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xF, 0xF, false));        // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xF, 0xF, false));        // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long x, int l) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
}

template <int NW>
__global__ __launch_bounds__(64) void osd0_small_reg_kernel(OsdSmallArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, lane = threadIdx.x;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(lds);                 // [n]
    uint16_t *ord = reinterpret_cast<uint16_t *>(keys + n);                                 // [n] columns in reliability order
    uint16_t *pivcol = ord + n;                                                             // [m] column of the pivot at position t
    uint8_t *dsol = reinterpret_cast<uint8_t *>(pivcol + m);                                // [n] solution bits (fused judge only)
    const ClkStamp clk0 = clk_begin(P.clk);
    const int total = *P.count;
    const int r0 = lane, r1 = lane + 64;
    for (int item = blockIdx.x; item < total;) {
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        if (P.ordering) {
            for (int c = lane; c < n; c += 64) ord[c] = (uint16_t)P.ordering[shot * n + c];
        } else {                                                                            // ascending |llr| (osd.py:11-12), ties by ascending index
            for (int j = lane; j < n; j += 64) keys[j] = osd_key(llr[j]);
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < n; i += 64) {
                const unsigned long long ki = keys[i];
                int rank = 0;
                for (int j = 0; j < n; j++) { const unsigned long long kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
                ord[rank] = (uint16_t)i;
            }
        }
        // rows of H and b = s + H hard (osd.py:8-9): a[slot][0 .. NW-1], a[slot][NW] = rhs
        unsigned long long a[2][NW + 1];
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            const int r = sl ? r1 : r0;
#pragma unroll
            for (int w = 0; w <= NW; w++) a[sl][w] = 0ull;
            if (r < m) {
                int sy = synd[r] & 1;
                for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) {
                    const int j = P.indices[e];
                    const unsigned long long bt = 1ull << (j & 63);
#pragma unroll
                    for (int w = 0; w < NW; w++) a[sl][w] |= ((j >> 6) == w) ? bt : 0ull;
                    sy ^= hard[j] & 1;
                }
                a[sl][NW] = (unsigned long long)sy;
            }
        }
        int pos0 = (r0 < m) ? r0 : 0x7fffffff, pos1 = (r1 < m) ? r1 : 0x7fffffff;            // current physical position of the lane's rows
        __builtin_amdgcn_wave_barrier();
        int rank = 0;
        // the sweep also ends when the residual syndrome is gone: no one of the reduced rhs in a row that has not pivoted (position >= rank).  A later pivot row
        // then holds rhs = 0 -- its addition changes no rhs bit, the swap exchanges two zeros, its column gets e = 0: the answer is final (osd_gj.hip has the argument)
        auto residual_gone = [&]() -> bool { return __ballot((pos0 >= rank && (a[0][NW] & 1ull)) || (pos1 >= rank && (a[1][NW] & 1ull))) == 0ull; };
        bool gone = residual_gone();
        for (int c = 0; c < n && rank < P.rankH && rank < m && !gone; c++) {
            const int j = ord[c], w = j >> 6;
            const unsigned long long bit = 1ull << (j & 63);
            unsigned long long x0 = a[0][0], x1 = a[1][0];
#pragma unroll
            for (int k = 1; k < NW; k++) { x0 = (w == k) ? a[0][k] : x0; x1 = (w == k) ? a[1][k] : x1; }
            const bool h0 = (x0 & bit) != 0ull, h1 = (x1 & bit) != 0ull;                     // the column's ones
            const int cand = min((h0 && pos0 >= rank) ? pos0 : 0x7fffffff, (h1 && pos1 >= rank) ? pos1 : 0x7fffffff);
            const int pp = wave_min_i32(cand);                                              // first candidate in physical order (kernels.py:71-75)
            if (pp == 0x7fffffff) continue;                                                 // dependent on the pivots so far
            const unsigned long long bp0 = __ballot(pos0 == pp), bp1 = __ballot(pos1 == pp);
            const int sl = bp0 ? 0 : 1, pl = __builtin_ctzll(bp0 ? bp0 : bp1);               // the pivot row: slot, lane (uniform)
            // kernels.py:79-82: the rows at positions pp and rank swap
            pos0 = (pos0 == rank) ? pp : ((pos0 == pp) ? rank : pos0);
            pos1 = (pos1 == rank) ? pp : ((pos1 == pp) ? rank : pos1);
            // kernels.py:88-92: every other row with a one in the column gets the pivot row added
            const bool e0 = h0 && !(sl == 0 && lane == pl), e1 = h1 && !(sl == 1 && lane == pl);
#pragma unroll
            for (int k = 0; k <= NW; k++) {
                const unsigned long long pr = sl ? readlane64(a[1][k], pl) : readlane64(a[0][k], pl);
                a[0][k] ^= e0 ? pr : 0ull;
                a[1][k] ^= e1 ? pr : 0ull;
            }
            if (lane == 0) pivcol[rank] = (uint16_t)j;
            rank++;
            gone = residual_gone();
        }
        __builtin_amdgcn_wave_barrier();
        // back-fill (osd.py:19-25): e[pivot col] = reduced rhs of the row at the pivot's position; solution = (hard + e) % 2
        if (P.jerr) for (int jj = lane; jj < n; jj += 64) dsol[jj] = (uint8_t)(hard[jj] & 1);
        if (sol != hard) for (int jj = lane; jj < n; jj += 64) sol[jj] = hard[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (pos0 < rank) { const int jj = pivcol[pos0]; const int8_t v = (int8_t)((hard[jj] ^ (int8_t)(a[0][NW] & 1ull)) & 1); sol[jj] = v; if (P.jerr) dsol[jj] = (uint8_t)v; }
        if (pos1 < rank) { const int jj = pivcol[pos1]; const int8_t v = (int8_t)((hard[jj] ^ (int8_t)(a[1][NW] & 1ull)) & 1); sol[jj] = v; if (P.jerr) dsol[jj] = (uint8_t)v; }
        __builtin_amdgcn_wave_barrier();
        if (P.jerr) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); osd_small_judge(P, shot, dsol, lane); }
        {   // next entry: a ticket (the first gridDim.x entries were handed out by workgroup index)
            int t = 0;
            if (lane == 0) t = atomicAdd(P.queue, 1);
            item = (int)gridDim.x + __builtin_amdgcn_readfirstlane(t);
        }
    }
    clk_end(P.clk, clk0);
    if (P.jerr) osd_small_judge_finish(P, total, lane);
}

int host_gf2_rank(const qldpc_graph *g);

// the ticket counter of the one-wave kernels: a buffer of its own on the handle, zeroed when it is created
int osd_small_queue(const qldpc_graph *g, int **queue) {
    const bool fresh = g->ws_squeue.p == nullptr;
    int rc = g->ws_squeue.ensure(16);
    if (rc != QLDPC_OK) return rc;
    if (fresh) QLDPC_HIP_TRY(zero_now(g->ws_squeue.p, 16));
    *queue = g->ws_squeue.as<int>();
    return QLDPC_OK;
}

// handled = true when the matrix is small enough for this kernel (callers hold g->mu)
int osd0_small_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                      const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled, OsdJudge *judge) {
    handled = false;
    if (g->m > 128 || g->n > 1024 || g->m < 1 || g->n < 1) return QLDPC_OK;
    OsdSmallArgs P;
    P.m = g->m; P.n = g->n; P.nw = (g->n + 63) / 64;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);
    P.rankH = g->gf2_rank;
    P.indptr = g->d_indptr; P.indices = g->d_indices;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = g->clk_probe;
    int rc;
    if ((rc = osd_small_queue(g, &P.queue)) != QLDPC_OK) return rc;
    // the ticket counter must be zero: a caller that zeroes it again after its launches (a Monte-Carlo plan's judge kernel) saves the enqueue here
    if (!(flags & QLDPC_FLAG_INTERNAL_OSD_QUEUE_CLEAN)) QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    P.jerr = nullptr; P.jLmask = nullptr; P.jtally = nullptr; P.jcount = nullptr;
    unsigned grid = 2048;
    if (judge && !d_ordering) {             // the caller's judge rides on this launch (it owns the ticket counter: QLDPC_FLAG_INTERNAL_OSD_QUEUE_CLEAN)
        P.jerr = judge->err; P.jLmask = judge->Lmask; P.jtally = judge->tally; P.jcount = judge->count;
        // every workgroup ends on one atomic (the counter reset belongs to the last one): a grid for the piece, not for the worst case of any piece
        grid = (unsigned)std::min<int64_t>(2048, std::max<int64_t>(32, judge->max_listed / 32));
        judge->fused = true;
    }
    const size_t lds = (size_t)g->m * (P.nw + 1) * 8 + (size_t)g->n * 8 + (size_t)g->n * 2 + (size_t)g->m * 2 + (size_t)g->n + 16;
    const size_t lds_reg = (size_t)g->n * 8 + (size_t)g->n * 2 + (size_t)g->m * 2 + (size_t)g->n + 16;
    switch (P.nw) {                                                  // n <= 256: the rows fit registers
        case 1: hipLaunchKernelGGL(osd0_small_reg_kernel<1>, dim3(grid), dim3(64), lds_reg, stream, P); break;
        case 2: hipLaunchKernelGGL(osd0_small_reg_kernel<2>, dim3(grid), dim3(64), lds_reg, stream, P); break;
        case 3: hipLaunchKernelGGL(osd0_small_reg_kernel<3>, dim3(grid), dim3(64), lds_reg, stream, P); break;
        case 4: hipLaunchKernelGGL(osd0_small_reg_kernel<4>, dim3(grid), dim3(64), lds_reg, stream, P); break;
        default: hipLaunchKernelGGL(osd0_small_kernel, dim3(grid), dim3(64), lds, stream, P);
    }
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
