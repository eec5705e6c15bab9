// Workgroup-per-shot min-sum decoder for large Tanner graphs, second form: EVERYTHING an iteration touches is in LDS or in registers.
//
// minsum_wg.hip keeps the posteriors V[n] and a 24-byte state per check in LDS and reads its index tables (column-side edge list, column
// degree, prior, slot -> column) from HBM / L2 in every variable pass: nine passes per iteration, each two dependent L2 round trips, VALU
// busy 40 % of that phase (profiles/r03_experiments.txt items 8, 14).  This kernel is for callers whose PRIOR IS KNOWN ON THE HOST when the
// launch is prepared (the circuit plan, the host-pointer decode entry point); the *_dev entry points keep the other kernel.  With the prior on
// the host:
//   * column slots are sorted by (degree, prior value): the circuit-level matrices have 8 - 12 distinct priors and 11 - 15 (degree, prior)
//     classes, so almost every wave of 64 column slots has ONE degree and ONE prior -- both arrive as scalars from a 16-byte record per wave-chunk;
//   * posteriors are indexed by column SLOT (the rows keep column slots in their index registers; outputs are permuted back once per shot), so
//     the variable pass needs no slot -> column table;
//   * the column-side edge list is 16 bits per edge (row slot << 6 | position in the row) and lives in LDS next to V and the check states
//     (61.6 + 70.9 + 24.2 KB for the 1008 x 8857 sector of BASELINE config 5) -- the variable pass issues no global load at all.
// Fewer instructions per edge, same arithmetic (every f64 operation of kernels.py:282-345 in the reference's order, bit-identical outputs):
//   * the state holds UNSIGNED products alpha*min1, alpha*min2 and sign bits already multiplied by the row's total sign, edge k at bit 63 - k
//     of a 64-bit word: R[e] = bfi(0x7fffffff, mag, word << k) -- one shift and one bit-field insert in either pass (kernels.py:311-314);
//   * sign bits are collected with one v_alignbit per edge;
//   * the last chunk of a row runs a body specialised on the number of edges the wave's rows have left (degree 35 = 4 x 8 + 3: the predicated
//     8-edge body spent 40 edge-slots on 35 edges).
// Eligibility (wg2_prepare): clean inputs (minsum_common.h), damping == 1, m <= 1024, row degree <= 40, column degree <= 8, n < 65536, the three
// arrays fit 160 KB of LDS, and at most a quarter of the 64-slot chunks mix classes.  Everything else takes minsum_wg.hip.
#include "common.h"
#include "minsum_common.h"
#include "osd_common.h"      // OSD_CLOCK / osd_timer_buffer: the diagnostic build (make timers) stamps the phases

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

namespace qldpc {

struct Wg2Chunk { double prior; int32_t deg, pure; };       // one per 64 column slots: pure != 0: every slot exists and shares (deg, prior); pure >> 8 = chunks from here on (this one
                                                            // included) that are pure with the same (deg, prior)

struct Wg2Args {
    int m, n, nnz, max_iter, fixed, rdeg, cdeg, nan_deg1_only;
    int eoff[8];                   // EL[eoff[d] + c] = d-th edge (ascending check order) of column slot c, for the slots whose degree exceeds d (a prefix)
    int wq[17];                    // variable pass: wave w takes the chunks wq[w] .. wq[w + 1] - 1 (contiguous, balanced by cost on the host)
    const int32_t *row_of_slot;    // [m]
    const uint8_t *degr;           // [m] degree of the row in slot s (descending)
    const uint16_t *ell_cs;        // [40][m] column SLOT of the k-th edge of row slot s (unused: 0)
    const uint16_t *el;            // [nnz] column-side edge list: row slot << 6 | k
    const Wg2Chunk *chunks;        // [ceil(n / 64)]
    const double *prior_s;         // [n] prior by column slot
    const uint8_t *degc;           // [n] degree by column slot
    const int32_t *slot_of_col;    // [n]
    int64_t B;
    const int8_t *synd; const double *alpha;
    double clip;
    int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
    int offEL, offF, offCH;
    unsigned long long *clk, *dbg;
    int *queue;
};

constexpr int kWg2Chunks = 5;      // 8 edges each: row degree <= 40
// LDS layout: the check states come first at FIXED offsets (m <= 1024), so that their addresses are an immediate offset of the LDS instruction and the
// state address of an edge is two instructions from its 16-bit list entry: (alpha*min1, alpha*min2) pairs [m + 1] at 0, sign words [m + 1] at kWg2OffSI,
// then the posteriors V [n] at kWg2OffV, the edge list, the flags and the chunk records
constexpr int kWg2OffSI = 16 * 1025, kWg2OffV = kWg2OffSI + 8 * 1025 + 8;

__device__ __forceinline__ double w2min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double w2min_s(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double w2max_s(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double w2min_abs2(double a, double b) { double r; asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double w2max_abs2(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
// sign of `t` (bit 31) on the magnitude `mag`: (+-alpha) * mag of kernels.py:311-314 as one bit-field insert
__device__ __forceinline__ double signed_mag(double mag, uint32_t t) {
    uint32_t hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "s"(0x7FFFFFFFu), "v"((uint32_t)__double2hiint(mag)), "v"(t));
    return __hiloint2double((int)hi, __double2loint(mag));
}

struct Wg2Row { bool par; uint32_t pxw; double min1, min2; int arg; uint32_t nA, nB; };

// An LDS read at an ABSOLUTE LDS byte address.  The kernel has no static LDS, so its dynamic LDS starts at address 0 (checked once per launch, below); reading
// through the `lds` symbol instead makes the compiler add the symbol's (zero, but relocatable) address to every hoisted per-edge offset again: one v_add per edge.
typedef __attribute__((address_space(3))) const double wg2_lds_f64;
__device__ __forceinline__ double wg2_lds_read_f64(uint32_t addr) { return *reinterpret_cast<wg2_lds_f64 *>(static_cast<uintptr_t>(addr)); }
typedef double wg2_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const wg2_d2 wg2_lds_f64x2;
typedef __attribute__((address_space(3))) const unsigned long long wg2_lds_u64;
typedef __attribute__((address_space(3))) const uint16_t wg2_lds_u16;
typedef __attribute__((address_space(3))) double wg2_lds_f64w;
__device__ __forceinline__ double2 wg2_lds_read_f64x2(uint32_t addr) { const wg2_d2 t = *reinterpret_cast<wg2_lds_f64x2 *>(static_cast<uintptr_t>(addr)); return make_double2(t.x, t.y); }
__device__ __forceinline__ unsigned long long wg2_lds_read_u64(uint32_t addr) { return *reinterpret_cast<wg2_lds_u64 *>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ uint32_t wg2_lds_read_u16(uint32_t addr) { return *reinterpret_cast<wg2_lds_u16 *>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ void wg2_lds_write_f64(uint32_t addr, double x) { *reinterpret_cast<wg2_lds_f64w *>(static_cast<uintptr_t>(addr)) = x; }

// Eight edge slots K0 .. K0 + 7 of the thread's row.  CNT = 8: every lane of the wave has all eight; 1 .. 7: every lane has exactly CNT of them
// (rows are handed out in degree order, a wave's rows share a degree but for three waves); 0: per-lane degree, predicated.
template <bool NANSEL, bool FIRST, int K0, int CNT>
__device__ __forceinline__ void wg2_chunk(const uint32_t (&idx)[4 * kWg2Chunks], const unsigned char *__restrict__ Vb, int deg, double p1, double p2, uint32_t sA,
                                          uint32_t sB, int argp, double clip, double nclip, Wg2Row &a) {
    constexpr int NE = (CNT == 0) ? 8 : CNT;
    double v[NE];
#pragma unroll
    for (int u = 0; u < NE; u++) {
        const uint32_t w = idx[(K0 + u) >> 1];
        const uint32_t off = ((K0 + u) & 1) ? ((w >> 13) & 0x7FFF8u) : ((w << 3) & 0x7FFF8u);          // column slot * 8
        v[u] = wg2_lds_read_f64(kWg2OffV + off);
    }
#pragma unroll
    for (int u = 0; u < NE; u++) {
        constexpr int dummy = 0; (void)dummy;
        const int k = K0 + u;
        if (CNT != 0 || k < deg) {
            // kernels.py:349,356.  A posterior is never -0.0 here; without degree-1 checks it is never NaN either and v < 0 is its sign bit
            if (NANSEL) a.par ^= (v[u] < 0.0); else a.pxw ^= (uint32_t)__double2hiint(v[u]);
            double x = v[u];
            if (!FIRST) {
                const double mag = (argp == k) ? p2 : p1;                                    // kernels.py:313
                const uint32_t word = (k < 32) ? sA : sB;
                const uint32_t t = (k & 31) ? (word << (k & 31)) : word;                         // the edge's sign (times the row sign) at bit 31
                x = v[u] - signed_mag(mag, t);                                               // kernels.py:311-314, 325
                if (NANSEL) x = (x != x) ? 0.0 : x;                                          // kernels.py:328-329
                x = w2max_s(w2min_s(x, clip), nclip);                                        // kernels.py:330-333
            }
            // x is never -0.0 or NaN here (clean inputs): its sign bit is kernels.py:296-299
            if (k < 32) a.nA = __builtin_amdgcn_alignbit(a.nA, (uint32_t)__double2hiint(x), 31);
            else a.nB = __builtin_amdgcn_alignbit(a.nB, (uint32_t)__double2hiint(x), 31);
            if (fabs(x) < a.min1) a.arg = k;                                                 // kernels.py:301-304 (strict: the first minimum wins)
            a.min2 = w2min(a.min2, w2max_abs2(a.min1, x));                                   // kernels.py:302,305-306
            a.min1 = w2min_abs2(a.min1, x);
        }
    }
}

template <bool NANSEL, bool FIRST, int K0>
__device__ __forceinline__ bool wg2_chunk_any(const uint32_t (&idx)[4 * kWg2Chunks], const unsigned char *__restrict__ Vb, int deg, double p1, double p2, uint32_t sA,
                                              uint32_t sB, int argp, double clip, double nclip, Wg2Row &a) {
    if (!__any(K0 < deg)) return false;
    const int d0 = __builtin_amdgcn_readfirstlane(deg);
    if (__all(K0 + 8 <= deg)) {                                                              // every lane has all eight (also in the three waves that mix degrees)
        wg2_chunk<NANSEL, FIRST, K0, 8>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);
    } else if (__all(deg == d0)) {                                                           // the rule: one degree per wave
        const int left = d0 - K0;
        if (FIRST) wg2_chunk<NANSEL, FIRST, K0, 0>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);       // iteration 0 runs once per shot: no specialised tails
        else switch (left) {
            case 1: wg2_chunk<NANSEL, FIRST, K0, 1>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            case 2: wg2_chunk<NANSEL, FIRST, K0, 2>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            case 3: wg2_chunk<NANSEL, FIRST, K0, 3>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            case 4: wg2_chunk<NANSEL, FIRST, K0, 4>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            case 5: wg2_chunk<NANSEL, FIRST, K0, 5>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            case 6: wg2_chunk<NANSEL, FIRST, K0, 6>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
            default: wg2_chunk<NANSEL, FIRST, K0, 7>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a); break;
        }
    } else {
        wg2_chunk<NANSEL, FIRST, K0, 0>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);
    }
    return true;
}

template <bool NANSEL, bool FIRST>
__device__ __forceinline__ void wg2_row(const uint32_t (&idx)[4 * kWg2Chunks], const unsigned char *__restrict__ Vb, int deg, double p1, double p2, uint32_t sA, uint32_t sB,
                                        int argp, double clip, double nclip, Wg2Row &a) {
    if (!wg2_chunk_any<NANSEL, FIRST, 0>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a)) return;
    if (!wg2_chunk_any<NANSEL, FIRST, 8>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a)) return;
    if (!wg2_chunk_any<NANSEL, FIRST, 16>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a)) return;
    if (!wg2_chunk_any<NANSEL, FIRST, 24>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a)) return;
    wg2_chunk_any<NANSEL, FIRST, 32>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);
}

// the D edges of one column, every lane of the wave has exactly D: entries, then the three words of each check state, then the sum in
// ascending check order (kernels.py:316).  ST64: the check states as 64-bit words, three per row slot (P1, P2, sign word).
template <int D>
__device__ __forceinline__ double wg2_col(const uint32_t (&elb)[8], uint32_t c2) {       // elb [u]: LDS byte address of the rank-u segment of the edge list; c2 = 2 * column slot
    uint32_t e[D];
    double2 pp[D];
    unsigned long long si[D];
#pragma unroll
    for (int u = 0; u < D; u++) e[u] = wg2_lds_read_u16(elb[u] + c2);                        // (one 2-cycle add per edge: the segment bases are scalars)
#pragma unroll
    for (int u = 0; u < D; u++) {                                                            // row slot = e >> 6: byte offsets 16 * slot and 8 * slot
        const uint32_t a = (e[u] >> 2) & 0x3FF0u;
        pp[u] = wg2_lds_read_f64x2(a);
        si[u] = wg2_lds_read_u64(kWg2OffSI + (a >> 1));
    }
    double s = 0.0;                                                                          // kernels.py:279
#pragma unroll
    for (int u = 0; u < D; u++) {
        const uint32_t k = e[u] & 63u;
        const double mag = (k == ((uint32_t)si[u] & 255u)) ? pp[u].y : pp[u].x;              // kernels.py:313
        s += signed_mag(mag, (uint32_t)((si[u] << k) >> 32));                                // kernels.py:316, ascending check order
    }
    return s;
}

__device__ __forceinline__ double wg2_edge(uint32_t e) {
    const uint32_t a = (e >> 2) & 0x3FF0u;
    const double2 pp = wg2_lds_read_f64x2(a);
    const unsigned long long w = wg2_lds_read_u64(kWg2OffSI + (a >> 1));
    const uint32_t k = e & 63u;
    return signed_mag((k == ((uint32_t)w & 255u)) ? pp.y : pp.x, (uint32_t)((w << k) >> 32));
}

template <bool NANSEL>
__global__ __launch_bounds__(1024) void minsum_wg2_kernel(Wg2Args A) {
    extern __shared__ unsigned char lds[];
    double *V = reinterpret_cast<double *>(lds + kWg2OffV);                                  // [n] by column slot
    double2 *PP = reinterpret_cast<double2 *>(lds);                                          // [m + 1] (alpha*min1, alpha*min2), unsigned
    unsigned long long *SI = reinterpret_cast<unsigned long long *>(lds + kWg2OffSI);        // [m + 1] sign word | argmin
    uint16_t *EL = reinterpret_cast<uint16_t *>(lds + A.offEL);                              // [nnz]
    int *unsat = reinterpret_cast<int *>(lds + A.offF);
    Wg2Chunk *CH = reinterpret_cast<Wg2Chunk *>(lds + A.offCH);                              // [ceil(n / 64)]
    const int m = A.m, n = A.n, max_iter = A.max_iter, tid = threadIdx.x, T = blockDim.x;
    const double clip = A.clip, nclip = -A.clip;
    const ClkStamp clk0 = clk_begin(A.clk);
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char *)lds) != 0) __builtin_trap();      // (see wg2_lds_read_f64)
    const int deg = (tid < m) ? (int)A.degr[tid] : 0;                                        // the thread's row slot, constant over shots
    const int row_own = (tid < m) ? A.row_of_slot[tid] : 0;
    uint32_t idx[4 * kWg2Chunks];                                                            // the row's column slots, two per register (unused: slot 0)
#pragma unroll
    for (int p2 = 0; p2 < 4 * kWg2Chunks; p2++) {
        const int k = 2 * p2;
        const uint32_t lo = (tid < m && k < A.rdeg) ? A.ell_cs[(size_t)k * m + tid] : 0u, hi = (tid < m && k + 1 < A.rdeg) ? A.ell_cs[(size_t)(k + 1) * m + tid] : 0u;
        idx[p2] = lo | (hi << 16);
    }
    for (int t = tid; t < (A.nnz + 1) / 2; t += T) reinterpret_cast<uint32_t *>(EL)[t] = reinterpret_cast<const uint32_t *>(A.el)[t];      // (padded to a pair)
    for (int q = tid; q < (n + 63) / 64; q += T) CH[q] = A.chunks[q];
    if (tid == 0) { PP[m] = make_double2(0.0, 0.0); SI[m] = 0ull; }                          // spare state (never selected: kept zero)
    int eoff[8];
#pragma unroll
    for (int d = 0; d < 8; d++) eoff[d] = A.eoff[d];
    uint32_t elb[8];
#pragma unroll
    for (int d = 0; d < 8; d++) elb[d] = (uint32_t)A.offEL + 2u * (uint32_t)A.eoff[d];
    long long t_chk = 0, t_b1 = 0, t_frz = 0, t_var = 0, t_b2 = 0;                           // (diagnostic build only: OSD_CLOCK() is 0 otherwise)
    unsigned long long n_it = 0;
    (void)t_chk; (void)t_b1; (void)t_frz; (void)t_var; (void)t_b2; (void)n_it;
    const int nanpath = NANSEL && (!A.nan_deg1_only || __any(deg == 1));                     // the NaN -> 0 test of kernels.py:328 is for the +-inf messages of degree-1 checks

    for (;;) {
        if (tid == 0) unsat[2] = atomicAdd(A.queue, 1);                                       // unsat[2]: the shot this workgroup decodes next
        __syncthreads();
        const int64_t b = unsat[2];
        if (b >= A.B) break;
        for (int c = tid; c < n; c += T) V[c] = A.prior_s[c];                                // Q_{-1} = prior[col] (kernels.py:263-265)
        if (tid < 2) unsat[tid] = 0;
        const bool csyn = (tid < m) ? (A.synd[b * m + row_own] & 1) : false;
        bool done = false;
        __syncthreads();
        for (int it = 0; it <= max_iter; it++) {
            long long tq = OSD_CLOCK();
            if ((A.fixed || !done) && tid < m) {
                const double alpha = (it < max_iter) ? A.alpha[it] : 0.0;
                Wg2Row a;
                a.par = csyn; a.pxw = 0u; a.min1 = INFINITY; a.min2 = INFINITY; a.arg = 127; a.nA = 0u; a.nB = 0u;
                const unsigned char *Vb = reinterpret_cast<const unsigned char *>(V);
                if (it == 0) {
                    if (nanpath) wg2_row<true, true>(idx, Vb, deg, 0.0, 0.0, 0u, 0u, 127, clip, nclip, a);
                    else wg2_row<false, true>(idx, Vb, deg, 0.0, 0.0, 0u, 0u, 127, clip, nclip, a);
                } else {
                    double p1 = 0.0, p2 = 0.0;
                    uint32_t sA = 0u, sB = 0u;
                    int argp = 127;
                    if (deg > 0) {
                        const double2 t = PP[tid];
                        p1 = t.x; p2 = t.y;
                        const unsigned long long w = SI[tid];
                        sA = (uint32_t)(w >> 32); sB = (uint32_t)w; argp = (int)(sB & 255u);
                    }
                    if (nanpath) wg2_row<true, false>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);
                    else wg2_row<false, false>(idx, Vb, deg, p1, p2, sA, sB, argp, clip, nclip, a);
                }
                a.par ^= (a.pxw >> 31) != 0u;
                if (it >= 1 && !done && a.par) unsat[it & 1] = 1;                            // kernels.py:357-359
                if (it < max_iter && deg > 0) {                                              // kernels.py:285-286
                    // alignbit left the first edge of a word at the highest occupied bit: left-justify (edge k at bit 31 - k % 32)
                    const uint32_t nA = (deg >= 32) ? a.nA : (a.nA << ((32 - deg) & 31));
                    const uint32_t nB = (deg > 32) ? (a.nB << ((64 - deg) & 31)) : 0u;
                    const uint32_t spm = (((uint32_t)csyn ^ (uint32_t)(__popc(nA) + __popc(nB))) & 1u) ? 0xFFFFFFFFu : 0u;     // total sign (kernels.py:289-299)
                    PP[tid] = make_double2(alpha * a.min1, alpha * a.min2);
                    SI[tid] = ((unsigned long long)(nA ^ spm) << 32) | (unsigned long long)(((nB ^ spm) & 0xFFFFFF00u) | (uint32_t)a.arg);
                }
            }
            t_chk += OSD_CLOCK() - tq; tq = OSD_CLOCK();
            __syncthreads();
            t_b1 += OSD_CLOCK() - tq; tq = OSD_CLOCK();
            if (!done) {                                                                     // freeze test (kernels.py:361-364)
                const bool conv = (it >= 1) && (unsat[it & 1] == 0);
                if (conv || it == max_iter) {
                    done = true;
                    for (int j = tid; j < n; j += T) {
                        const double x = (it >= 1) ? V[A.slot_of_col[j]] : 0.0;              // V still holds values_{it-1}
                        A.out_llr[b * n + j] = x;
                        A.out_err[b * n + j] = (x < 0.0) ? 1 : 0;                            // kernels.py:349
                    }
                    if (tid == 0) { A.out_conv[b] = conv ? 1 : 0; A.out_iter[b] = conv ? it - 1 : max_iter - 1; }   // kernels.py:267,362
                    if (A.fixed) __syncthreads();                                            // fixed-work mode goes on: the copy must finish before the variable pass rewrites V
                }
            }
            if (done && !A.fixed) break;
            if (it == max_iter) break;
            if (tid == 0) unsat[(it + 1) & 1] = 0;
            t_frz += OSD_CLOCK() - tq; tq = OSD_CLOCK(); n_it++;
            // variable pass: values_it
            // Every wave owns a contiguous range of chunks (columns are sorted by (degree, prior), so a range is one to three RUNS of chunks of one class): the
            // degree switch and the chunk record are per run, inside a run a pass is the edges, one add for the next chunk's addresses and the store.
            {
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                const int q1 = A.wq[wv + 1];
                int q = A.wq[wv];
                while (q < q1) {
                    const uint4 cur = reinterpret_cast<const uint4 *>(CH)[q];                   // (a broadcast read)
                    const int flag = __builtin_amdgcn_readfirstlane((int)cur.w);
                    uint32_t c2 = 2u * (uint32_t)(64 * q + (tid & 63));
                    if (flag) {
                        const int qe = min(q1, q + (flag >> 8));
                        const int dgu = __builtin_amdgcn_readfirstlane((int)cur.z);
                        const double prior = __hiloint2double(__builtin_amdgcn_readfirstlane((int)cur.y), __builtin_amdgcn_readfirstlane((int)cur.x));
#define QLDPC_WG2_RUN(D) for (; q < qe; q++, c2 += 128u) wg2_lds_write_f64(kWg2OffV + 4u * c2, wg2_col<D>(elb, c2) + prior);      /* kernels.py:320 */
                        switch (dgu) {
                            case 2: QLDPC_WG2_RUN(2) break;
                            case 3: QLDPC_WG2_RUN(3) break;
                            case 4: QLDPC_WG2_RUN(4) break;
                            case 5: QLDPC_WG2_RUN(5) break;
                            case 6: QLDPC_WG2_RUN(6) break;
                            default:                                                         // degree 0, 1, 7, 8
                                for (; q < qe; q++, c2 += 128u) {
                                    double s = 0.0;
                                    for (int d = 0; d < dgu; d++) s += wg2_edge(wg2_lds_read_u16(elb[d] + c2));
                                    wg2_lds_write_f64(kWg2OffV + 4u * c2, s + prior);
                                }
                        }
#undef QLDPC_WG2_RUN
                    } else {                                                                 // a chunk that mixes classes (or the ragged last one): per-lane degree and prior
                        const int c = (int)(c2 >> 1);
                        if (c < n) {
                            const int dj = A.degc[c];
                            double s = 0.0;
                            for (int d = 0; d < dj; d++) s += wg2_edge(EL[eoff[d] + c]);
                            V[c] = s + A.prior_s[c];
                        }
                        q++;
                    }
                }
            }
            t_var += OSD_CLOCK() - tq; tq = OSD_CLOCK();
            __syncthreads();
            t_b2 += OSD_CLOCK() - tq;
        }
        __syncthreads();
    }
#ifdef QLDPC_OSD_TIMERS
    if (A.dbg && (tid & 63) == 0) {          // per-wave sums (the reader divides by the wave-iterations in [17])
        atomicAdd(&A.dbg[17], n_it); atomicAdd(&A.dbg[18], (unsigned long long)t_chk); atomicAdd(&A.dbg[19], (unsigned long long)t_b1);
        atomicAdd(&A.dbg[20], (unsigned long long)t_frz); atomicAdd(&A.dbg[21], (unsigned long long)t_var); atomicAdd(&A.dbg[22], (unsigned long long)t_b2);
    }
#endif
    clk_end(A.clk, clk0);
}

// ------------------------------------------------------------------------------------------ host side
// Device tables of one (graph, prior) pair; cached on the graph handle (callers hold g->mu).
struct Wg2Prep {
    std::vector<double> prior;                 // the key
    bool usable = false;
    int nan_deg1_only = 1, has_deg1 = 0;
    int eoff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int wq[17] = {0};
    size_t lds = 0;
    int offEL = 0, offF = 0, offCH = 0;
    DevBuf row_of_slot, degr, ell_cs, el, chunks, prior_s, degc, slot_of_col;
    ~Wg2Prep() { for (DevBuf *b : {&row_of_slot, &degr, &ell_cs, &el, &chunks, &prior_s, &degc, &slot_of_col}) b->release(); }
};

struct Wg2Cache { std::vector<std::unique_ptr<Wg2Prep>> entries; };

void wg2_cache_free(void *p) { delete static_cast<Wg2Cache *>(p); }

template <class T>
static int up(DevBuf &b, const std::vector<T> &v) {
    int rc = b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T) + 16);
    if (rc != QLDPC_OK) return rc;
    if (!v.empty()) QLDPC_HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return QLDPC_OK;
}

static int wg2_build(const qldpc_graph *g, const double *prior, Wg2Prep &P) {
    const int m = g->m, n = g->n, nnz = g->nnz;
    P.prior.assign(prior, prior + n);
    P.usable = false;
    if (m < 1 || n < 1 || m > 1024 || n >= 65536 || g->max_row_deg > 8 * kWg2Chunks || g->max_col_deg > 8 || nnz < 1) return QLDPC_OK;
    P.offEL = (int)round_up((int64_t)kWg2OffV + (int64_t)n * 8, 16);
    P.offF = (int)round_up((int64_t)P.offEL + ((int64_t)nnz + 1) / 2 * 4, 16);
    P.offCH = P.offF + 32;
    P.lds = (size_t)P.offCH + (size_t)((n + 63) / 64) * sizeof(Wg2Chunk);
    if (P.lds > 160 * 1024) return QLDPC_OK;
    // rows: descending degree (stable), as the graph's own slot order
    std::vector<int32_t> ros(m), slot_of_row(m);
    for (int i = 0; i < m; i++) ros[i] = i;
    auto rdeg = [&](int i) { return g->indptr[i + 1] - g->indptr[i]; };
    auto cdeg = [&](int j) { return g->colptr[j + 1] - g->colptr[j]; };
    std::stable_sort(ros.begin(), ros.end(), [&](int a, int b) { return rdeg(a) > rdeg(b); });
    for (int s = 0; s < m; s++) slot_of_row[ros[s]] = s;
    // columns: descending degree, then prior value (by bit pattern), then index
    std::vector<int32_t> cos(n), slot_of_col(n);
    for (int j = 0; j < n; j++) cos[j] = j;
    auto bits = [&](int j) { uint64_t u; std::memcpy(&u, &prior[j], 8); return u; };
    std::sort(cos.begin(), cos.end(), [&](int a, int b) {
        if (cdeg(a) != cdeg(b)) return cdeg(a) > cdeg(b);
        if (bits(a) != bits(b)) return bits(a) < bits(b);
        return a < b;
    });
    for (int c = 0; c < n; c++) slot_of_col[cos[c]] = c;
    const int nch = (n + 63) / 64;
    std::vector<Wg2Chunk> chunks(nch);
    int mixed = 0;
    for (int q = 0; q < nch; q++) {
        const int c0 = 64 * q, c1 = std::min(n, c0 + 64);
        bool pure = (c1 - c0 == 64);
        for (int c = c0 + 1; c < c1 && pure; c++) pure = cdeg(cos[c]) == cdeg(cos[c0]) && bits(cos[c]) == bits(cos[c0]);
        chunks[q].prior = prior[cos[c0]]; chunks[q].deg = cdeg(cos[c0]); chunks[q].pure = pure ? 1 : 0;
        mixed += pure ? 0 : 1;
    }
    if (4 * mixed > nch + 3) return QLDPC_OK;                       // priors too diverse for class scalars: the table kernel serves this input
    for (int q = nch - 1; q >= 0; q--) {                            // run lengths: chunks from q on that are pure with the class of q
        if (!chunks[q].pure) continue;
        int run = 1;
        if (q + 1 < nch && chunks[q + 1].pure && chunks[q + 1].deg == chunks[q].deg && std::memcmp(&chunks[q + 1].prior, &chunks[q].prior, 8) == 0) run += chunks[q + 1].pure >> 8;
        chunks[q].pure = 1 | (run << 8);
    }
    {   // contiguous chunk ranges of the 16 waves, balanced by cost (a pass: ~36 instruction slots + ~13 per edge; a mixed chunk runs the per-lane form)
        std::vector<int> cost(nch);
        long long total = 0;
        // (measured on the circuit-level matrices: 10 .. 60 per pass and 10 .. 20 per edge all land within 4 % of each other -- 139 chunks over 16 waves leave
        //  a granularity of half a chunk whatever the model; edges alone, without the per-pass term, cost 15 %)
        for (int q = 0; q < nch; q++) { cost[q] = chunks[q].pure ? 36 + 13 * chunks[q].deg : 60 + 20 * g->max_col_deg; total += cost[q]; }
        long long acc = 0;
        int q = 0;
        P.wq[0] = 0;
        for (int w = 0; w < 16; w++) {
            const long long goal = total * (w + 1) / 16;
            while (q < nch && (acc + cost[q] / 2 <= goal || w == 15)) { acc += cost[q]; q++; }
            P.wq[w + 1] = q;
        }
        P.wq[16] = nch;
    }
    std::vector<uint8_t> degr(m), degc(n);
    for (int s = 0; s < m; s++) degr[s] = (uint8_t)rdeg(ros[s]);
    for (int c = 0; c < n; c++) degc[c] = (uint8_t)cdeg(cos[c]);
    std::vector<uint16_t> ell_cs((size_t)8 * kWg2Chunks * m, 0);
    for (int s = 0; s < m; s++) {
        const int i = ros[s];
        for (int e = g->indptr[i]; e < g->indptr[i + 1]; e++) ell_cs[(size_t)(e - g->indptr[i]) * m + s] = (uint16_t)slot_of_col[g->indices[e]];
    }
    int off = 0;
    for (int d = 0; d < 8; d++) {                                   // slots with degree > d form a prefix (descending degree)
        P.eoff[d] = off;
        int cnt = 0;
        while (cnt < n && cdeg(cos[cnt]) > d) cnt++;
        off += cnt;
    }
    std::vector<uint16_t> el(((size_t)nnz + 1) / 2 * 2, 0);
    for (int c = 0; c < n; c++) {
        const int j = cos[c];
        for (int k = g->colptr[j]; k < g->colptr[j + 1]; k++) {       // ascending check order (kernels.py:316)
            const int row = g->rowidx[k], pos = g->csc2csr[k] - g->indptr[row];
            el[(size_t)P.eoff[k - g->colptr[j]] + c] = (uint16_t)((slot_of_row[row] << 6) | pos);
        }
    }
    std::vector<double> prior_s(n);
    for (int c = 0; c < n; c++) prior_s[c] = prior[cos[c]];
    P.has_deg1 = 0; P.nan_deg1_only = 1;
    {
        std::vector<uint8_t> hit(n, 0);
        for (int i = 0; i < m; i++)
            if (rdeg(i) == 1) { P.has_deg1 = 1; const int j = g->indices[g->indptr[i]]; if (hit[j]++) P.nan_deg1_only = 0; }
    }
    int rc;
    if ((rc = up(P.row_of_slot, ros)) || (rc = up(P.degr, degr)) || (rc = up(P.ell_cs, ell_cs)) || (rc = up(P.el, el)) || (rc = up(P.chunks, chunks)) ||
        (rc = up(P.prior_s, prior_s)) || (rc = up(P.degc, degc)) || (rc = up(P.slot_of_col, slot_of_col)))
        return rc;
    P.usable = true;
    return QLDPC_OK;
}

// The tables for (g, prior), built on first use and cached on the handle (a circuit plan asks once per sector; the host-pointer decode entry point
// once per distinct prior).  *out = NULL when this input is not eligible.  Callers hold g->mu.
int wg2_prepare(const qldpc_graph *g, const double *h_prior, const Wg2Prep **out) {
    *out = nullptr;
    if (!h_prior || g->n < 1) return QLDPC_OK;
    if (!g->wg2_cache) g->wg2_cache = new Wg2Cache();
    Wg2Cache *C = static_cast<Wg2Cache *>(g->wg2_cache);
    for (auto &e : C->entries)
        if (std::memcmp(e->prior.data(), h_prior, (size_t)g->n * 8) == 0) { *out = e->usable ? e.get() : nullptr; return QLDPC_OK; }
    if (C->entries.size() >= 4) {                          // a caller cycling through priors: start over once everything in flight is done
        QLDPC_HIP_TRY(hipDeviceSynchronize());
        C->entries.clear();
    }
    std::unique_ptr<Wg2Prep> P(new Wg2Prep());
    const int rc = wg2_build(g, h_prior, *P);
    if (rc != QLDPC_OK) return rc;
    *out = P->usable ? P.get() : nullptr;
    C->entries.push_back(std::move(P));
    return QLDPC_OK;
}

int minsum_wg2_launch(const qldpc_graph *g, const Wg2Prep *P, int64_t B, const int8_t *d_synd, int max_iter, const double *d_alpha, double clip, int flags,
                      int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    Wg2Args A;
    A.m = g->m; A.n = g->n; A.nnz = g->nnz; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
    A.rdeg = g->max_row_deg; A.cdeg = g->max_col_deg; A.nan_deg1_only = P->nan_deg1_only;
    for (int d = 0; d < 8; d++) A.eoff[d] = P->eoff[d];
    for (int w = 0; w <= 16; w++) A.wq[w] = P->wq[w];
    A.row_of_slot = P->row_of_slot.as<int32_t>(); A.degr = P->degr.as<uint8_t>(); A.ell_cs = P->ell_cs.as<uint16_t>(); A.el = P->el.as<uint16_t>();
    A.chunks = P->chunks.as<Wg2Chunk>(); A.prior_s = P->prior_s.as<double>(); A.degc = P->degc.as<uint8_t>(); A.slot_of_col = P->slot_of_col.as<int32_t>();
    A.B = B; A.synd = d_synd; A.alpha = d_alpha; A.clip = clip;
    A.out_err = d_err; A.out_llr = d_llr; A.out_conv = d_conv; A.out_iter = d_iter;
    A.offEL = P->offEL; A.offF = P->offF; A.offCH = P->offCH;
    const unsigned grid = (unsigned)std::min<int64_t>(B, 256 * 2);
    int rc = g->ws_queue.ensure(16);
    if (rc != QLDPC_OK) return rc;
    QLDPC_HIP_TRY(hipMemsetAsync(g->ws_queue.p, 0, 16, stream));
    A.queue = g->ws_queue.as<int>();
    A.clk = g->clk_probe;
    A.dbg = osd_timer_buffer();
    using K = void (*)(Wg2Args);
    K kern = P->has_deg1 ? minsum_wg2_kernel<true> : minsum_wg2_kernel<false>;
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(kern), 160 * 1024)) != QLDPC_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), P->lds, stream, A);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

}  // namespace qldpc
