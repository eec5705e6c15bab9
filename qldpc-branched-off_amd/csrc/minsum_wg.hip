// One-workgroup-per-shot min-sum decoder for LARGE Tanner graphs (circuit-level decoding matrices, BASELINE config 5:
// 1008 x 8785, nnz 30672, check degree <= 35) with the whole decoder state resident in LDS.
//
// The per-edge messages (245 KB/shot in f64) do not fit in LDS, but a min-sum check node only ever emits two magnitudes:
// state per check = (alpha*min1, alpha*min2, index of the first minimum, sign bits of its inputs, total sign) = 24 bytes.
// R[e] = +-(alpha*mag) is reconstructed bit-exactly from it (kernels.py:311-314), both by the variable pass (posterior
// V[j] = prior + sum of R in ASCENDING CHECK ORDER, kernels.py:316-320) and by the next check pass
// (Q = clip(V[col] - R), kernels.py:323-345).  LDS: V[n] f64 + 24 B per check (95 KB for the [[144,12,12]] matrices).
//   check pass: thread per row, ONE sweep over the row (no second pass: the compressed state is the output);
//   variable pass: thread per column; index tables are ELL/slot-major in global memory, so a wave's index loads are
//   contiguous.  The syndrome test of iteration k is a by-product of the sweep of iteration k+1; a workgroup that
//   converged fetches its next shot (persistent grid), so per-shot early exit costs nothing.
// damping != 1 needs Q_old per edge (245 KB per shot): it lives in a slot-major HBM/L2 slab per workgroup (DAMP variants).
// Round 2: rows and columns are handed to threads by SLOT in descending-degree order (a wave's rows / columns share a degree); where a
// thread owns one row for the whole launch (m <= 1024) its column indices stay in registers (RIDX: no index loads in the check pass), and the
// variable pass switches on the wave's column degree into a predicate-free body (wg_lean_col_edges).
#include "common.h"
#include "minsum_common.h"
#include "osd_common.h"      // OSD_CLOCK / osd_timer_buffer: the diagnostic build (make timers) also times the phases of the lean kernel

#include <cmath>
#include <cstdlib>

namespace qldpc {

struct WgArgs {
    int m, n, max_iter, fixed, rdeg, cdeg, nfcheck;
    // rows / columns are handed to threads by SLOT (degree order, or the natural order with QLDPC_FLAG_WG_ROWMAJOR); check state and the
    // entries of ell_var live in row-slot space, posteriors in column space
    const int32_t *row_of_slot, *col_of_slot;
    const uint8_t *degr;           // [m] degree of the row in slot s
    const uint16_t *degc;          // [n] degree of the column in slot c
    const uint16_t *ell_col;       // [round_up(rdeg, 8)][m] by row slot
    const uint32_t *ell_var;       // [cdeg][n] by column slot
    const double *prior_s;         // [n] prior by column slot
    const int32_t *indptr, *indices;   // CSR (rows by original index): the edge-lane check pass reads a row's columns as one coalesced run
    int nan_deg1_only;             // host-verified: no column meets two degree-1 checks, so a NaN can only arise on the edge of a degree-1 check itself
    int edge_lanes;                // QLDPC_FLAG_WG_EDGE_LANES: 16 lanes per check with shuffle reductions (SURVEY 7-6 option B; measured slower)
    int64_t B;
    const int8_t *synd; const double *prior, *alpha;
    double clip;
    int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
    int offP, offI, offF;
    double damping;
    double *qold;          // DAMP kernels: Q_old per edge, slot-major [round_up(rdeg, 8)][m] per workgroup in HBM/L2 (kernels.py:336-345)
    int qstride;           // doubles per workgroup in qold
    double *vglobal;       // VG kernels: posteriors V[n] per workgroup in HBM/L2 (graphs whose V does not fit next to the check states in LDS)
    unsigned long long *clk;   // QLDPC_FLAG_CLOCK_PROBE buffer of the launching plan, else NULL
    unsigned long long *dbg;   // diagnostic build: phase cycle counters (else NULL)
    int *queue;            // next shot to decode (zeroed before the launch): shots are handed out one at a time, so the
                           // workgroups finish together although their shots run 1..max_iter iterations
};


template <bool VG, bool DAMP>
__global__ __launch_bounds__(1024) void minsum_wg_kernel(WgArgs A) {
    extern __shared__ unsigned char lds[];
    double *V;
    if (VG) V = A.vglobal + (size_t)blockIdx.x * A.n; else V = reinterpret_cast<double *>(lds);
    double2 *SP = reinterpret_cast<double2 *>(lds + A.offP);                       // (alpha*min1, alpha*min2) per check
    unsigned long long *SI = reinterpret_cast<unsigned long long *>(lds + A.offI); // bits 0-55 input signs, 56-62 argmin (127 = none), 63 total sign
    int *unsat = reinterpret_cast<int *>(lds + A.offF);
    const int m = A.m, n = A.n, max_iter = A.max_iter, tid = threadIdx.x, T = blockDim.x;
    const double clip = A.clip, damping = A.damping, one_minus_d = 1.0 - A.damping;
    double *Qo = DAMP ? A.qold + (size_t)blockIdx.x * A.qstride : nullptr;

    for (;;) {
        if (tid == 0) unsat[2] = atomicAdd(A.queue, 1);                                       // unsat[2]: the shot this workgroup decodes next
        __syncthreads();
        const int64_t b = unsat[2];
        if (b >= A.B) break;
        for (int j = tid; j < n; j += T) V[j] = A.prior[j];                                  // Q_{-1} = prior[col] (kernels.py:263-265)
        if (tid < 2) unsat[tid] = 0;
        bool done = false;
        __syncthreads();
        for (int it = 0; it <= max_iter; it++) {
            // ---------------- check pass ----------------
            if (A.fixed || !done) {
                const double alpha = (it < max_iter) ? A.alpha[it] : 0.0;
                for (int i = tid; i < m; i += T) {                                           // i = row slot
                    const int deg = A.degr[i];
                    const bool csyn = A.synd[b * m + A.row_of_slot[i]] & 1;
                    double p1p = 0.0, p2p = 0.0;
                    unsigned long long ip = 0ull;
                    if (it > 0 && deg > 0) { const double2 t = SP[i]; p1p = t.x; p2p = t.y; ip = SI[i]; }
                    const int argp = (int)((ip >> 56) & 127);
                    const bool spp = (ip >> 63) & 1;
                    bool par = csyn, sp = csyn;
                    double min1 = INFINITY, min2 = INFINITY;
                    int arg = 127;
                    unsigned long long negbits = 0ull;
                    for (int k = 0; k < deg; k++) {
                        const int col = A.ell_col[(size_t)k * m + i];
                        const double v = V[col];
                        par ^= (v < 0.0);                                                    // kernels.py:349,356
                        double x = v;
                        if (it > 0) {
                            const double mag = (k == argp) ? p2p : p1p;                      // kernels.py:313
                            const double r = (spp != (bool)((ip >> k) & 1)) ? -mag : mag;    // R_{it-1}[e], kernels.py:311-314
                            x = clip_nan(v - r, clip);                                       // kernels.py:325-333
                            if (DAMP) x = clip_only(damping * x + one_minus_d * Qo[(size_t)k * m + i], clip);   // kernels.py:336-342
                            else if (A.nfcheck && prior_not_finite(A.prior[col])) x = NAN;   // kernels.py:336 with Q_old = +-inf (see minsum_common.h)
                        }
                        if (DAMP && it < max_iter) Qo[(size_t)k * m + i] = x;                // kernels.py:344-345 (iteration 0 stores the unclipped prior)
                        const bool neg = !(x >= 0.0);                                        // kernels.py:296-299
                        sp ^= neg;
                        negbits |= (unsigned long long)neg << k;
                        const double a = fabs(x);
                        if (a < min1) { min2 = min1; min1 = a; arg = k; }                    // kernels.py:301-306
                        else if (a < min2) { min2 = a; }
                    }
                    if (it >= 1 && !done && par) unsat[it & 1] = 1;                          // kernels.py:357-359
                    if (it < max_iter && deg > 0) {                                          // kernels.py:285-286
                        SP[i] = make_double2(alpha * min1, alpha * min2);
                        SI[i] = negbits | ((unsigned long long)arg << 56) | ((unsigned long long)sp << 63);
                    }
                }
            }
            __syncthreads();
            // ---------------- freeze test (kernels.py:361-364) ----------------
            if (!done) {
                const bool conv = (it >= 1) && (unsat[it & 1] == 0);
                if (conv || it == max_iter) {
                    done = true;
                    for (int j = tid; j < n; j += T) {
                        const double x = (it >= 1) ? V[j] : 0.0;                             // V still holds values_{it-1}
                        A.out_llr[b * n + j] = x;
                        A.out_err[b * n + j] = (x < 0.0) ? 1 : 0;                            // kernels.py:349
                    }
                    if (tid == 0) { A.out_conv[b] = conv ? 1 : 0; A.out_iter[b] = conv ? it - 1 : max_iter - 1; }   // kernels.py:267,362
                    if (A.fixed) __syncthreads();                                            // fixed-work mode goes on: the copy must finish before the variable pass rewrites V
                }
            }
            if (done && !A.fixed) break;                                                     // uniform: every thread read the same flag
            if (it == max_iter) break;
            if (tid == 0) unsat[(it + 1) & 1] = 0;
            // ---------------- variable pass: values_it ----------------
            for (int c = tid; c < n; c += T) {                                               // c = column slot
                const int j = A.col_of_slot[c];
                double s = 0.0;                                                              // kernels.py:279
                for (int d = 0; d < A.cdeg; d++) {
                    const uint32_t e = A.ell_var[(size_t)d * n + c];
                    if (e == 0xFFFFFFFFu) break;
                    const int i = (int)(e >> 8), k = (int)(e & 255u);
                    const double2 pp = SP[i];
                    const unsigned long long inf = SI[i];
                    const double mag = (k == (int)((inf >> 56) & 127)) ? pp.y : pp.x;
                    s += ((bool)((inf >> 63) & 1) != (bool)((inf >> k) & 1)) ? -mag : mag;   // kernels.py:316, ascending check order
                }
                V[j] = s + A.prior_s[c];                                                     // kernels.py:320
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lean variant.  PMC on the kernel above: 81 VALU + 50 SALU instructions per edge-iteration, i.e. instruction-issue bound.
// When the launcher has verified "clean" inputs (every prior finite and not -0.0, clip finite > 0, every alpha finite > 0)
// no posterior or message can be -0.0 and |q| is never NaN, so:
//   * the sign of q is its sign bit (kernels.py:296 treats -0.0 as positive; it cannot occur here), collected on 32-bit words;
//   * min1/min2 follow min2 = min(min2, max(min1,a)), min1 = min(min1,a) -- identical to the reference's compare chain
//     (kernels.py:301-306) for non-NaN magnitudes; the first-minimum index still comes from the strict compare;
//   * the state stores the products already signed by the row's total sign, so both passes rebuild R with one XOR.
// NANSEL keeps the NaN -> 0 test of kernels.py:328 for graphs with degree-1 checks (their messages are +-inf).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// second operand uniform (held in scalar registers): avoids a VGPR copy of the clip bound per edge
__device__ __forceinline__ double wmin_s(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double wmax_s(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double wmin_abs2(double a, double b) { double r; asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wmax_abs2(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double flip_sign(double x, uint32_t signword) { return __hiloint2double(__double2hiint(x) ^ (int)signword, __double2loint(x)); }

// One row of the check pass, edges taken 8 at a time: the 8 index loads (slot-major table, coalesced), then the 8 posterior
// gathers from LDS are issued back to back before the dependent min/sign chain starts, so their latencies overlap instead
// of adding up per edge (the per-edge loop the compiler emits otherwise waits for memory twice per edge).
// FULL: every lane of the wave has all 8 edges of this chunk (rows are handed out in degree order, so that is the rule, not the
// exception): no per-edge predicate, which is 4 of the ~21 instructions an edge costs.
template <bool NANSEL, bool FIRST, bool DAMP, bool FULL>
__device__ __forceinline__ void wg_lean_chunk_cols(const uint32_t (&c)[8], int m, const double *__restrict__ V, int deg, int k0, double p1s, double p2s,
                                                   uint32_t ip_lo, uint32_t ip_hi, int argp, double clip, double nclip, double damping, double one_minus_d,
                                                   double *__restrict__ qo, bool store_q, bool &par, double &min1, double &min2, int &arg, uint32_t &nlo,
                                                   uint32_t &nhi) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = V[c[u]];
    double qprev[8];
    if (DAMP && !FIRST) {
#pragma unroll
        for (int u = 0; u < 8; u++) qprev[u] = qo[(size_t)(k0 + u) * m];                     // slab rows are padded to a multiple of 8 slots
    }
    const uint32_t pw = (k0 < 32) ? (ip_lo >> k0) : (ip_hi >> (k0 - 32));                    // previous sign bits of this chunk
    const int au = argp - k0;
    uint32_t cb = 0u, pxw = 0u;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (FULL || k0 + u < deg) {
            // kernels.py:349,356.  A posterior is never -0.0 here; without degree-1 checks it is never NaN either, and v < 0 is its sign bit
            if (NANSEL) par ^= (v[u] < 0.0); else pxw ^= (uint32_t)__double2hiint(v[u]);
            double x = v[u];
            if (!FIRST) {
                const double mag = (u == au) ? p2s : p1s;                                    // kernels.py:313 (already carries the row sign)
                const double r = flip_sign(mag, (pw >> u) << 31);                            // R_{it-1}[e], kernels.py:311-314
                x = v[u] - r;                                                                // kernels.py:325
                if (NANSEL) x = (x != x) ? 0.0 : x;                                          // kernels.py:328-329
                x = wmax_s(wmin_s(x, clip), nclip);                                          // kernels.py:330-333
                if (DAMP) x = wmax_s(wmin_s(damping * x + one_minus_d * qprev[u], clip), nclip);   // kernels.py:336-342 (finite operands)
            }
            if (DAMP && store_q) qo[(size_t)(k0 + u) * m] = x;                               // kernels.py:344-345
            // without damping x is never -0.0 or NaN here (see above); a damped x may underflow to -0.0, which counts as positive
            cb |= (DAMP ? (x < 0.0 ? 1u : 0u) : ((uint32_t)__double2hiint(x) >> 31)) << u;
            if (fabs(x) < min1) arg = k0 + u;                                                // kernels.py:301-304 (strict: first minimum wins)
            min2 = wmin(min2, wmax_abs2(min1, x));                                           // kernels.py:302,305-306
            min1 = wmin_abs2(min1, x);
        }
    }
    par ^= (pxw >> 31) != 0u;
    if (k0 < 32) nlo |= cb << k0; else nhi |= cb << (k0 - 32);
}

template <bool NANSEL, bool FIRST, bool DAMP, bool FULL>
__device__ __forceinline__ void wg_lean_chunk(const uint16_t *__restrict__ ec, int m, const double *__restrict__ V, int deg, int k0, double p1s, double p2s,
                                              uint32_t ip_lo, uint32_t ip_hi, int argp, double clip, double nclip, double damping, double one_minus_d,
                                              double *__restrict__ qo, bool store_q, bool &par, double &min1, double &min2, int &arg, uint32_t &nlo,
                                              uint32_t &nhi) {
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = ec[(size_t)(k0 + u) * m];                            // table rows are padded to a multiple of 8 with column 0
    wg_lean_chunk_cols<NANSEL, FIRST, DAMP, FULL>(c, m, V, deg, k0, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, damping, one_minus_d, qo, store_q, par, min1,
                                                  min2, arg, nlo, nhi);
}

// RIDX kernels (m <= blockDim: a thread owns ONE row for the whole launch, check degree <= 40): the row's column indices stay in 20 registers,
// two per register, so the check pass has no index loads at all (in-kernel stamps: 18.9 k -> 15.5 k cycles per iteration for the check pass)
constexpr int kIdxChunks = 5;
template <bool NANSEL, bool FIRST>
__device__ __forceinline__ void wg_lean_row_idx(const uint32_t (&idx)[4 * kIdxChunks], int m, const double *__restrict__ V, int deg, double p1s, double p2s,
                                                uint32_t ip_lo, uint32_t ip_hi, int argp, double clip, double nclip, bool &par, double &min1, double &min2,
                                                int &arg, uint32_t &nlo, uint32_t &nhi) {
#pragma unroll
    for (int ch = 0; ch < kIdxChunks; ch++) {
        const int k0 = 8 * ch;
        if (!__any(k0 < deg)) break;
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u += 2) { c[u] = idx[4 * ch + u / 2] & 0xFFFFu; c[u + 1] = idx[4 * ch + u / 2] >> 16; }
        if (__all(k0 + 8 <= deg))
            wg_lean_chunk_cols<NANSEL, FIRST, false, true>(c, m, V, deg, k0, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, 1.0, 0.0, nullptr, false, par, min1, min2,
                                                           arg, nlo, nhi);
        else
            wg_lean_chunk_cols<NANSEL, FIRST, false, false>(c, m, V, deg, k0, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, 1.0, 0.0, nullptr, false, par, min1, min2,
                                                            arg, nlo, nhi);
    }
}

template <bool NANSEL, bool FIRST, bool DAMP>
__device__ __forceinline__ void wg_lean_row(const uint16_t *__restrict__ ec, int m, const double *__restrict__ V, int deg, double p1s, double p2s,
                                            uint32_t ip_lo, uint32_t ip_hi, int argp, double clip, double nclip, double damping, double one_minus_d,
                                            double *__restrict__ qo, bool store_q, bool &par, double &min1, double &min2, int &arg, uint32_t &nlo,
                                            uint32_t &nhi) {
    for (int k0 = 0; k0 < deg; k0 += 8) {
        if (__all(k0 + 8 <= deg))
            wg_lean_chunk<NANSEL, FIRST, DAMP, true>(ec, m, V, deg, k0, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, damping, one_minus_d, qo, store_q, par, min1,
                                                     min2, arg, nlo, nhi);
        else
            wg_lean_chunk<NANSEL, FIRST, DAMP, false>(ec, m, V, deg, k0, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, damping, one_minus_d, qo, store_q, par, min1,
                                                      min2, arg, nlo, nhi);
    }
}

// the edges of one column of the variable pass when every lane of the wave has exactly D of them (column slots are handed out in degree order,
// so that is the rule): D table loads, 2 D state reads, no predicate -- the 4-at-a-time loop below it runs three of every four columns of the
// circuit-level matrices (degree 3, 5, 6, 2) through its predicated tail
template <int D>
__device__ __forceinline__ double wg_lean_col_edges(const uint32_t *__restrict__ ev, int n, const double2 *__restrict__ SP, const uint2 *__restrict__ SI) {
    uint32_t e[D];
    double2 pp[D];
    uint2 si[D];
#pragma unroll
    for (int u = 0; u < D; u++) e[u] = ev[(size_t)u * n];
#pragma unroll
    for (int u = 0; u < D; u++) { const uint32_t i = e[u] >> 8; pp[u] = SP[i]; si[u] = SI[i]; }
    double s = 0.0;                                                                          // kernels.py:279
#pragma unroll
    for (int u = 0; u < D; u++) {
        const uint32_t k = e[u] & 255u;
        const double mag = (k == (si[u].y >> 24)) ? pp[u].y : pp[u].x;
        const uint32_t w = (k < 32u) ? (si[u].x >> k) : (si[u].y >> (k - 32u));
        s += flip_sign(mag, w << 31);                                                        // kernels.py:316, ascending check order
    }
    return s;
}

template <bool NANSEL, bool VG, bool DAMP, bool RIDX = false>
__global__ __launch_bounds__(1024) void minsum_wg_lean_kernel(WgArgs A) {
    extern __shared__ unsigned char lds[];
    double *V;
    if (VG) V = A.vglobal + (size_t)blockIdx.x * A.n; else V = reinterpret_cast<double *>(lds);
    double2 *SP = reinterpret_cast<double2 *>(lds + A.offP);        // (alpha*min1, alpha*min2), both already multiplied by the row's total sign; [m + 1]
    uint2 *SI = reinterpret_cast<uint2 *>(lds + A.offI);            // .x = sign bits 0-31, .y = sign bits 32-55 | argmin << 24; [m + 1]
    int *unsat = reinterpret_cast<int *>(lds + A.offF);
    const int m = A.m, n = A.n, max_iter = A.max_iter, tid = threadIdx.x, T = blockDim.x;
    const double clip = A.clip, nclip = -A.clip, damping = A.damping, one_minus_d = 1.0 - A.damping;
    double *Qo = DAMP ? A.qold + (size_t)blockIdx.x * A.qstride : nullptr;
    const ClkStamp clk0 = clk_begin(A.clk);
    const int deg_own = (tid < m) ? (int)A.degr[tid] : 0;                                    // the thread's first row slot, constant over shots
    const int row_own = (tid < m) ? A.row_of_slot[tid] : 0;
    if (tid == 0) { SP[m] = make_double2(0.0, 0.0); SI[m] = make_uint2(0u, 0u); }            // dummy check read by padded column slots
    uint32_t idx[4 * kIdxChunks];                                                            // RIDX: the row's columns, two per register (unused slots: column 0)
    if (RIDX) {
#pragma unroll
        for (int p2 = 0; p2 < 4 * kIdxChunks; p2++) {
            const int k = 2 * p2;
            const uint32_t lo = (tid < m && k < A.rdeg) ? A.ell_col[(size_t)k * m + tid] : 0u, hi = (tid < m && k + 1 < A.rdeg) ? A.ell_col[(size_t)(k + 1) * m + tid] : 0u;
            idx[p2] = lo | (hi << 16);
        }
    }
    long long t_chk = 0, t_b1 = 0, t_frz = 0, t_var = 0, t_b2 = 0;                           // (diagnostic build only: OSD_CLOCK() is 0 otherwise)
    unsigned long long n_it = 0;
    (void)t_chk; (void)t_b1; (void)t_frz; (void)t_var; (void)t_b2; (void)n_it;

    for (;;) {
        if (tid == 0) unsat[2] = atomicAdd(A.queue, 1);                                       // unsat[2]: the shot this workgroup decodes next
        __syncthreads();
        const int64_t b = unsat[2];
        if (b >= A.B) break;
        for (int j = tid; j < n; j += T) V[j] = A.prior[j];                                  // Q_{-1} = prior[col] (kernels.py:263-265)
        if (tid < 2) unsat[tid] = 0;
        const bool csyn_own = (tid < m) ? (A.synd[b * m + row_own] & 1) : false;
        bool done = false;
        __syncthreads();
        for (int it = 0; it <= max_iter; it++) {
            long long tq = OSD_CLOCK();
            if (A.fixed || !done) {
                const double alpha = (it < max_iter) ? A.alpha[it] : 0.0;
#ifdef QLDPC_EXPERIMENTS
                const bool x_edge_lanes = A.edge_lanes;
#else
                constexpr bool x_edge_lanes = false;
#endif
                if (!DAMP && x_edge_lanes) {
                    // SURVEY 7-6 option B, kept as a measured alternative: a check is owned by 16 lanes (lane = edge k, k + 16, k + 32), the row's
                    // min1 / min2 / first-argmin come from a 4-step shuffle butterfly, sign bits and the parity from ballots.  The reduction
                    // (value, index) -> smallest value, then smallest index reproduces "first strict minimum" (kernels.py:301-306); min2 is
                    // the second smallest with multiplicity = min(winner's min2, loser's min1).
                    const int gl = tid & 15, gid = tid >> 4, ngroups = T >> 4, gsh = (tid & 63) & ~15;
                    for (int s0 = 0; s0 < m; s0 += ngroups) {                                // uniform trip count: every lane takes part in the ballots
                        const int s = s0 + gid;
                        const bool live = s < m;
                        const int row = live ? A.row_of_slot[s] : 0;
                        const int deg = live ? (int)A.degr[s] : 0, e0 = live ? A.indptr[row] : 0;
                        const bool csyn = live ? (bool)(A.synd[b * m + row] & 1) : false;
                        double p1s = 0.0, p2s = 0.0;
                        uint32_t ip_lo = 0u, ip_hi = 0u;
                        int argp = 127;
                        if (it > 0 && deg > 0) {
                            const double2 t = SP[s]; const uint2 u = SI[s];
                            p1s = t.x; p2s = t.y; argp = (int)(u.y >> 24); ip_lo = u.x; ip_hi = u.y & 0x00FFFFFFu;
                        }
                        double m1 = INFINITY, m2 = INFINITY;
                        int k1 = 127;
                        uint32_t nlo = 0u, nhi = 0u, parbits = 0u;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const int k = 16 * c + gl;
                            const bool has = k < deg;
                            bool vneg = false, xneg = false;
                            if (has) {
                                const double v = V[A.indices[e0 + k]];
                                vneg = v < 0.0;                                              // kernels.py:349,356
                                double x = v;
                                if (it > 0) {
                                    const double mag = (k == argp) ? p2s : p1s;              // kernels.py:313
                                    const uint32_t pwb = (k < 32) ? (ip_lo >> k) : (ip_hi >> (k - 32));
                                    x = v - flip_sign(mag, pwb << 31);                       // kernels.py:311-314, 325
                                    if (NANSEL) x = (x != x) ? 0.0 : x;                      // kernels.py:328-329
                                    x = wmax_s(wmin_s(x, clip), nclip);                      // kernels.py:330-333
                                }
                                xneg = ((uint32_t)__double2hiint(x) >> 31) != 0u;
                                const double a = fabs(x);
                                if (a < m1) { m2 = m1; m1 = a; k1 = k; } else if (a < m2) { m2 = a; }   // kernels.py:301-306 on this lane's edges
                            }
                            const uint32_t gv = (uint32_t)(__ballot(has && vneg) >> gsh) & 0xFFFFu, gx = (uint32_t)(__ballot(has && xneg) >> gsh) & 0xFFFFu;
                            parbits ^= gv;
                            if (c < 2) nlo |= gx << (16 * c); else nhi |= gx;
                        }
#pragma unroll
                        for (int off = 8; off > 0; off >>= 1) {
                            const double om1 = __shfl_xor(m1, off, 16), om2 = __shfl_xor(m2, off, 16);
                            const int ok1 = __shfl_xor(k1, off, 16);
                            const bool take = (om1 < m1) || (om1 == m1 && ok1 < k1);
                            const double lose1 = take ? m1 : om1, win2 = take ? om2 : m2;
                            m1 = take ? om1 : m1; k1 = take ? ok1 : k1;
                            m2 = wmin(win2, lose1);
                        }
                        if (gl == 0 && live) {
                            const bool par = csyn ^ (bool)(__popc(parbits) & 1);
                            if (it >= 1 && !done && par) unsat[it & 1] = 1;                  // kernels.py:357-359
                            if (it < max_iter && deg > 0) {                                  // kernels.py:285-286
                                const uint32_t sp = ((uint32_t)csyn ^ (uint32_t)(__popc(nlo) + __popc(nhi))) << 31;
                                SP[s] = make_double2(flip_sign(alpha * m1, sp), flip_sign(alpha * m2, sp));
                                SI[s] = make_uint2(nlo, nhi | ((uint32_t)k1 << 24));
                            }
                        }
                    }
                } else
                for (int i = tid; i < m; i += T) {                                           // i = row slot
                    const int deg = (i == tid) ? deg_own : (int)A.degr[i];
                    const bool csyn = (i == tid) ? csyn_own : (bool)(A.synd[b * m + A.row_of_slot[i]] & 1);
                    bool par = csyn;
                    double min1 = INFINITY, min2 = INFINITY;
                    int arg = 127;
                    uint32_t nlo = 0u, nhi = 0u;
                    if (it == 0) {
                        if (RIDX) wg_lean_row_idx<NANSEL, true>(idx, m, V, deg, 0.0, 0.0, 0u, 0u, 127, clip, nclip, par, min1, min2, arg, nlo, nhi);
                        else
                        wg_lean_row<NANSEL, true, DAMP>(A.ell_col + i, m, V, deg, 0.0, 0.0, 0u, 0u, 127, clip, nclip, damping, one_minus_d, DAMP ? Qo + i : nullptr,
                                                        it < max_iter, par, min1, min2, arg, nlo, nhi);
                    } else {
                        double p1s = 0.0, p2s = 0.0;
                        uint32_t ip_lo = 0u, ip_hi = 0u;
                        int argp = 127;
                        if (deg > 0) {
                            const double2 t = SP[i]; const uint2 u = SI[i];
                            p1s = t.x; p2s = t.y; argp = (int)(u.y >> 24); ip_lo = u.x; ip_hi = u.y & 0x00FFFFFFu;
                        }
                        if (RIDX) {
                            // the NaN -> 0 test (kernels.py:328) is for the +-inf messages of degree-1 checks; when no column meets two of them
                            // (host-verified) only the waves that hold such rows need it
                            if (NANSEL && (!A.nan_deg1_only || __any(deg == 1)))
                                wg_lean_row_idx<true, false>(idx, m, V, deg, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, par, min1, min2, arg, nlo, nhi);
                            else
                                wg_lean_row_idx<false, false>(idx, m, V, deg, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, par, min1, min2, arg, nlo, nhi);
                        }
                        else
                        wg_lean_row<NANSEL, false, DAMP>(A.ell_col + i, m, V, deg, p1s, p2s, ip_lo, ip_hi, argp, clip, nclip, damping, one_minus_d,
                                                         DAMP ? Qo + i : nullptr, it < max_iter, par, min1, min2, arg, nlo, nhi);
                    }
                    if (it >= 1 && !done && par) unsat[it & 1] = 1;                          // kernels.py:357-359
                    if (it < max_iter && deg > 0) {                                          // kernels.py:285-286
                        const uint32_t sp = ((uint32_t)csyn ^ (uint32_t)(__popc(nlo) + __popc(nhi))) << 31;     // total sign (kernels.py:289-299)
                        SP[i] = make_double2(flip_sign(alpha * min1, sp), flip_sign(alpha * min2, sp));
                        SI[i] = make_uint2(nlo, nhi | ((uint32_t)arg << 24));
                    }
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
                        const double x = (it >= 1) ? V[j] : 0.0;                             // V still holds values_{it-1}
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
            // variable pass: values_it.  Same batching: the column's edge slots are loaded 4 at a time, then the 4 check states.
            for (int c = tid; c < n; c += T) {                                               // c = column slot
                const int j = A.col_of_slot[c], cdj = A.degc[c];
                const double pr = A.prior_s[c];
                double s = 0.0;                                                              // kernels.py:279
                const int dmax = __builtin_amdgcn_readfirstlane(cdj);
                if (dmax >= 2 && dmax <= 6 && __all(cdj == dmax)) {                          // the whole wave has one degree
                    const uint32_t *ev = A.ell_var + c;
                    switch (dmax) {
                        case 2: s = wg_lean_col_edges<2>(ev, n, SP, SI); break;
                        case 3: s = wg_lean_col_edges<3>(ev, n, SP, SI); break;
                        case 4: s = wg_lean_col_edges<4>(ev, n, SP, SI); break;
                        case 5: s = wg_lean_col_edges<5>(ev, n, SP, SI); break;
                        default: s = wg_lean_col_edges<6>(ev, n, SP, SI); break;
                    }
                    V[j] = s + pr;                                                           // kernels.py:320
                    continue;
                }
                for (int d0 = 0; d0 < cdj; d0 += 4) {                                        // (slots of one degree share a wave: no idle chunk)
                    uint32_t e[4];
                    double2 pp[4];
                    uint2 si[4];
                    if (__all(d0 + 4 <= cdj)) {                                              // every lane has all four edges: no predicates
#pragma unroll
                        for (int u = 0; u < 4; u++) e[u] = A.ell_var[(size_t)(d0 + u) * n + c];
#pragma unroll
                        for (int u = 0; u < 4; u++) { const uint32_t i = e[u] >> 8; pp[u] = SP[i]; si[u] = SI[i]; }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t k = e[u] & 255u;
                            const double mag = (k == (si[u].y >> 24)) ? pp[u].y : pp[u].x;
                            const uint32_t w = (k < 32u) ? (si[u].x >> k) : (si[u].y >> (k - 32u));
                            s += flip_sign(mag, w << 31);                                    // kernels.py:316, ascending check order
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = (d0 + u < A.cdeg) ? A.ell_var[(size_t)(d0 + u) * n + c] : 0xFFFFFFFFu;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t i = min(e[u] >> 8, (uint32_t)m);                      // empty slot -> the dummy check
                        pp[u] = SP[i]; si[u] = SI[i];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (e[u] != 0xFFFFFFFFu) {
                            const uint32_t k = e[u] & 255u;
                            const double mag = (k == (si[u].y >> 24)) ? pp[u].y : pp[u].x;
                            const uint32_t w = (k < 32u) ? (si[u].x >> k) : (si[u].y >> (k - 32u));
                            s += flip_sign(mag, w << 31);                                    // kernels.py:316, ascending check order
                        }
                    }
                }
                V[j] = s + pr;                                                               // kernels.py:320
            }
            t_var += OSD_CLOCK() - tq; tq = OSD_CLOCK();
            __syncthreads();
            t_b2 += OSD_CLOCK() - tq;
        }
        __syncthreads();
    }
#ifdef QLDPC_OSD_TIMERS
    if (A.dbg && (tid & 63) == 0) {          // per-wave sums (the reader divides by the wave-iterations in [1])
        atomicAdd(&A.dbg[17], n_it); atomicAdd(&A.dbg[18], (unsigned long long)t_chk); atomicAdd(&A.dbg[19], (unsigned long long)t_b1);      // slots 16.. : BP
        atomicAdd(&A.dbg[20], (unsigned long long)t_frz); atomicAdd(&A.dbg[21], (unsigned long long)t_var); atomicAdd(&A.dbg[22], (unsigned long long)t_b2);
    }
#endif
    clk_end(A.clk, clk0);
}

__global__ void permute_prior_kernel(int n, const int32_t *__restrict__ col_of_slot, const double *__restrict__ prior, double *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) out[c] = prior[col_of_slot[c]];
}

static size_t wg_lds_bytes(const qldpc_graph *g, bool vg, int &offP, int &offI, int &offF) {
    offP = vg ? 0 : (int)round_up((int64_t)g->n * 8, 16);
    offI = offP + (g->m + 1) * 16;     // one spare check state: the target of empty column slots in the lean kernel
    offF = offI + (g->m + 1) * 8;
    return (size_t)offF + 16;
}

// 0: not supported, 1: everything in LDS, 2: check states in LDS, posteriors in global memory
static int wg_mode(const qldpc_graph *g, double damping, int flags) {
    (void)damping;                      // damping != 1 keeps Q_old in an HBM/L2 slab
    if (!g->d_ell_col || !g->d_ell_var || !g->d_ell_col_s || !g->d_ell_var_s) return 0;
    if (g->m <= 0 || g->n <= 0 || g->max_row_deg > 56) return 0;
    int a, b, c;
    if (wg_lds_bytes(g, false, a, b, c) <= 160 * 1024 && !(flags & QLDPC_FLAG_WG_VGLOBAL)) return 1;
    return wg_lds_bytes(g, true, a, b, c) <= 160 * 1024 ? 2 : 0;
}

bool wg_supported(const qldpc_graph *g, double damping) { return wg_mode(g, damping, 0) != 0; }

int minsum_wg_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter, const double *d_alpha,
                     double damping, double clip, int flags, bool clean, int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    WgArgs A;
    A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
    A.rdeg = g->max_row_deg; A.cdeg = g->max_col_deg; A.nfcheck = (flags & QLDPC_FLAG_INTERNAL_PRIOR_FINITE) ? 0 : 1;
    const bool natural = (flags & QLDPC_FLAG_WG_ROWMAJOR) != 0;
    A.row_of_slot = natural ? g->d_identity : g->d_row_of_slot;
    A.col_of_slot = natural ? g->d_identity : g->d_col_of_slot;
    A.degr = natural ? g->d_deg_of_row : g->d_deg_of_rslot;
    A.degc = natural ? g->d_deg_of_col : g->d_deg_of_cslot;
    A.ell_col = natural ? g->d_ell_col : g->d_ell_col_s;
    A.ell_var = natural ? g->d_ell_var : g->d_ell_var_s;
    A.indptr = g->d_indptr; A.indices = g->d_indices;
#ifndef QLDPC_EXPERIMENTS
    if (flags & (QLDPC_FLAG_WG_EDGE_LANES | QLDPC_FLAG_WG_IDXLOAD)) {
        set_error("this decoder variant (flags %#x) is a measured-and-rejected experiment: it exists in libqldpc_hip_experiments.so only (make experiments)", flags);
        return QLDPC_ERR_UNSUPPORTED;
    }
#endif
    A.edge_lanes = ((flags & QLDPC_FLAG_WG_EDGE_LANES) && g->max_row_deg <= 48) ? 1 : 0;
    A.B = B; A.synd = d_synd; A.prior = d_prior; A.alpha = d_alpha; A.clip = clip;
    A.out_err = d_err; A.out_llr = d_llr; A.out_conv = d_conv; A.out_iter = d_iter;
    const bool vg = (wg_mode(g, damping, flags) == 2), damp = (damping != 1.0);
    const size_t lds = wg_lds_bytes(g, vg, A.offP, A.offI, A.offF);
    bool has_deg1 = false;
    for (int i = 0; i < g->m; i++) has_deg1 = has_deg1 || (g->indptr[i + 1] - g->indptr[i] == 1);
    A.nan_deg1_only = 1;
    if (has_deg1) {
        std::vector<uint8_t> hit(g->n, 0);
        for (int i = 0; i < g->m && A.nan_deg1_only; i++)
            if (g->indptr[i + 1] - g->indptr[i] == 1) { const int j = g->indices[g->indptr[i]]; if (hit[j]++) A.nan_deg1_only = 0; }
    }
    const int block = (g->m > 512 || g->n > 4096) ? 1024 : 512;
    const unsigned grid = (unsigned)std::min<int64_t>(B, 256 * 2);
    int rcq = g->ws_queue.ensure(16);
    if (rcq != QLDPC_OK) return rcq;
    QLDPC_HIP_TRY(hipMemsetAsync(g->ws_queue.p, 0, 16, stream));
    A.queue = g->ws_queue.as<int>();
    A.clk = g->clk_probe;
    A.dbg = osd_timer_buffer();      // NULL unless built with -DQLDPC_OSD_TIMERS (make timers)
    if ((rcq = g->ws_prior.ensure((size_t)g->n * 8)) != QLDPC_OK) return rcq;                // the prior in column-slot order (per launch: it is an input)
    hipLaunchKernelGGL(permute_prior_kernel, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, stream, g->n, A.col_of_slot, d_prior, g->ws_prior.as<double>());
    A.prior_s = g->ws_prior.as<double>();
    A.vglobal = nullptr; A.qold = nullptr; A.qstride = 0; A.damping = damping;
    if (vg) {
        if ((rcq = g->ws_vals.ensure((size_t)grid * g->n * 8)) != QLDPC_OK) return rcq;
        A.vglobal = g->ws_vals.as<double>();
    }
    if (damp) {
        A.qstride = (int)round_up(std::max(g->max_row_deg, 1), 8) * g->m;
        if ((rcq = g->ws_qold.ensure((size_t)grid * A.qstride * 8)) != QLDPC_OK) return rcq;
        A.qold = g->ws_qold.as<double>();
    }
    const bool lean = clean && std::isfinite(damping) && !(flags & QLDPC_FLAG_WG_GENERIC);
    using K = void (*)(WgArgs);
    // [lean][nansel][vg][damp]
    static const K table[2][2][2][2] = {
        {{{minsum_wg_kernel<false, false>, minsum_wg_kernel<false, true>}, {minsum_wg_kernel<true, false>, minsum_wg_kernel<true, true>}},
         {{minsum_wg_kernel<false, false>, minsum_wg_kernel<false, true>}, {minsum_wg_kernel<true, false>, minsum_wg_kernel<true, true>}}},
        {{{minsum_wg_lean_kernel<false, false, false>, minsum_wg_lean_kernel<false, false, true>},
          {minsum_wg_lean_kernel<false, true, false>, minsum_wg_lean_kernel<false, true, true>}},
         {{minsum_wg_lean_kernel<true, false, false>, minsum_wg_lean_kernel<true, false, true>},
          {minsum_wg_lean_kernel<true, true, false>, minsum_wg_lean_kernel<true, true, true>}}}};
    K kern = table[lean ? 1 : 0][has_deg1 ? 1 : 0][vg ? 1 : 0][damp ? 1 : 0];
    // a thread owns one row for the whole launch: its column indices stay in registers (QLDPC_FLAG_WG_IDXLOAD keeps the per-iteration index loads)
    if (lean && !vg && !damp && !A.edge_lanes && !(flags & QLDPC_FLAG_WG_IDXLOAD) && g->m <= block && g->max_row_deg <= 8 * kIdxChunks)
        kern = has_deg1 ? minsum_wg_lean_kernel<true, false, false, true> : minsum_wg_lean_kernel<false, false, false, true>;
    if ((rcq = ensure_max_lds(g->device, reinterpret_cast<const void *>(kern), 160 * 1024)) != QLDPC_OK) return rcq;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, A);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

}  // namespace qldpc
