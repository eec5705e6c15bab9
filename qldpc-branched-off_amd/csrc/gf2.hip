// GF(2) Gauss-Jordan elimination on bit-packed rows (a7/a8) and batched OSD-0 (a9).
//
// One workgroup owns one matrix.  The reference (src/decoding/kernels.py:48-96) walks the columns one by one;
// here a single parallel search returns, for the current `row`, the FIRST column >= col that has a one in some
// row >= row and the SMALLEST such row -- exactly the pivot the column-by-column loop would reach next -- so the
// number of workgroup-wide steps is rank+1, not n.  Rows that became all-zero are remembered and skipped.
// The pivot row has only zeros left of its pivot column (all earlier columns were either eliminated or had no
// one at/below `row`), so XOR-ing words >= pivot_col/64 reproduces the reference's full-row XOR bit for bit.
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"
#include "osd_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace qldpc {

struct ElimShared {      // carved from dynamic LDS
    unsigned long long *key;   // [1] packed (col << 32) | row of the next pivot
    int *cnt;                  // [1]
    uint64_t *prow;            // [nwords] pivot row
    int *list;                 // [m] rows to update
    uint8_t *dead;             // [m] all-zero rows
};

__device__ inline ElimShared carve_elim(unsigned char *base, int m, int nwords) {
    ElimShared S;
    S.key = reinterpret_cast<unsigned long long *>(base);
    S.prow = reinterpret_cast<uint64_t *>(base + 16);
    S.list = reinterpret_cast<int *>(base + 16 + (size_t)nwords * 8);
    S.cnt = S.list + m;
    S.dead = reinterpret_cast<uint8_t *>(S.cnt + 2);
    return S;
}
static size_t elim_lds_bytes(int m, int nwords) { return 16 + (size_t)nwords * 8 + (size_t)m * 4 + 8 + (size_t)m + 16; }

// In-place elimination of A[m][nwords] (global), b[m].  Returns the number of pivots (uniform).
template <class PIdx>
__device__ int eliminate_packed(uint64_t *A, uint8_t *b, int m, int n, int nwords, PIdx *pivot_rows, PIdx *pivot_cols, ElimShared S) {
    const int tid = threadIdx.x, T = blockDim.x;
    for (int r = tid; r < m; r += T) S.dead[r] = 0;
    int row = 0, col = 0, np = 0;
    __syncthreads();
    while (col < n && row < m) {                                              // kernels.py:64-66
        if (tid == 0) *S.key = ~0ull;
        __syncthreads();
        for (int r = row + tid; r < m; r += T) {                              // kernels.py:71-75, all candidate columns at once
            if (S.dead[r]) continue;
            int w = col >> 6;
            uint64_t x = A[(size_t)r * nwords + w] & (~0ull << (col & 63));
            while (x == 0 && ++w < nwords) x = A[(size_t)r * nwords + w];
            if (x == 0) { S.dead[r] = 1; continue; }
            const unsigned long long pos = (unsigned long long)w * 64 + __builtin_ctzll(x);
            if (pos < (unsigned long long)n) atomicMin(S.key, (pos << 32) | (unsigned)r);
        }
        __syncthreads();
        const unsigned long long k = *S.key;
        if (k == ~0ull) break;                                                // no further pivot in any column
        const int pcol = (int)(k >> 32), pr = (int)(k & 0xffffffffu);
        __syncthreads();
        if (pr != row) {                                                      // kernels.py:79-82
            for (int w = tid; w < nwords; w += T) {
                const uint64_t t = A[(size_t)row * nwords + w];
                A[(size_t)row * nwords + w] = A[(size_t)pr * nwords + w];
                A[(size_t)pr * nwords + w] = t;
            }
            if (tid == 0) {
                const uint8_t t = b[row]; b[row] = b[pr]; b[pr] = t;
                S.dead[pr] = S.dead[row]; S.dead[row] = 0;
            }
        }
        if (tid == 0) { pivot_rows[np] = (PIdx)row; pivot_cols[np] = (PIdx)pcol; *S.cnt = 0; }   // kernels.py:84-86
        __syncthreads();
        const int w0 = pcol >> 6;
        const uint64_t bit = 1ull << (pcol & 63);
        for (int w = tid; w < nwords; w += T) S.prow[w] = A[(size_t)row * nwords + w];
        for (int r = tid; r < m; r += T)                                       // kernels.py:88-89
            if (r != row && (A[(size_t)r * nwords + w0] & bit)) S.list[atomicAdd(S.cnt, 1)] = r;
        __syncthreads();
        const int cnt = *S.cnt, nact = nwords - w0;
        const uint8_t pb = b[row];
        for (int idx = tid; idx < cnt * nact; idx += T) {                      // kernels.py:90-91
            const int r = S.list[idx / nact], w = w0 + idx % nact;
            A[(size_t)r * nwords + w] ^= S.prow[w];
        }
        for (int idx = tid; idx < cnt; idx += T) b[S.list[idx]] ^= pb;         // kernels.py:92
        __syncthreads();
        np++; row++; col = pcol + 1;                                           // kernels.py:94
    }
    return np;
}

__global__ __launch_bounds__(1024) void gf2_eliminate_packed_kernel(int m, int n, int nwords, uint64_t *A, uint8_t *b, int64_t *prow,
                                                                    int64_t *pcol, int32_t *npiv) {
    extern __shared__ unsigned char lds_raw[];
    const int64_t B = blockIdx.x;
    const int maxp = m < n ? m : n;
    ElimShared S = carve_elim(lds_raw, m, nwords);
    const int np = eliminate_packed<int64_t>(A + (size_t)B * m * nwords, b + (size_t)B * m, m, n, nwords, prow + (size_t)B * maxp,
                                             pcol + (size_t)B * maxp, S);
    if (threadIdx.x == 0) npiv[B] = np;
}

// a8 packing layout (_pack_rows_uint64, kernels.py:36-46): bit c of a row lives in word c>>6, bit c&63
__global__ void pack_rows_kernel(int64_t rows, int n, int nwords, const uint8_t *__restrict__ A, uint64_t *__restrict__ P) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * nwords) return;
    const int64_t r = t / nwords;
    const int w = (int)(t - r * nwords);
    uint64_t x = 0;
    for (int c = 0; c < 64 && w * 64 + c < n; c++)
        if (A[r * n + w * 64 + c] & 1) x |= 1ull << c;
    P[t] = x;
}
__global__ void unpack_rows_kernel(int64_t rows, int n, int nwords, const uint64_t *__restrict__ P, uint8_t *__restrict__ A) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * n) return;
    const int64_t r = t / n;
    const int c = (int)(t - r * n);
    A[t] = (uint8_t)((P[r * nwords + (c >> 6)] >> (c & 63)) & 1);
}

// ------------------------------------------------------------------------------------------ OSD-0
struct OsdArgs {
    int m, n, nwords;
    const int32_t *indptr, *indices;
    const int32_t *list, *count;       // shots to process
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;   // indexed by shot
    int8_t *solution;
    // per-workgroup slabs
    uint64_t *A; uint8_t *b; int32_t *ord, *inv, *prow, *pcol; double *keys;
};

__global__ __launch_bounds__(1024) void osd0_kernel(OsdArgs P) {
    extern __shared__ unsigned char lds_raw[];
    const int m = P.m, n = P.n, nwords = P.nwords, tid = threadIdx.x, T = blockDim.x;
    const int maxp = m < n ? m : n;
    ElimShared S = carve_elim(lds_raw, m, nwords);
    uint64_t *A = P.A + (size_t)blockIdx.x * m * nwords;
    uint8_t *b = P.b + (size_t)blockIdx.x * m;
    int32_t *ord = P.ord + (size_t)blockIdx.x * n, *inv = P.inv + (size_t)blockIdx.x * n;
    int32_t *prow = P.prow + (size_t)blockIdx.x * maxp, *pcol = P.pcol + (size_t)blockIdx.x * maxp;
    double *keys = P.keys + (size_t)blockIdx.x * n;
    const int total = *P.count;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        // (1) column order: ascending |llr| (osd.py:11-12); ties by ascending index unless an explicit order is given
        if (P.ordering) {
            for (int c = tid; c < n; c += T) ord[c] = P.ordering[shot * n + c];
        } else {
            for (int j = tid; j < n; j += T) { const double a = fabs(llr[j]); keys[j] = (a != a) ? INFINITY : a; }
            __syncthreads();
            for (int j = tid; j < n; j += T) {
                const double kj = keys[j];
                int rank = 0;
                for (int i = 0; i < n; i++) { const double ki = keys[i]; rank += (ki < kj || (ki == kj && i < j)) ? 1 : 0; }
                ord[rank] = j;
            }
        }
        __syncthreads();
        for (int c = tid; c < n; c += T) inv[ord[c]] = c;
        for (int64_t t = tid; t < (int64_t)m * nwords; t += T) A[t] = 0;
        __syncthreads();
        // (2) H[:, ordering] packed + residual syndrome (osd.py:8-9,13); one thread per row
        for (int i = tid; i < m; i += T) {
            int s = 0;
            for (int e = P.indptr[i]; e < P.indptr[i + 1]; e++) {
                const int j = P.indices[e];
                s ^= hard[j];
                const int c = inv[j];
                A[(size_t)i * nwords + (c >> 6)] |= 1ull << (c & 63);
            }
            b[i] = (uint8_t)((synd[i] + s) & 1);
        }
        __syncthreads();
        // (3) eliminate (osd.py:15-17)
        const int np = eliminate_packed<int32_t>(A, b, m, n, nwords, prow, pcol, S);
        __syncthreads();
        // (4) back-fill (osd.py:19-25): e_permuted[c] = s_reduced[r] at pivots; solution = (hard + e_correction) % 2
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < np; t += T) {
            const int j = ord[pcol[t]];
            sol[j] = (int8_t)((hard[j] + b[prow[t]]) & 1);
        }
        __syncthreads();
    }
}

int osd0_lds_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled, OsdJudge *judge);

static int osd0_global_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                              const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream);

// callers hold g->mu; the graph's device workspaces are handed over in stream order (common.h)
int osd0_listed_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                       const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, OsdJudge *judge) {
    if (g->m == 0 || g->n == 0) return QLDPC_OK;
    int rc = g->ws_acquire(stream);                 // (a no-op for a caller that already holds the workspaces on this stream)
    if (rc != QLDPC_OK) return rc;
    bool handled = false;           // LDS-resident kernels for m <= 4096; the global-memory kernel is the general fallback
    rc = osd0_lds_launch(g, d_list, d_count, max_listed, d_synd, d_llr, d_hard, d_ordering, d_solution, flags, stream, handled, judge);
    if (rc == QLDPC_OK && !handled) rc = osd0_global_launch(g, d_list, d_count, d_synd, d_llr, d_hard, d_ordering, d_solution, stream);
    const int rel = g->ws_release(stream);          // always: a failing call may have enqueued launches the next stream has to wait for
    return rc != QLDPC_OK ? rc : rel;
}

static int osd0_global_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                              const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream) {
    const int m = g->m, n = g->n;
    const int nwords = ((n + 7) / 8 + 7) / 8;
    const int maxp = m < n ? m : n;
    const size_t slab = (size_t)m * nwords * 8 + (size_t)m + (size_t)n * 8 + (size_t)maxp * 8 + (size_t)n * 8 + 64;
    int grid = 512;
    while (grid > 1 && (size_t)grid * slab > ((size_t)2 << 30)) grid /= 2;
    int rc;
    // layout inside ws_misc: A | keys | ord | inv | prow | pcol | b  (8-byte aligned pieces first)
    const size_t szA = (size_t)grid * m * nwords * 8, szK = (size_t)grid * n * 8, szO = round_up((size_t)grid * n * 4, 8),
                 szP = round_up((size_t)grid * maxp * 4, 8), szB = round_up((size_t)grid * m, 8);
    if ((rc = g->ws_misc.ensure(szA + szK + 2 * szO + 2 * szP + szB)) != QLDPC_OK) return rc;
    unsigned char *base = g->ws_misc.as<unsigned char>();
    OsdArgs P;
    P.m = m; P.n = n; P.nwords = nwords; P.indptr = g->d_indptr; P.indices = g->d_indices;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.A = reinterpret_cast<uint64_t *>(base);
    P.keys = reinterpret_cast<double *>(base + szA);
    P.ord = reinterpret_cast<int32_t *>(base + szA + szK);
    P.inv = reinterpret_cast<int32_t *>(base + szA + szK + szO);
    P.prow = reinterpret_cast<int32_t *>(base + szA + szK + 2 * szO);
    P.pcol = reinterpret_cast<int32_t *>(base + szA + szK + 2 * szO + szP);
    P.b = reinterpret_cast<uint8_t *>(base + szA + szK + 2 * szO + 2 * szP);
    const size_t lds = elim_lds_bytes(m, nwords);
    if (lds > 150 * 1024) { set_error("OSD-0: matrix too large for the LDS scratch (m=%d nwords=%d)", m, nwords); return QLDPC_ERR_UNSUPPORTED; }
    const int block = (m >= 512 || n >= 2048) ? 1024 : 256;
    hipLaunchKernelGGL(osd0_kernel, dim3(grid), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

// ------------------------------------------------------------------------------------------ OSD-w (f1)
// performOSD_enhanced with order > 0 (reference src/decoding/osd.py:5-77).  The reference enters the combination sweep only when
// the OSD-0 solution misses the syndrome (osd.py:27-29), i.e. for syndromes outside the column space of H -- never for the
// syndromes the Monte-Carlo engine produces -- so this is a completeness path, written for clarity: one workgroup per shot, the
// literal elimination above, then one THREAD per candidate flip set (<= C(order+10, <= order) of them).
constexpr int kOsdwMaxOrder = 10;
constexpr int kOsdwMaxTest = kOsdwMaxOrder + 10;
constexpr int kOsdwMaxCand = 1 << 16;

struct OsdwArgs {
    OsdArgs base;
    int order;
    int64_t maxc;                 // <= 0: unlimited
    uint8_t *isp, *eperm;         // [grid][n]
    int8_t *efull;                // [grid][block][n]  per-thread candidate in permuted positions
    double *cmetric;              // [grid][kOsdwMaxCand]
    int32_t *cweight;             // [grid][kOsdwMaxCand]
};

__device__ inline unsigned osdw_binom(const unsigned (*bin)[kOsdwMaxOrder + 2], int a, int b) { return (b < 0 || a < b) ? 0u : bin[a][b]; }

// candidate -> flips (w positions into the test list), itertools.combinations order within one weight
__device__ inline void osdw_unrank(const unsigned (*bin)[kOsdwMaxOrder + 2], int K, int w, unsigned r, int *comb) {
    int x = 0;
    for (int t = 0; t < w; t++) {
        for (;;) {
            const unsigned c = osdw_binom(bin, K - 1 - x, w - 1 - t);
            if (c > r) break;
            r -= c; x++;
        }
        comb[t] = x++;
    }
}

// recompute_solution (kernels.py:205-219) on the ORIGINAL permuted matrix, then syndrome weight and metric (kernels.py:195-203)
__device__ inline void osdw_evaluate(const OsdArgs &P, int np, const int32_t *ord, const int32_t *inv, const int32_t *prow, const int32_t *pcol,
                                     const uint8_t *b, const int8_t *hard, const int8_t *synd, const double *llr, int8_t *e, int &weight,
                                     double &metric) {
    for (int t = 0; t < np; t++) {
        const int r = prow[t], c = pcol[t];
        int acc = 0;
        for (int k = P.indptr[r]; k < P.indptr[r + 1]; k++) { const int col = inv[P.indices[k]]; if (col != c) acc ^= e[col]; }
        e[c] = (int8_t)(b[r] ^ acc);
    }
    int wt = 0;
    for (int i = 0; i < P.m; i++) {
        int sp = 0;
        for (int k = P.indptr[i]; k < P.indptr[i + 1]; k++) { const int j = P.indices[k]; sp ^= (hard[j] ^ e[inv[j]]) & 1; }
        wt += (sp != (synd[i] & 1));
    }
    double mt = (wt > 0) ? 1e10 + wt * 1e8 : 0.0;
    for (int j = 0; j < P.n; j++) mt += (double)((hard[j] ^ e[inv[j]]) & 1) * fabs(llr[j]);
    weight = wt; metric = mt;
}

__global__ __launch_bounds__(256) void osdw_kernel(OsdwArgs W) {
    extern __shared__ unsigned char lds_raw[];
    __shared__ unsigned s_bin[kOsdwMaxTest + 1][kOsdwMaxOrder + 2];
    __shared__ int s_tp[kOsdwMaxTest];
    __shared__ double s_tk[kOsdwMaxTest];
    __shared__ int s_w0, s_K, s_best;
    const OsdArgs &P = W.base;
    const int m = P.m, n = P.n, nwords = P.nwords, tid = threadIdx.x, T = blockDim.x;
    const int maxp = m < n ? m : n;
    ElimShared S = carve_elim(lds_raw, m, nwords);
    uint64_t *A = P.A + (size_t)blockIdx.x * m * nwords;
    uint8_t *b = P.b + (size_t)blockIdx.x * m;
    int32_t *ord = P.ord + (size_t)blockIdx.x * n, *inv = P.inv + (size_t)blockIdx.x * n;
    int32_t *prow = P.prow + (size_t)blockIdx.x * maxp, *pcol = P.pcol + (size_t)blockIdx.x * maxp;
    double *keys = P.keys + (size_t)blockIdx.x * n;
    uint8_t *isp = W.isp + (size_t)blockIdx.x * n, *eperm = W.eperm + (size_t)blockIdx.x * n;
    int8_t *mine = W.efull + ((size_t)blockIdx.x * T + tid) * n;
    double *cmetric = W.cmetric + (size_t)blockIdx.x * kOsdwMaxCand;
    int32_t *cweight = W.cweight + (size_t)blockIdx.x * kOsdwMaxCand;
    if (tid == 0)
        for (int a = 0; a <= kOsdwMaxTest; a++)
            for (int c = 0; c <= kOsdwMaxOrder + 1; c++) s_bin[a][c] = (c == 0) ? 1u : (a == 0 ? 0u : s_bin[a - 1][c - 1] + (c <= a - 1 ? s_bin[a - 1][c] : 0u));
    const int total = *P.count;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        if (P.ordering) {
            for (int c = tid; c < n; c += T) ord[c] = P.ordering[shot * n + c];
        } else {
            for (int j = tid; j < n; j += T) { const double a = fabs(llr[j]); keys[j] = (a != a) ? INFINITY : a; }
            __syncthreads();
            for (int j = tid; j < n; j += T) {
                const double kj = keys[j];
                int rank = 0;
                for (int i = 0; i < n; i++) { const double ki = keys[i]; rank += (ki < kj || (ki == kj && i < j)) ? 1 : 0; }
                ord[rank] = j;
            }
        }
        __syncthreads();
        for (int c = tid; c < n; c += T) { inv[ord[c]] = c; isp[c] = 0; eperm[c] = 0; }
        for (int64_t t = tid; t < (int64_t)m * nwords; t += T) A[t] = 0;
        if (tid == 0) { s_w0 = 0; s_best = -1; }
        __syncthreads();
        for (int i = tid; i < m; i += T) {
            int sp = 0;
            for (int e = P.indptr[i]; e < P.indptr[i + 1]; e++) {
                const int j = P.indices[e];
                sp ^= hard[j];
                const int c = inv[j];
                A[(size_t)i * nwords + (c >> 6)] |= 1ull << (c & 63);
            }
            b[i] = (uint8_t)((synd[i] + sp) & 1);
        }
        __syncthreads();
        const int np = eliminate_packed<int32_t>(A, b, m, n, nwords, prow, pcol, S);      // osd.py:15-17; b = s_reduced from here on
        __syncthreads();
        for (int t = tid; t < np; t += T) { eperm[pcol[t]] = b[prow[t]]; isp[pcol[t]] = 1; }   // osd.py:19-21
        __syncthreads();
        for (int j = tid; j < n; j += T) sol[j] = (int8_t)((hard[j] + eperm[inv[j]]) & 1);     // osd.py:23-25
        __syncthreads();
        for (int i = tid; i < m; i += T) {                                                 // osd.py:27
            int sp = 0;
            for (int e = P.indptr[i]; e < P.indptr[i + 1]; e++) sp ^= sol[P.indices[e]] & 1;
            if (sp != (synd[i] & 1)) atomicAdd(&s_w0, 1);
        }
        __syncthreads();
        const int w0 = s_w0;
        if (w0 == 0 || W.order == 0) { __syncthreads(); continue; }                        // osd.py:28-29
        // test positions: the order+10 least reliable non-pivot positions, stable in (|llr|, position) (osd.py:31-41)
        if (tid == 0) {
            const int want = W.order + 10;
            int K = 0;
            for (int c = 0; c < n; c++) {
                if (isp[c]) continue;
                double a = fabs(llr[ord[c]]);
                if (a != a) a = INFINITY;
                if (K == want && !(a < s_tk[K - 1])) continue;
                int at = (K < want) ? K : K - 1;
                while (at > 0 && a < s_tk[at - 1]) { s_tk[at] = s_tk[at - 1]; s_tp[at] = s_tp[at - 1]; at--; }
                s_tk[at] = a; s_tp[at] = c;
                if (K < want) K++;
            }
            s_K = K;
        }
        __syncthreads();
        const int K = s_K;
        if (K == 0) { __syncthreads(); continue; }                                         // osd.py:33-34
        const int wmax = W.order < K ? W.order : K;
        long long ncand = 0;
        for (int w = 1; w <= wmax; w++) ncand += s_bin[K][w];
        if (W.maxc > 0 && ncand > W.maxc) ncand = W.maxc;                                   // osd.py:49,51
        if (ncand > kOsdwMaxCand) ncand = kOsdwMaxCand;                                     // host refuses configurations that could exceed this
        for (int idx = tid; idx < (int)ncand; idx += T) {
            unsigned r = (unsigned)idx;
            int w = 1;
            while (r >= s_bin[K][w]) { r -= s_bin[K][w]; w++; }
            int comb[kOsdwMaxOrder];
            osdw_unrank(s_bin, K, w, r, comb);
            for (int c = 0; c < n; c++) mine[c] = (int8_t)eperm[c];                         // osd.py:53
            for (int t = 0; t < w; t++) mine[s_tp[comb[t]]] ^= 1;                           // osd.py:54
            int wt; double mt;
            osdw_evaluate(P, np, ord, inv, prow, pcol, b, hard, synd, llr, mine, wt, mt);
            cweight[idx] = wt; cmetric[idx] = mt;
        }
        __syncthreads();
        if (tid == 0) {                                                                     // osd.py:43-46, 63-75 in enumeration order
            double best_metric = 1e10 + w0 * 1e8;
            for (int j = 0; j < n; j++) best_metric += (double)(sol[j] & 1) * fabs(llr[j]);
            bool found_valid = false;
            int best = -1;
            for (int idx = 0; idx < (int)ncand; idx++) {
                const double mt = cmetric[idx];
                if (cweight[idx] == 0) {
                    if (!found_valid || mt < best_metric) { best = idx; best_metric = mt; found_valid = true; }
                } else if (!found_valid && mt < best_metric) { best = idx; best_metric = mt; }
            }
            s_best = best;
            if (best >= 0) {                                                                // rebuild the winner in this thread's slab
                unsigned r = (unsigned)best;
                int w = 1;
                while (r >= s_bin[K][w]) { r -= s_bin[K][w]; w++; }
                int comb[kOsdwMaxOrder];
                osdw_unrank(s_bin, K, w, r, comb);
                for (int c = 0; c < n; c++) mine[c] = (int8_t)eperm[c];
                for (int t = 0; t < w; t++) mine[s_tp[comb[t]]] ^= 1;
                int wt; double mt;
                osdw_evaluate(P, np, ord, inv, prow, pcol, b, hard, synd, llr, mine, wt, mt);
            }
        }
        __syncthreads();
        if (s_best >= 0) {
            const int8_t *win = W.efull + (size_t)blockIdx.x * T * n;                      // thread 0's slab
            for (int j = tid; j < n; j += T) sol[j] = (int8_t)((hard[j] + win[inv[j]]) & 1);
        }
        __syncthreads();
    }
}

__global__ void iota_list_kernel(int64_t B, int32_t *list, int32_t *count) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) list[t] = (int32_t)t;
    if (t == 0) *count = (int32_t)B;
}

}  // namespace qldpc

using namespace qldpc;

QLDPC_EXPORT int qldpc_gf2_eliminate_packed(int64_t B, int m, int n, int nwords, uint64_t *A, uint8_t *b, int64_t *pivot_rows,
                                            int64_t *pivot_cols, int32_t *num_pivots) {
    QLDPC_REQUIRE(B >= 0 && m >= 0 && n >= 0, "negative size");
    QLDPC_REQUIRE(nwords * 64 >= n, "nwords=%d too small for n=%d", nwords, n);
    QLDPC_USE_DEVICE(0);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(num_pivots != nullptr, "num_pivots is NULL");
    if (m == 0 || n == 0) { for (int64_t i = 0; i < B; i++) num_pivots[i] = 0; return QLDPC_OK; }
    QLDPC_REQUIRE(A && b && pivot_rows && pivot_cols, "NULL buffer");
    const int maxp = m < n ? m : n;
    const size_t lds = elim_lds_bytes(m, nwords);
    QLDPC_REQUIRE(lds <= 150 * 1024, "matrix too large for the LDS scratch");
    DevTmp dA, db, dpr, dpc, dn;
    if ((rc = dA.alloc((size_t)B * m * nwords * 8)) || (rc = db.alloc((size_t)B * m)) || (rc = dpr.alloc((size_t)B * maxp * 8)) ||
        (rc = dpc.alloc((size_t)B * maxp * 8)) || (rc = dn.alloc((size_t)B * 4)))
        return rc;
    QLDPC_HIP_TRY(hipMemcpy(dA.p, A, (size_t)B * m * nwords * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(db.p, b, (size_t)B * m, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(zero_now(dpr.p, (size_t)B * maxp * 8));
    QLDPC_HIP_TRY(zero_now(dpc.p, (size_t)B * maxp * 8));
    hipLaunchKernelGGL(gf2_eliminate_packed_kernel, dim3((unsigned)B), dim3(m >= 512 ? 1024 : 256), lds, nullptr, m, n, nwords,
                       dA.as<uint64_t>(), db.as<uint8_t>(), dpr.as<int64_t>(), dpc.as<int64_t>(), dn.as<int32_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(A, dA.p, (size_t)B * m * nwords * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(b, db.p, (size_t)B * m, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(pivot_rows, dpr.p, (size_t)B * maxp * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(pivot_cols, dpc.p, (size_t)B * maxp * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(num_pivots, dn.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

// a7: the byte-matrix elimination (kernels.py:5-34) is the same algorithm on one element per byte; it is run as
// pack -> packed elimination -> unpack on the device, which yields the identical reduced matrix, rhs and pivots.
QLDPC_EXPORT int qldpc_gf2_eliminate(int64_t B, int m, int n, uint8_t *A, uint8_t *b, int64_t *pivot_rows, int64_t *pivot_cols,
                                     int32_t *num_pivots) {
    QLDPC_REQUIRE(B >= 0 && m >= 0 && n >= 0, "negative size");
    QLDPC_USE_DEVICE(0);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(num_pivots != nullptr, "num_pivots is NULL");
    if (m == 0 || n == 0) { for (int64_t i = 0; i < B; i++) num_pivots[i] = 0; return QLDPC_OK; }
    QLDPC_REQUIRE(A && b && pivot_rows && pivot_cols, "NULL buffer");
    const int nwords = ((n + 7) / 8 + 7) / 8, maxp = m < n ? m : n;
    const size_t lds = elim_lds_bytes(m, nwords);
    QLDPC_REQUIRE(lds <= 150 * 1024, "matrix too large for the LDS scratch");
    DevTmp dA8, dA, db, dpr, dpc, dn;
    const int64_t rows = B * m;
    if ((rc = dA8.alloc((size_t)rows * n)) || (rc = dA.alloc((size_t)rows * nwords * 8)) || (rc = db.alloc((size_t)rows)) ||
        (rc = dpr.alloc((size_t)B * maxp * 8)) || (rc = dpc.alloc((size_t)B * maxp * 8)) || (rc = dn.alloc((size_t)B * 4)))
        return rc;
    QLDPC_HIP_TRY(hipMemcpy(dA8.p, A, (size_t)rows * n, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(db.p, b, (size_t)rows, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(zero_now(dpr.p, (size_t)B * maxp * 8));
    QLDPC_HIP_TRY(zero_now(dpc.p, (size_t)B * maxp * 8));
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((rows * nwords + 255) / 256)), dim3(256), 0, nullptr, rows, n, nwords,
                       dA8.as<uint8_t>(), dA.as<uint64_t>());
    hipLaunchKernelGGL(gf2_eliminate_packed_kernel, dim3((unsigned)B), dim3(m >= 512 ? 1024 : 256), lds, nullptr, m, n, nwords,
                       dA.as<uint64_t>(), db.as<uint8_t>(), dpr.as<int64_t>(), dpc.as<int64_t>(), dn.as<int32_t>());
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((rows * n + 255) / 256)), dim3(256), 0, nullptr, rows, n, nwords,
                       dA.as<uint64_t>(), dA8.as<uint8_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(A, dA8.p, (size_t)rows * n, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(b, db.p, (size_t)rows, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(pivot_rows, dpr.p, (size_t)B * maxp * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(pivot_cols, dpc.p, (size_t)B * maxp * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(num_pivots, dn.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_osd0_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *llr, const int8_t *hard,
                                  const int32_t *ordering, int flags, int8_t *solution) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0 && B < ((int64_t)1 << 31), "batch out of range");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || g->n == 0) return QLDPC_OK;
    QLDPC_REQUIRE(llr && hard && solution && (syndromes || g->m == 0), "NULL buffer");
    const size_t m = g->m, n = g->n;
    DevTmp ds, dl, dh, dord, dsol, dlist, dcnt;
    if ((rc = ds.alloc(B * m)) || (rc = dl.alloc(B * n * 8)) || (rc = dh.alloc(B * n)) || (rc = dsol.alloc(B * n)) ||
        (rc = dlist.alloc(B * 4)) || (rc = dcnt.alloc(16)))
        return rc;
    if (ordering && (rc = dord.alloc(B * n * 4))) return rc;
    if (m) QLDPC_HIP_TRY(hipMemcpy(ds.p, syndromes, B * m, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dl.p, llr, B * n * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dh.p, hard, B * n, hipMemcpyHostToDevice));
    if (ordering) QLDPC_HIP_TRY(hipMemcpy(dord.p, ordering, B * n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(iota_list_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, nullptr, B, dlist.as<int32_t>(), dcnt.as<int32_t>());
    {
        std::lock_guard<std::mutex> lk(g->mu);
        rc = osd0_listed_launch(g, dlist.as<int32_t>(), dcnt.as<int32_t>(), B, ds.as<int8_t>(), dl.as<double>(), dh.as<int8_t>(),
                                ordering ? dord.as<int32_t>() : nullptr, dsol.as<int8_t>(), flags & QLDPC_FLAG_PUBLIC_MASK, nullptr);
        if (rc == QLDPC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("OSD-0 kernel failed: %s", hipGetErrorString(hipGetLastError())); rc = QLDPC_ERR_HIP; }
    }
    if (rc != QLDPC_OK) return rc;
    QLDPC_HIP_TRY(hipMemcpy(solution, dsol.p, B * n, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

// device-pointer form of qldpc_osd0_batch: only enqueues on `stream`.  d_select (may be NULL = every shot) lists the shots to solve, e.g. the
// ones a decode left unconverged; *d_select_count is read on the device, so no host round trip is needed between decode and OSD-0.
QLDPC_EXPORT int qldpc_osd0_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_syndromes, const double *d_llr, const int8_t *d_hard,
                                      const int32_t *d_ordering, const int32_t *d_select, const int32_t *d_select_count, int flags,
                                      int8_t *d_solution, void *stream) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0 && B < ((int64_t)1 << 31), "batch out of range");
    QLDPC_REQUIRE((d_select == nullptr) == (d_select_count == nullptr), "d_select and d_select_count go together");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || g->n == 0) return QLDPC_OK;
    QLDPC_REQUIRE(d_llr && d_hard && d_solution && (d_syndromes || g->m == 0), "NULL buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(g->mu);
    if (!d_select) {
        // the iota list lives in a shared workspace: order this stream behind its previous user BEFORE touching it (an OSD kernel still in flight on
        // another stream reads count and list from the same buffer)
        if ((rc = g->ws_acquire(s)) != QLDPC_OK) return rc;
        if ((rc = g->ws_list.ensure((size_t)B * 4 + 16)) != QLDPC_OK) { (void)g->ws_release(s); return rc; }
        int32_t *cnt = g->ws_list.as<int32_t>(), *list = cnt + 4;
        hipLaunchKernelGGL(iota_list_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, list, cnt);
        d_select = list; d_select_count = cnt;
    }
    return osd0_listed_launch(g, d_select, d_select_count, B, d_syndromes, d_llr, d_hard, d_ordering, d_solution, flags & QLDPC_FLAG_PUBLIC_MASK, s);
}

// f1: batched performOSD_enhanced(order, max_combinations) (osd.py:5-77); order == 0 is qldpc_osd0_batch.
QLDPC_EXPORT int qldpc_osdw_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *llr, const int8_t *hard,
                                  const int32_t *ordering, int order, int64_t max_combinations, int8_t *solution) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(order >= 0, "order must be >= 0");
    if (order == 0) return qldpc_osd0_batch(g, B, syndromes, llr, hard, ordering, 0, solution);
    QLDPC_REQUIRE(B >= 0 && B < ((int64_t)1 << 31), "batch out of range");
    QLDPC_REQUIRE(order <= kOsdwMaxOrder, "OSD order %d above the supported maximum %d", order, kOsdwMaxOrder);
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || g->n == 0) return QLDPC_OK;
    QLDPC_REQUIRE(llr && hard && solution && (syndromes || g->m == 0), "NULL buffer");
    const int m = g->m, n = g->n;
    if (m == 0) { std::memcpy(solution, hard, (size_t)B * n); return QLDPC_OK; }
    {   // worst-case number of candidates per shot: sum_{w<=order} C(min(n, order+10), w), cut by max_combinations
        const int K = std::min(n, order + 10);
        double worst = 0, c = 1;
        for (int w = 1; w <= std::min(order, K); w++) { c = c * (K - w + 1) / w; worst += c; }
        if (max_combinations > 0 && worst > (double)max_combinations) worst = (double)max_combinations;
        if (worst > kOsdwMaxCand) {
            set_error("OSD-%d would test up to %.0f flip sets per shot; pass max_combinations <= %d", order, worst, kOsdwMaxCand);
            return QLDPC_ERR_UNSUPPORTED;
        }
    }
    const int nwords = ((n + 7) / 8 + 7) / 8, maxp = m < n ? m : n, block = 256;
    const size_t lds = elim_lds_bytes(m, nwords);
    if (lds > 150 * 1024) { set_error("OSD-w: matrix too large for the LDS scratch (m=%d nwords=%d)", m, nwords); return QLDPC_ERR_UNSUPPORTED; }
    const size_t slab = (size_t)m * nwords * 8 + (size_t)m + (size_t)n * 18 + (size_t)maxp * 8 + (size_t)block * n + (size_t)kOsdwMaxCand * 12 + 256;
    int grid = (int)std::min<int64_t>(B, 64);
    while (grid > 1 && (size_t)grid * slab > ((size_t)2 << 30)) grid /= 2;
    DevTmp ds, dl, dh, dord, dsol, dlist, dcnt, dA, dkeys, do_, dinv, dpr, dpc, db, disp, dep, def, dcm, dcw;
    if ((rc = ds.alloc((size_t)B * m)) || (rc = dl.alloc((size_t)B * n * 8)) || (rc = dh.alloc((size_t)B * n)) || (rc = dsol.alloc((size_t)B * n)) ||
        (rc = dlist.alloc((size_t)B * 4)) || (rc = dcnt.alloc(16)) || (rc = dA.alloc((size_t)grid * m * nwords * 8)) ||
        (rc = dkeys.alloc((size_t)grid * n * 8)) || (rc = do_.alloc((size_t)grid * n * 4)) || (rc = dinv.alloc((size_t)grid * n * 4)) ||
        (rc = dpr.alloc((size_t)grid * maxp * 4)) || (rc = dpc.alloc((size_t)grid * maxp * 4)) || (rc = db.alloc((size_t)grid * m)) ||
        (rc = disp.alloc((size_t)grid * n)) || (rc = dep.alloc((size_t)grid * n)) || (rc = def.alloc((size_t)grid * block * n)) ||
        (rc = dcm.alloc((size_t)grid * kOsdwMaxCand * 8)) || (rc = dcw.alloc((size_t)grid * kOsdwMaxCand * 4)))
        return rc;
    if (ordering && (rc = dord.alloc((size_t)B * n * 4))) return rc;
    QLDPC_HIP_TRY(hipMemcpy(ds.p, syndromes, (size_t)B * m, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dl.p, llr, (size_t)B * n * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dh.p, hard, (size_t)B * n, hipMemcpyHostToDevice));
    if (ordering) QLDPC_HIP_TRY(hipMemcpy(dord.p, ordering, (size_t)B * n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(iota_list_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, nullptr, B, dlist.as<int32_t>(), dcnt.as<int32_t>());
    OsdwArgs W;
    W.base.m = m; W.base.n = n; W.base.nwords = nwords; W.base.indptr = g->d_indptr; W.base.indices = g->d_indices;
    W.base.list = dlist.as<int32_t>(); W.base.count = dcnt.as<int32_t>(); W.base.synd = ds.as<int8_t>(); W.base.llr = dl.as<double>();
    W.base.hard = dh.as<int8_t>(); W.base.ordering = ordering ? dord.as<int32_t>() : nullptr; W.base.solution = dsol.as<int8_t>();
    W.base.A = dA.as<uint64_t>(); W.base.b = db.as<uint8_t>(); W.base.ord = do_.as<int32_t>(); W.base.inv = dinv.as<int32_t>();
    W.base.prow = dpr.as<int32_t>(); W.base.pcol = dpc.as<int32_t>(); W.base.keys = dkeys.as<double>();
    W.order = order; W.maxc = max_combinations;
    W.isp = disp.as<uint8_t>(); W.eperm = dep.as<uint8_t>(); W.efull = def.as<int8_t>(); W.cmetric = dcm.as<double>(); W.cweight = dcw.as<int32_t>();
    hipLaunchKernelGGL(osdw_kernel, dim3(grid), dim3(block), lds, nullptr, W);
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(solution, dsol.p, (size_t)B * n, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

// =====================================================================================================================
// LDS-resident OSD-0 (a9) for matrices with m <= 1024 rows.
//
// The reference permutes the columns of H by reliability and runs a full Gauss-Jordan elimination on the dense
// m x n matrix (osd.py:11-17): 1.1 MB of row data per shot for the circuit-level matrices.  OSD-0 only needs the
// pivot columns and the reduced right-hand side, so this kernel never materialises the permuted matrix:
//   * it keeps the accumulated row transformation T (current rows = T * original rows) -- as U = T^T, m x m bits,
//     XOR-swizzled, 126 KB for m = 1008 -- in LDS;
//   * the next column in reliability order is a SPARSE column h of H (<= 6 ones); its current (reduced) form is
//     T h = XOR of the rows U[i], i in supp(h): 6 x 16 words;
//   * pivot = the candidate row at the smallest CURRENT position >= `row` (the reference's physical row order after its
//     swaps, kernels.py:71-82, is tracked by a position table instead of moving data);
//   * eliminating the other ones of the column is "U[q] ^= mask for every q with bit pivot set" and b ^= mask if b[pivot];
//   * the order (ascending |llr|, ties by index) comes from ONE bitonic sort of all columns done in the LDS that later
//     holds U; it is parked in global memory and streamed back in chunks of K columns;
//   * measured on the circuit-level matrices, full rank is only reached ~6,400 columns deep while only ~940 of them
//     pivot.  A column that is dependent on the pivots so far stays dependent, so whenever the sweep meets a dependent
//     column it tests ALL remaining columns of the chunk in parallel (one thread per column) and drops every dependent
//     one at once; the sweep stops as soon as rank(H) pivots exist.
// The result (pivot columns, reduced rhs at the pivots) is identical to the reference's; tests compare solutions.
// =====================================================================================================================
namespace qldpc {

struct OsdLdsArgs {
    int m, n, mw, rankH, K, cdeg, npad;
    const int32_t *indptr, *indices, *colptr, *rowidx;
    const int32_t *list, *count;
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    uint16_t *ordws;               // [grid][n] sorted column order of the shot in flight (global, L2-resident)
    unsigned long long *ug;        // UG kernels: [grid][(m + 2) * mw] row transform in HBM/L2, followed by [grid][n] sort keys
    unsigned long long *ugkeys;
    int *queue;                    // next list entry to process (zeroed before the launch): work is handed out one shot at a time
    unsigned long long *clk;       // QLDPC_FLAG_CLOCK_PROBE buffer of the launching plan, else NULL
    unsigned long long *dbg;       // optional counters: [0] shots, [1] chunks, [2] columns swept, [3] pivots, [4] cycles, [5] kill passes, [6] blocks
    int offIdx, offAlive, offRows, offPc, offR, offBlk, offMisc, offSort;
};

#ifndef QLDPC_OSD_BLOCK
#define QLDPC_OSD_BLOCK 16
#endif
constexpr int kOsdBlock = QLDPC_OSD_BLOCK;      // columns resolved per block (4 per register of the resolving wave)
static_assert(kOsdBlock <= 16, "one wave per column, at most 4 columns per wave with 256-thread blocks");

// Position-space formulation.  T (current rows = T * original rows, rows in their CURRENT physical order, i.e. after the
// reference's swaps kernels.py:79-82) is kept as U = T^T: U[q] bit p = T[p][q].  Rows 0..m-1 of U belong to the original
// rows, row m is all zero (padding target of short columns) and row m+1 carries the right-hand side b (it transforms like
// a column).  The reduced form of a sparse column h is XOR_{i in supp h} U[i]; its pivot is simply the first set bit at a
// position >= row (kernels.py:71-75); the swap row <-> pivot is a 2-bit swap in every row of U; the elimination
// (kernels.py:88-92) is U[q] ^= mask for every q whose bit `row` is set.
// UG = true: U (1 MB per shot for m = 2880) and the sort scratch live in global memory (one slab per workgroup); everything else is unchanged.
template <bool UG>
__global__ __launch_bounds__(1024) void osd0_lds_kernel(OsdLdsArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, cd = P.cdeg, tid = threadIdx.x, T = blockDim.x;
    unsigned long long *U;
    if (UG) U = P.ug + (size_t)blockIdx.x * (size_t)(m + 2) * mw; else U = reinterpret_cast<unsigned long long *>(lds);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);         // [K] columns of the current chunk
    uint8_t *alive = reinterpret_cast<uint8_t *>(lds + P.offAlive);        // [K]
    uint16_t *colrows = reinterpret_cast<uint16_t *>(lds + P.offRows);     // [K][cd] supports
    uint16_t *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);         // [m] pivot t sits at position t
    unsigned long long *R = reinterpret_cast<unsigned long long *>(lds + P.offR);       // [kOsdBlock][mw] reduced columns -> masks
    int *blk = reinterpret_cast<int *>(lds + P.offBlk);                    // [0] nb, [1] nops, [2] anydep, [3] next c; [4..] cols[32], opa[32], opp[32], opt[32]
    int *bcol = blk + 4, *opa = bcol + kOsdBlock, *opp = opa + kOsdBlock, *opt = opp + kOsdBlock;
    int2 *stp = reinterpret_cast<int2 *>(opt + kOsdBlock);      // (a or -1, pp) published by the owner of column t
    uint16_t *ordw = P.ordws + (size_t)blockIdx.x * n;
    const int brow = m + 1;                                                // U row that carries b
    // U[q][w]: LDS: row-major, swizzled (uswz); HBM/L2 (UG): WORD-major, w * (m + 2) + q -- a thread owns a row and walks its words, so the
    // lanes of a wave-load then touch neighbouring addresses instead of 64 different cache lines (the row-major form was address-rate bound)
    auto uix = [&](int q, int w) -> int { return UG ? w * (m + 2) + q : uswz(q, w, mw); };

    const int total = *P.count;
    const ClkStamp clk0 = clk_begin(P.clk);
    int *s_item = reinterpret_cast<int *>(stp + kOsdBlock);                  // the list entry this workgroup processes next
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
        // ---- column order: ascending |llr| (osd.py:11-12), ties by ascending index; bitonic sort of (key, index) in LDS ----
        if (!P.ordering) {
            unsigned long long *keys;                                                               // [n]   (LDS: aliases U)
            uint16_t *pa, *pb;                                                                      // [n] each
            unsigned *cnt;                                                                          // [256][waves] + [waves]
            if (UG) {
                keys = P.ugkeys + (size_t)blockIdx.x * (size_t)(n + (n + 1) / 2);
                pa = reinterpret_cast<uint16_t *>(keys + n); pb = pa + n;
                cnt = reinterpret_cast<unsigned *>(lds + P.offSort);
            } else {
                keys = reinterpret_cast<unsigned long long *>(lds);
                pa = reinterpret_cast<uint16_t *>(lds + (size_t)n * 8); pb = pa + n;
                cnt = reinterpret_cast<unsigned *>(lds + (((size_t)n * 12 + 15) & ~(size_t)15));
            }
            osd_radix_sort(llr, n, keys, pa, pb, cnt, ordw);
        }
        // ---- init: T = I (positions = original rows), b = s + H hard (osd.py:8-9) ----
        for (int t = tid; t < (m + 2) * mw; t += T) U[t] = 0ull;
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            U[uix(r, r >> 6)] = 1ull << (r & 63);
            int sy = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) sy ^= hard[P.indices[e]] & 1;
            if (sy) atomicOr(&U[uix(brow, r >> 6)], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0;
        unsigned long long d_cols = 0, d_chunks = 0, d_kills = 0, d_blocks = 0, c_p1 = 0, c_p2 = 0, c_p3 = 0, c_kill = 0, d_nzw = 0, d_wops = 0, d_lops = 0, c_p3own = 0;
        (void)d_wops; (void)d_lops; (void)c_p3own; (void)d_nzw;
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
                const int c = t / cd, d = t - c * cd, j = sidx[c];
                const int k = P.colptr[j] + d;
                colrows[t] = (k < P.colptr[j + 1]) ? (uint16_t)P.rowidx[k] : (uint16_t)m;          // row m of U is all zero
            }
            __syncthreads();
            // drops every still-alive column of the chunk from c0 on that is dependent on the pivots found so far (one thread per column)
            auto kill_pass = [&](int c0, int t0, int tstride) {               // threads t0 = 0 .. tstride - 1 take part
                const int wq = row >> 6;
                for (int c2 = c0 + t0; c2 < L; c2 += tstride) {
                    if (!alive[c2]) continue;
                    const uint16_t *cr2 = colrows + c2 * cd;
                    int rr[8];                                           // the column's support once (cd <= 8; short columns point at the zero row m)
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr2[d] : m;
                    unsigned long long any = 0ull;
                    for (int w = wq; w < mw; w++) {
                        unsigned long long xs[8];
#pragma unroll
                        for (int d = 0; d < 8; d++) xs[d] = U[uix(rr[d], w)];
                        unsigned long long x = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                        for (int d = 8; d < cd; d++) x ^= U[uix(cr2[d], w)];     // columns heavier than 8 (not the circuit-level matrices)
                        any |= (w == wq) ? (x & (~0ull << (row & 63))) : x;
                    }
                    if (!any) alive[c2] = 0;
                }
            };
            bool kill_due = false;                                           // a block met dependent columns: test the rest of the chunk (deferred, see phase 2)
            if (row > 0) {                                                  // a fresh chunk late in the sweep is mostly dependent columns: one pass up
                long long tk = OSD_CLOCK();                                    // front instead of one serial pivot step per dependent column
                d_kills++;
                kill_pass(0, tid, T);
                __syncthreads();
                c_kill += OSD_CLOCK() - tk;
            }
            // ================= blocks of up to kOsdBlock alive columns =================
            while (true) {
                if (tid < 64) {                                              // wave 0 collects the next alive columns of the chunk (ballot scan)
                    int c = blk[3], nbc = 0;
                    while (c < L && nbc < kOsdBlock) {
                        const int cc = c + tid;
                        const bool al = (cc < L) && alive[cc];
                        const unsigned long long bal = __ballot(al);
                        const int before = __builtin_popcountll(bal & ((1ull << tid) - 1ull));
                        if (al && nbc + before < kOsdBlock) bcol[nbc + before] = cc;
                        const int got = __builtin_popcountll(bal);
                        if (nbc + got >= kOsdBlock) {                         // stop right behind the column that filled the block
                            int need = kOsdBlock - nbc;
                            unsigned long long bb = bal;
                            int lastpos = 0;
                            while (need-- > 0) { lastpos = __builtin_ctzll(bb); bb &= bb - 1; }
                            c += lastpos + 1; nbc = kOsdBlock;
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
                // ---- phase 1: reduced columns R[t] = XOR of U rows ----
                for (int x = tid; x < nb * mw; x += T) {
                    const int t = x / mw, w = x - t * mw;
                    const uint16_t *cr = colrows + bcol[t] * cd;
                    int rr[8];
                    unsigned long long xs[8];
#pragma unroll
                    for (int d = 0; d < 8; d++) rr[d] = (d < cd) ? (int)cr[d] : m;           // short columns point at the zero row m
#pragma unroll
                    for (int d = 0; d < 8; d++) xs[d] = U[uix(rr[d], w)];
                    unsigned long long acc = ((xs[0] ^ xs[1]) ^ (xs[2] ^ xs[3])) ^ ((xs[4] ^ xs[5]) ^ (xs[6] ^ xs[7]));
                    for (int d = 8; d < cd; d++) acc ^= U[uix(cr[d], w)];
                    R[t * mw + w] = acc;
                }
                __syncthreads();
                c_p1 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 2: the block's pivots.  Waves 0-3 (one per SIMD) hold 4 columns each: lane = (grp, w) has word w of column
                // 4*wave + grp.  Column t's owner finds its pivot (first set bit at a position >= lrow, kernels.py:71-75), turns the
                // column into the elimination mask and publishes (a, pp); after ONE barrier the waves holding later columns apply that
                // swap + XOR (kernels.py:79-92 restricted to the block).  Measured alternatives: all 16 columns in a single wave
                // (1.7 M cycles per shot), one wave per column (1.85 M), four columns resolved inside a wave per barrier (1.75 M);
                // this form: 1.36 M -- the cost is the dependent ballot -> scalar -> lane-read chain of a step, not the barrier.
                int nops = 0, anydep = 0;
                if (!UG) {
                    // rows of <= 16 words: the whole block in wave 0, registers only (quad_pivot_step above); the other waves wait at the barrier
                    if (tid < 64) {
                        const int lane = tid, g = lane & 3, w = lane >> 2, wq = row >> 6;
                        QuadPivot S;
#pragma unroll
                        for (int i = 0; i < 4; i++) S.X[i] = (4 * i + g < nb && w < mw) ? R[(4 * i + g) * mw + w] : 0ull;
                        S.live = (w > wq) ? ~0ull : ((w == wq) ? (~0ull << (row & 63)) : 0ull);
                        S.lrow = row; S.nops = 0; S.depmask = 0u; S.oppv = 0; S.optv = 0; S.stop = false; S.nzw = 0u;
#define QLDPC_QSTEP(TT) if (TT < nb && !S.stop) quad_pivot_step<TT>(S, R, mw, lane, P.rankH, m);
                        QLDPC_QSTEP(0) QLDPC_QSTEP(1) QLDPC_QSTEP(2) QLDPC_QSTEP(3) QLDPC_QSTEP(4) QLDPC_QSTEP(5) QLDPC_QSTEP(6) QLDPC_QSTEP(7)
                        QLDPC_QSTEP(8) QLDPC_QSTEP(9) QLDPC_QSTEP(10) QLDPC_QSTEP(11) QLDPC_QSTEP(12) QLDPC_QSTEP(13) QLDPC_QSTEP(14) QLDPC_QSTEP(15)
#undef QLDPC_QSTEP
                        if (lane < S.nops) { opa[lane] = row + lane; opp[lane] = S.oppv; opt[lane] = S.optv; pvcol[row + lane] = sidx[bcol[S.optv]]; }
                        if (lane < nb && ((S.depmask >> lane) & 1u)) alive[bcol[lane]] = 0;
                        if (lane == 0) { blk[1] = S.nops; blk[2] = (S.depmask != 0u) ? 1 : 0; }
#ifdef QLDPC_OSD_TIMERS
                        d_nzw += S.nzw;
#endif
                    } else if (kill_due) {
                        // the other waves: the dependent-column tests an earlier block asked for, against the transform as it stands (without this
                        // block's pivots: a dependent column stays dependent, the test is only one block less eager); the block's own columns
                        // lie before blk[3], so the two do not touch the same alive[] entries
                        kill_pass(blk[3], tid - 64, T - 64);
                    }
                    if (kill_due) { d_kills++; kill_due = false; }
                    __syncthreads();
                    nops = blk[1]; anydep = blk[2];
                } else {
                    // lanes per column: 16 / 32 / 64 for rows of <= 16 / 32 / 64 words; columns per wave 4 / 2 / 1; holder waves 4 / 8 / 16
                    const int lsh = (mw <= 16) ? 4 : ((mw <= 32) ? 5 : 6), LPC = 1 << lsh, cpw = 64 >> lsh, osh = 6 - lsh;
                    const unsigned long long gmask = (LPC == 64) ? ~0ull : ((1ull << LPC) - 1ull);
                    const int wv = tid >> 6, lane = tid & 63, w = lane & (LPC - 1), grp = lane >> lsh, sc = cpw * wv + grp;
                    const bool holder = wv < (kOsdBlock >> osh);
                    unsigned long long X = (holder && sc < nb && w < mw) ? R[sc * mw + w] : 0ull;
                    const int colid = (holder && sc < nb) ? (int)sidx[bcol[sc]] : 0;
                    int lrow = row;
                    for (int t = 0; t < nb; t++) {
                        const int gt = t & (cpw - 1);
                        if (wv == (t >> osh)) {
                            const bool ing = (grp == gt);
                            const int wq = lrow >> 6;
                            const unsigned long long mword = (!ing || w < wq) ? 0ull : ((w == wq) ? (X & (~0ull << (lrow & 63))) : X);
                            const unsigned long long bal = (__ballot(mword != 0ull) >> (gt * LPC)) & gmask;
                            if (bal == 0ull) {                                                   // dependent on the pivots so far
                                if (lane == gt * LPC) { stp[t] = make_int2(-1, 0); alive[bcol[t]] = 0; }
                            } else {
                                const int pw = __builtin_amdgcn_readfirstlane(__builtin_ctzll(bal)), src = gt * LPC + pw;
                                const unsigned long long pword = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mword >> 32), src) << 32) |
                                                                 (unsigned)__builtin_amdgcn_readlane((int)mword, src);
                                const int pp = pw * 64 + __builtin_ctzll(pword), a = lrow, wa = a >> 6, wp = pp >> 6;
                                const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                                const bool olda = __ballot(ing && w == wa && (X & abit) != 0ull) != 0ull;
                                // swap bits a <-> pp of the pivot column itself (bit pp is 1), then clear bit a: that is the elimination mask
                                unsigned long long rm = X;
                                if (w == wp) rm = olda ? (rm | pbit) : (rm & ~pbit);
                                if (w == wa) rm &= ~abit;
                                if (ing) { X = rm; if (w < mw) R[t * mw + w] = rm; }
                                if (lane == gt * LPC) { stp[t] = make_int2(a, pp); opa[nops] = a; opp[nops] = pp; opt[nops] = t; pvcol[a] = (uint16_t)colid; }
                            }
                        }
                        __syncthreads();
                        const int2 st = stp[t];                                                  // both loads issue together: one LDS round trip
                        const unsigned long long rmw = (holder && w < mw) ? R[t * mw + w] : 0ull;
                        if (st.x < 0) { anydep = 1; continue; }
                        if (holder && cpw * wv + cpw - 1 > t) {                                  // wave-uniform: this wave still holds a later column
                            const int a = st.x, pp = st.y, wa = a >> 6, wp = pp >> 6;
                            const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                            unsigned long long x = X;
                            // bits a / pp of a column live in the lanes holding words wa / wp of its group: two ballots instead of shuffles
                            const unsigned long long balA = __ballot(w == wa && (x & abit) != 0ull), balP = __ballot(w == wp && (x & pbit) != 0ull);
                            const bool ba = (balA >> (grp * LPC + wa)) & 1ull, bp = (balP >> (grp * LPC + wp)) & 1ull;
                            if (ba != bp) { if (w == wa) x ^= abit; if (w == wp) x ^= pbit; }
                            if (bp) x ^= rmw;                                                    // after the swap, bit a of the column is bp
                            if (sc > t && sc < nb) X = x;
                        }
                        nops++; lrow++;
                        if (lrow >= P.rankH || lrow >= m) break;                                 // full rank: the remaining columns cannot pivot
                    }
                    __syncthreads();
                }
                c_p2 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                // ---- phase 3: apply the block's operations to every row of U (and to b) ----
                // rows of 16 words go to threads as q = 16 * lane + wave: the rows an operation changes cluster (the rows near the pivot
                // position change under every operation of the block), consecutive rows in one wave made that wave the critical path
                const bool strided = (mw == 16);                             // (mw == 16 implies 1024 threads)
                auto rowq = [&](int qb) { return strided ? qb + ((tid & 63) << 4) + (((tid >> 6) + tid) & 15) : qb + tid; };
                // the two halves of an operation (kernels.py:79-92) on row q: swap bits a <-> pp; add the elimination mask
                auto swap_bits = [&](int q, int a, int pp) {
                    const int wa = a >> 6, wp = pp >> 6;
                    const unsigned long long abit = 1ull << (a & 63), pbit = 1ull << (pp & 63);
                    if (wp == wa) { U[uix(q, wa)] ^= abit ^ pbit; }
                    else { const unsigned long long xa = U[uix(q, wa)], xp = U[uix(q, wp)]; U[uix(q, wa)] = xa ^ abit; U[uix(q, wp)] = xp ^ pbit; }
                };
                auto add_mask = [&](int q, const unsigned long long *mk) {
                    if (!UG && mw == 16) {                                   // all reads in flight before the first XOR (a rolled loop waits per word)
                        ulonglong2 u[8], k2[8];
                        ulonglong2 *Uq = reinterpret_cast<ulonglong2 *>(U + q * 16);
                        const ulonglong2 *mk2 = reinterpret_cast<const ulonglong2 *>(mk);
                        const int sz = (q >> 4) & 7;                         // the row's swizzle on pairs of words
#pragma unroll
                        for (int w = 0; w < 8; w++) u[w] = Uq[w ^ sz];
#pragma unroll
                        for (int w = 0; w < 8; w++) k2[w] = mk2[w];
#pragma unroll
                        for (int w = 0; w < 8; w++) { u[w].x ^= k2[w].x; u[w].y ^= k2[w].y; }
#pragma unroll
                        for (int w = 0; w < 8; w++) Uq[w ^ sz] = u[w];
                    } else {
                        for (int w0 = 0; w0 < mw; w0 += 16) {                // 16 words in flight (rows live in HBM/L2 in the UG kernel)
                            unsigned long long u[16];
#pragma unroll
                            for (int j2 = 0; j2 < 16; j2++) u[j2] = (w0 + j2 < mw) ? U[uix(q, w0 + j2)] : 0ull;
#pragma unroll
                            for (int j2 = 0; j2 < 16; j2++) if (w0 + j2 < mw) U[uix(q, w0 + j2)] = u[j2] ^ mk[w0 + j2];
                        }
                    }
                };
                if (nops > 0) {
                    // rows in LDS: osd_rows_apply (osd_common.h) -- tested bits of the whole block read at once, only touched operations visited;
                    // rows in HBM/L2 (UG): the same scheme written out below on the word-major transform
                    int ppv = opp[tid & 15], ptv = opt[tid & 15];           // operation k's (pp, column) sit in lane k of every 16
                    asm volatile("" : "+v"(ppv), "+v"(ptv));
                    const int ws = row >> 6, sh = row & 63;
                    const uint32_t valid = (1u << nops) - 1u;
                    const uint32_t *U32 = reinterpret_cast<const uint32_t *>(U);
                    // lanes 0-15 of every 32 stand for position a_j = row + j, lanes 16-31 for pp_j (j = lane % 16): see the bit updates below
                    const int j16 = tid & 15, mypos = (j16 < nops) ? ((tid & 16) ? ppv : row + j16) : 0;
                    for (int qb = 0; qb < m + 2; qb += T) {
                        const int q = rowq(qb);
                        const bool act = (q < m + 2) && (q != m);
                        const int qq = act ? q : m;                          // idle lanes look at the all-zero row
                        if (!UG) {                                           // rows in LDS: osd_common.h
                            osd_rows_apply(U + qq * mw, (mw == 16) ? ((qq >> 3) & 14) : 0, act, row, nops, mw, ppv, ptv, R, tid, d_wops, d_lops);
                            continue;
                        }
                        uint32_t ab, pb = 0u;
                        {
                            const unsigned long long A0 = U[uix(qq, ws)], A1 = (ws + 1 < mw) ? U[uix(qq, ws + 1)] : 0ull;
                            uint32_t Pw[kOsdBlock];
#pragma unroll
                            for (int k = 0; k < kOsdBlock; k++) {
                                const int pk = (k < nops) ? __builtin_amdgcn_readlane(ppv, k) : 0;
                                Pw[k] = U32[2 * uix(qq, pk >> 6) + ((pk >> 5) & 1)];
                            }
                            ab = (uint32_t)((A0 >> sh) | (sh ? (A1 << (64 - sh)) : 0ull)) & valid;             // positions row .. row + nops - 1
#pragma unroll
                            for (int k = 0; k < kOsdBlock; k++) {
                                const int pk = __builtin_amdgcn_readlane(ppv, k);
                                pb |= ((Pw[k] >> (pk & 31)) & 1u) << k;
                            }
                            pb &= valid;
                        }
                        int kdone = -1;
                        for (;;) {
                            uint32_t x = ab | pb;                            // OR over the wave
                            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
                            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);      // row_half_mirror
                            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);      // row_mirror
                            uint32_t wtb = (uint32_t)__builtin_amdgcn_readlane((int)x, 0) | (uint32_t)__builtin_amdgcn_readlane((int)x, 16) |
                                           (uint32_t)__builtin_amdgcn_readlane((int)x, 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
                            if (kdone >= 0) wtb &= ~1u << kdone;
                            if (wtb == 0u) break;
                            const int k = __builtin_ctz(wtb);
                            kdone = k;
                            const int ppk = __builtin_amdgcn_readlane(ppv, k);
                            const unsigned long long *mk = R + __builtin_amdgcn_readlane(ptv, k) * mw;
                            // what operation k does to the bits the later operations j > k test in a row it changes: the swap puts the row's old
                            // bit a_k at position pp_k (xx: bit j = (a_j == pp_k), bit 16 + j = (pp_j == pp_k)); the XOR flips them by the mask's
                            // bits at those positions (mm: bit j = mask_k[a_j], bit 16 + j = mask_k[pp_j])
                            const bool later = (j16 > k) && (j16 < nops);
                            const uint32_t mword = reinterpret_cast<const uint32_t *>(mk)[2 * (mypos >> 6) + ((mypos >> 5) & 1)];
                            const uint32_t xx = (uint32_t)__ballot(later && mypos == ppk);
                            const uint32_t mm = (uint32_t)__ballot(later && ((mword >> (mypos & 31)) & 1u));
                            const bool ba = (ab >> k) & 1u, bp = (pb >> k) & 1u;
#ifdef QLDPC_OSD_TIMERS
                            { d_wops++; d_lops += __builtin_popcountll(__ballot(ba || bp)); }
#endif
                            if (ba != bp) {
                                swap_bits(q, row + k, ppk);
                                const uint32_t sa = xx & 0xFFFFu, sp = xx >> 16;
                                ab = ba ? (ab | sa) : (ab & ~sa);
                                pb = ba ? (pb | sp) : (pb & ~sp);
                            }
                            if (bp) {                                        // bit a after the swap: add the pivot row (kernels.py:88-92)
                                add_mask(q, mk);
                                ab ^= mm & 0xFFFFu;
                                pb ^= mm >> 16;
                            }
                        }
                    }
                }
                row += nops;
#ifdef QLDPC_OSD_TIMERS
                c_p3own += OSD_CLOCK() - tp;
#endif
                __syncthreads();
                c_p3 += OSD_CLOCK() - tp; tp = OSD_CLOCK();
                if (row >= P.rankH || row >= m) { finished = true; break; }
                // ---- dependent columns were met: drop every column of the chunk that is dependent by now ----
                if (anydep) {
                    if (!UG) {
                        kill_due = true;                                     // done by the idle waves beside the next block's pivot chain (phase 2 above)
                    } else {
                        d_kills++;
                        kill_pass(blk[3], tid, T);
                        __syncthreads();
                        c_kill += OSD_CLOCK() - tp;
                    }
                }
            }
            __syncthreads();      // nobody may refill alive[]/sidx[] while others still use them
        }
        if (P.dbg && tid == 0) {
            atomicAdd(&P.dbg[0], 1ull); atomicAdd(&P.dbg[1], d_chunks); atomicAdd(&P.dbg[2], d_cols); atomicAdd(&P.dbg[3], (unsigned long long)row);
            atomicAdd(&P.dbg[4], (unsigned long long)(OSD_CLOCK() - t_start)); atomicAdd(&P.dbg[5], d_kills); atomicAdd(&P.dbg[6], d_blocks);
            atomicAdd(&P.dbg[8], (unsigned long long)(t_sorted - t_start)); atomicAdd(&P.dbg[9], c_p1); atomicAdd(&P.dbg[10], c_p2); atomicAdd(&P.dbg[11], c_p3);
            atomicAdd(&P.dbg[12], c_kill); atomicAdd(&P.dbg[7], d_nzw);
        }
#ifdef QLDPC_OSD_TIMERS
        if (P.dbg && (tid & 63) == 0) { atomicAdd(&P.dbg[14], d_wops); atomicAdd(&P.dbg[15], d_lops); atomicAdd(&P.dbg[13], c_p3own / (blockDim.x >> 6)); }
#endif
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < row; t += T) {
            const int j = pvcol[t];
            const int8_t bbit = (int8_t)((U[uix(brow, t >> 6)] >> (t & 63)) & 1ull);
            sol[j] = (int8_t)((hard[j] ^ bbit) & 1);
        }
        __syncthreads();
    }
    clk_end(P.clk, clk0);
}

// ELL view of the columns for the OSD kernels: [n][max(max_col_deg, 1)] rows of every column in ascending order, padded with m (the
// all-zero row of the transform).  Built once per graph on first use (callers hold g->mu).
int ensure_col_rows(const qldpc_graph *g) {
    if (g->d_col_rows) return QLDPC_OK;
    const int cdeg = std::max(g->max_col_deg, 1);
    std::vector<uint16_t> cr((size_t)std::max(g->n, 1) * cdeg, (uint16_t)g->m);
    for (int j = 0; j < g->n; j++)
        for (int k = g->colptr[j]; k < g->colptr[j + 1]; k++) cr[(size_t)j * cdeg + (k - g->colptr[j])] = (uint16_t)g->rowidx[k];
    QLDPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_col_rows), cr.size() * 2 + 16));
    QLDPC_HIP_TRY(hipMemcpy(g->d_col_rows, cr.data(), cr.size() * 2, hipMemcpyHostToDevice));
    return QLDPC_OK;
}

int osd0_small_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                      const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled, OsdJudge *judge);

// rank of H over GF(2) (host, once per graph): the sweep above can stop as soon as this many pivots exist
int host_gf2_rank(const qldpc_graph *g) {
    const int m = g->m, n = g->n, nw = (n + 63) / 64;
    std::vector<uint64_t> A((size_t)m * nw, 0);
    for (int i = 0; i < m; i++)
        for (int e = g->indptr[i]; e < g->indptr[i + 1]; e++) A[(size_t)i * nw + (g->indices[e] >> 6)] |= 1ull << (g->indices[e] & 63);
    int rank = 0;
    for (int c = 0; c < n && rank < m; c++) {
        const int w = c >> 6;
        const uint64_t bit = 1ull << (c & 63);
        int pr = -1;
        for (int r = rank; r < m; r++) if (A[(size_t)r * nw + w] & bit) { pr = r; break; }
        if (pr < 0) continue;
        if (pr != rank) for (int k = 0; k < nw; k++) std::swap(A[(size_t)pr * nw + k], A[(size_t)rank * nw + k]);
        for (int r = rank + 1; r < m; r++)
            if (A[(size_t)r * nw + w] & bit) for (int k = w; k < nw; k++) A[(size_t)r * nw + k] ^= A[(size_t)rank * nw + k];
        rank++;
    }
    return rank;
}

// 0: not applicable (use the global-memory elimination), 1: U in LDS (m <= 1024), 2: U in HBM/L2 (m <= 4096, "UG")
static int plan_osd_lds(const qldpc_graph *g, OsdLdsArgs &P, size_t &lds, int flags) {
    if (g->m > 4096 || g->n >= 65535 || g->m < 1) return 0;
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024; P.cdeg = std::max(g->max_col_deg, 1);
    P.npad = 1;
    while (P.npad < g->n) P.npad <<= 1;
    for (int mode = (g->m <= 1024 && !(flags & QLDPC_FLAG_OSD_UG)) ? 1 : 2; mode <= 2; mode++) {
        const size_t sort_cnt = (size_t)256 * 16 * 4 + 16 * 4 + 64;                         // [256][waves] radix counters + per-wave sums
        size_t off = (mode == 1) ? std::max((size_t)(g->m + 2) * P.mw * 8, (size_t)g->n * 12 + 16 + sort_cnt) : 0;     // U, aliased by the sort scratch
        off = (size_t)round_up((int64_t)off, 16);
        P.offIdx = (int)off; off += (size_t)P.K * 2;
        P.offAlive = (int)off; off += (size_t)P.K;
        P.offRows = (int)off; off += (size_t)P.K * P.cdeg * 2;
        P.offPc = (int)off; off += round_up((int64_t)g->m * 2, 8);
        P.offR = (int)off; off += (size_t)kOsdBlock * P.mw * 8;
        P.offBlk = (int)off; off += (4 + 6 * kOsdBlock + 4) * 4;
        P.offMisc = (int)off; off += 64;
        P.offSort = (int)off; off += (mode == 2) ? sort_cnt : 0;
        lds = off + 16;
        if (lds <= 160 * 1024) return mode;
    }
    return 0;
}

int osd0_gj_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                   const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled);

int osd0_gjg_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream, size_t ws_offset, bool &handled);

int osd0_lds_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, int64_t max_listed, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, int flags, hipStream_t stream, bool &handled, OsdJudge *judge) {
    OsdLdsArgs P;
    size_t lds = 0;
    handled = false;
    if (!(flags & (QLDPC_FLAG_OSD_LDS | QLDPC_FLAG_OSD_UG | QLDPC_FLAG_OSD_GLOBAL))) {      // small matrices: the literal elimination, one wave per shot
        const int rcs = osd0_small_launch(g, d_list, d_count, d_synd, d_llr, d_hard, d_ordering, d_solution, flags, stream, handled, judge);
        if (rcs != QLDPC_OK || handled) return rcs;
    }
#ifndef QLDPC_EXPERIMENTS
    if (flags & QLDPC_FLAG_OSD_QUEUE) {
        set_error("this OSD-0 variant (flags %#x) is a measured-and-rejected experiment: it exists in libqldpc_hip_experiments.so only (make experiments)", flags);
        return QLDPC_ERR_UNSUPPORTED;
    }
#endif
    if (!(flags & (QLDPC_FLAG_OSD_REFORDER | QLDPC_FLAG_OSD_GLOBAL))) {
        // the free-pivot kernels take every shot -- osd_gj.hip with the row transform in LDS (m <= 1024), osd_gjg.hip with it in HBM / L2 (m <= 4096, or
        // asked for by QLDPC_FLAG_OSD_UG); the shots they list (right-hand side outside the column space, where the answer depends on the reference's row
        // choice) go through the reference-order kernel below, behind them on the same stream
        bool took = false;
        if (!(flags & QLDPC_FLAG_OSD_UG)) {
            const int rcg = osd0_gj_launch(g, d_list, d_count, max_listed, d_synd, d_llr, d_hard, d_ordering, d_solution, flags, stream, took);
            if (rcg != QLDPC_OK) return rcg;
        }
        if (!took) {
            const int rcg = osd0_gjg_launch(g, d_list, d_count, max_listed, d_synd, d_llr, d_hard, d_ordering, d_solution, stream, 0, took);
            if (rcg != QLDPC_OK) return rcg;
        }
        if (took) { d_count = g->ws_redo.as<int32_t>(); d_list = d_count + 4; }
    }
    const int mode = (flags & QLDPC_FLAG_OSD_GLOBAL) ? 0 : plan_osd_lds(g, P, lds, flags);
    if (mode == 0) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_gf2_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    const int grid = 512;
    const size_t sz_ord = (size_t)round_up((int64_t)grid * g->n * 2 + 64, 16);
    const size_t sz_u = (mode == 2) ? (size_t)grid * (size_t)(g->m + 2) * P.mw * 8 : 0;
    const size_t per_keys = (size_t)g->n + (size_t)(g->n + 1) / 2;                       // u64 units: n keys + 2 x n u16 indices
    const size_t sz_k = (mode == 2) ? (size_t)grid * per_keys * 8 : 0;
    int rc = g->ws_misc.ensure(sz_ord + sz_u + sz_k);
    if (rc != QLDPC_OK) return rc;
    P.ordws = g->ws_misc.as<uint16_t>();
    P.ug = reinterpret_cast<unsigned long long *>(g->ws_misc.as<unsigned char>() + sz_ord);
    P.ugkeys = reinterpret_cast<unsigned long long *>(g->ws_misc.as<unsigned char>() + sz_ord + sz_u);
    P.indptr = g->d_indptr; P.indices = g->d_indices; P.colptr = g->d_colptr; P.rowidx = g->d_rowidx;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    P.clk = g->clk_probe;
    P.dbg = osd_timer_buffer();      // NULL unless built with -DQLDPC_OSD_TIMERS (make timers)
    const int block = (int)std::min<int64_t>(1024, round_up(std::max(g->m + 2, 256), 64));
    if ((rc = g->ws_queue.ensure(16)) != QLDPC_OK) return rc;
    P.queue = g->ws_queue.as<int>() + 2;
    QLDPC_HIP_TRY(hipMemsetAsync(P.queue, 0, 4, stream));
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_lds_kernel<false>), 160 * 1024)) != QLDPC_OK) return rc;
    if ((rc = ensure_max_lds(g->device, reinterpret_cast<const void *>(osd0_lds_kernel<true>), 160 * 1024)) != QLDPC_OK) return rc;
    if (mode == 2) hipLaunchKernelGGL(osd0_lds_kernel<true>, dim3(grid), dim3(1024), lds, stream, P);
    else hipLaunchKernelGGL(osd0_lds_kernel<false>, dim3(grid), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc

// ---- diagnostic phase counters (see osd_common.h) ----
namespace qldpc {
unsigned long long *osd_timer_buffer() {
#ifdef QLDPC_OSD_TIMERS
    static std::mutex mu;
    static unsigned long long *d_buf[64] = {nullptr};
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!d_buf[dev]) {
        if (hipMalloc(reinterpret_cast<void **>(&d_buf[dev]), 32 * 8) != hipSuccess) { d_buf[dev] = nullptr; return nullptr; }
        (void)zero_now(d_buf[dev], 32 * 8);
    }
    return d_buf[dev];
#else
    return nullptr;
#endif
}
}  // namespace qldpc

// Phase counters of the OSD-0 kernels ([0..15]) and of the workgroup BP kernel ([16..31]) accumulated on the CURRENT device since the last reset
// (uint64[32], layout in osd_common.h).
// Only the diagnostic build (make timers) counts; the default build returns QLDPC_ERR_UNSUPPORTED.  Synchronises the device.
QLDPC_EXPORT int qldpc_osd_timers_read(uint64_t *out, int reset) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    unsigned long long *d = qldpc::osd_timer_buffer();
    if (!d) { qldpc::set_error("OSD phase timers are compiled out of this build (make -C csrc timers)"); return QLDPC_ERR_UNSUPPORTED; }
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(out, d, 32 * 8, hipMemcpyDeviceToHost));
    if (reset) QLDPC_HIP_TRY(zero_now(d, 32 * 8));
    return QLDPC_OK;
}
