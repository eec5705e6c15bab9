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

#include <algorithm>
#include <cstdlib>
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

int osd0_lds_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream, bool &handled);

int osd0_listed_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                       const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream) {
    const int m = g->m, n = g->n;
    if (m == 0 || n == 0) return QLDPC_OK;
    {
        bool handled = false;           // LDS-resident kernel for m <= 1024; the global-memory kernel below is the general fallback
        const int rc0 = osd0_lds_launch(g, d_list, d_count, d_synd, d_llr, d_hard, d_ordering, d_solution, stream, handled);
        if (rc0 != QLDPC_OK || handled) return rc0;
    }
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
    int rc = use_device(0);
    if (rc != QLDPC_OK) return rc;
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
    QLDPC_HIP_TRY(hipMemset(dpr.p, 0, (size_t)B * maxp * 8));
    QLDPC_HIP_TRY(hipMemset(dpc.p, 0, (size_t)B * maxp * 8));
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
    int rc = use_device(0);
    if (rc != QLDPC_OK) return rc;
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
    QLDPC_HIP_TRY(hipMemset(dpr.p, 0, (size_t)B * maxp * 8));
    QLDPC_HIP_TRY(hipMemset(dpc.p, 0, (size_t)B * maxp * 8));
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
                                  const int32_t *ordering, int8_t *solution) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0 && B < ((int64_t)1 << 31), "batch out of range");
    int rc = use_device(g->device);
    if (rc != QLDPC_OK) return rc;
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
        rc = osd0_listed_launch(g, dlist.as<int32_t>(), dcnt.as<int32_t>(), ds.as<int8_t>(), dl.as<double>(), dh.as<int8_t>(),
                                ordering ? dord.as<int32_t>() : nullptr, dsol.as<int8_t>(), nullptr);
        if (rc == QLDPC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("OSD-0 kernel failed: %s", hipGetErrorString(hipGetLastError())); rc = QLDPC_ERR_HIP; }
    }
    if (rc != QLDPC_OK) return rc;
    QLDPC_HIP_TRY(hipMemcpy(solution, dsol.p, B * n, hipMemcpyDeviceToHost));
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
//   * columns are produced lazily in chunks of the K most reliable-to-flip ones (radix select + bitonic sort in LDS)
//     and the sweep stops as soon as rank(H) pivots exist -- later columns cannot pivot.
// The result (pivot columns, reduced rhs at the pivots) is identical to the reference's; tests compare solutions.
// =====================================================================================================================
namespace qldpc {

struct OsdLdsArgs {
    int m, n, mw, rankH, K;
    const int32_t *indptr, *indices, *colptr, *rowidx;
    const int32_t *list, *count;
    const int8_t *synd; const double *llr; const int8_t *hard; const int32_t *ordering;
    int8_t *solution;
    int offKey, offIdx, offPhys, offPos, offPr, offPc, offRc, offB, offHist, offMisc, offScan;
};

__device__ __forceinline__ unsigned long long osd_key(double x) {
    double a = fabs(x);
    if (a != a) a = INFINITY;
    return (unsigned long long)__double_as_longlong(a);          // non-negative doubles order like their bit patterns
}
__device__ __forceinline__ int uswz(int q, int w, int mw) { return q * mw + ((mw == 16) ? (w ^ (q & 15)) : w); }

__global__ __launch_bounds__(1024) void osd0_lds_kernel(OsdLdsArgs P) {
    extern __shared__ unsigned char lds[];
    const int m = P.m, n = P.n, mw = P.mw, K = P.K, tid = threadIdx.x, T = blockDim.x;
    unsigned long long *U = reinterpret_cast<unsigned long long *>(lds);
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(lds + P.offKey);
    uint16_t *sidx = reinterpret_cast<uint16_t *>(lds + P.offIdx);
    uint16_t *phys = reinterpret_cast<uint16_t *>(lds + P.offPhys), *pos_of = reinterpret_cast<uint16_t *>(lds + P.offPos);
    uint16_t *pvrow = reinterpret_cast<uint16_t *>(lds + P.offPr), *pvcol = reinterpret_cast<uint16_t *>(lds + P.offPc);
    unsigned long long *rc = reinterpret_cast<unsigned long long *>(lds + P.offRc);
    unsigned long long *bvec = reinterpret_cast<unsigned long long *>(lds + P.offB);
    unsigned int *hist = reinterpret_cast<unsigned int *>(lds + P.offHist);
    unsigned int *misc = reinterpret_cast<unsigned int *>(lds + P.offMisc);     // [0] best, [1] compaction counter, [2..7] select state
    unsigned int *scan = reinterpret_cast<unsigned int *>(lds + P.offScan);     // [T] tie counts

    const int total = *P.count;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t shot = P.list[item];
        const double *llr = P.llr + shot * n;
        const int8_t *hard = P.hard + shot * n, *synd = P.synd + shot * m;
        int8_t *sol = P.solution + shot * n;
        // ---- init: T = I, positions = identity, b = s + H hard (osd.py:8-9) ----
        for (int t = tid; t < m * mw; t += T) U[t] = 0ull;
        for (int w = tid; w < mw; w += T) bvec[w] = 0ull;
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            U[uswz(r, r >> 6, mw)] = 1ull << (r & 63);
            phys[r] = (uint16_t)r; pos_of[r] = (uint16_t)r;
            int s = synd[r] & 1;
            for (int e = P.indptr[r]; e < P.indptr[r + 1]; e++) s ^= hard[P.indices[e]] & 1;
            if (s) atomicOr(&bvec[r >> 6], 1ull << (r & 63));
        }
        __syncthreads();
        int row = 0, npiv = 0, chunk_base = 0;
        unsigned long long last_key = 0ull;
        int last_idx = -1;
        bool finished = false;
        while (!finished) {
            // ================= next chunk of columns in reliability order =================
            int L = 0;
            if (P.ordering) {
                L = min(K, n - chunk_base);
                for (int c = tid; c < L; c += T) sidx[c] = (uint16_t)P.ordering[shot * n + chunk_base + c];
                chunk_base += L;
                __syncthreads();
            } else {
                // eligible(j): (key_j, j) > (last_key, last_idx).  Select the K smallest, sort them by (key, index).
                if (tid == 0) { misc[1] = 0; misc[2] = 0; }
                __syncthreads();
                unsigned int elig = 0;
                for (int j = tid; j < n; j += T) {
                    const unsigned long long k = osd_key(llr[j]);
                    elig += (k > last_key || (k == last_key && j > last_idx)) ? 1u : 0u;
                }
                if (elig) atomicAdd(&misc[2], elig);
                __syncthreads();
                const unsigned int E = misc[2];
                unsigned long long thr_key = ~0ull;        // take keys < thr_key entirely, and `need` ties with key == thr_key
                unsigned int need = 0;
                if (E > (unsigned)K) {
                    unsigned long long prefix = 0ull, pmask = 0ull;
                    unsigned int want = (unsigned)K;
                    for (int pass = 7; pass >= 0; pass--) {
                        for (int d = tid; d < 256; d += T) hist[d] = 0;
                        __syncthreads();
                        for (int j = tid; j < n; j += T) {
                            const unsigned long long k = osd_key(llr[j]);
                            if ((k > last_key || (k == last_key && j > last_idx)) && (k & pmask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255], 1u);
                        }
                        __syncthreads();
                        if (tid == 0) {
                            unsigned int cum = 0, d = 0;
                            for (; d < 255; d++) { if (cum + hist[d] >= want) break; cum += hist[d]; }
                            misc[3] = d; misc[4] = cum;
                        }
                        __syncthreads();
                        prefix |= (unsigned long long)misc[3] << (8 * pass);
                        pmask |= 255ull << (8 * pass);
                        want -= misc[4];
                        __syncthreads();
                    }
                    thr_key = prefix; need = want;
                }
                // compaction: keys below the threshold in any order, ties in ascending index order
                for (int c = tid; c < K; c += T) { skey[c] = ~0ull; sidx[c] = 0xFFFF; }
                __syncthreads();
                const int per = (n + T - 1) / T, j0 = tid * per, j1 = min(n, j0 + per);
                unsigned int myties = 0;
                for (int j = j0; j < j1; j++) {
                    const unsigned long long k = osd_key(llr[j]);
                    if (!(k > last_key || (k == last_key && j > last_idx))) continue;
                    if (k < thr_key) { const unsigned int s = atomicAdd(&misc[1], 1u); skey[s] = k; sidx[s] = (uint16_t)j; }
                    else if (k == thr_key) myties++;
                }
                scan[tid] = myties;
                __syncthreads();
                if (E > (unsigned)K) {
                    if (tid == 0) { unsigned int run = 0; for (int t = 0; t < T; t++) { const unsigned int c = scan[t]; scan[t] = run; run += c; } }
                    __syncthreads();
                    unsigned int rank = scan[tid];
                    const unsigned int base = misc[1];          // number of keys below the threshold (= K - need)
                    for (int j = j0; j < j1 && rank < need; j++) {
                        const unsigned long long k = osd_key(llr[j]);
                        if ((k > last_key || (k == last_key && j > last_idx)) && k == thr_key) { skey[base + rank] = k; sidx[base + rank] = (uint16_t)j; rank++; }
                    }
                }
                __syncthreads();
                L = (int)min(E, (unsigned)K);
                // bitonic sort of the K slots by (key, index); empty slots (key ~0, index 0xFFFF) sink to the end
                for (int size = 2; size <= K; size <<= 1)
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        for (int i = tid; i < K; i += T) {
                            const int p = i ^ stride;
                            if (p > i) {
                                const unsigned long long ka = skey[i], kb = skey[p];
                                const uint16_t ia = sidx[i], ib = sidx[p];
                                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                                if (gt == ((i & size) == 0)) { skey[i] = kb; skey[p] = ka; sidx[i] = ib; sidx[p] = ia; }
                            }
                        }
                        __syncthreads();
                    }
                if (L > 0) { last_key = skey[L - 1]; last_idx = sidx[L - 1]; }
                __syncthreads();
            }
            if (L == 0) break;
            // ================= sweep the chunk (kernels.py:64-94 on the reduced sparse columns) =================
            for (int c = 0; c < L; c++) {
                const int j = sidx[c];
                if (tid < mw) {
                    unsigned long long w = 0ull;
                    for (int k = P.colptr[j]; k < P.colptr[j + 1]; k++) w ^= U[uswz(P.rowidx[k], tid, mw)];
                    rc[tid] = w;
                }
                unsigned int *bestp = &misc[8 + (c & 1)];         // two slots: a fast thread may already reset the next column's slot
                if (tid == 0) *bestp = 0xFFFFFFFFu;
                __syncthreads();
                for (int r = tid; r < m; r += T)
                    if (((rc[r >> 6] >> (r & 63)) & 1ull) && pos_of[r] >= row) atomicMin(bestp, ((unsigned int)pos_of[r] << 16) | (unsigned int)r);
                __syncthreads();
                const unsigned int best = *bestp;
                if (best != 0xFFFFFFFFu) {
                    const int pr = (int)(best & 0xFFFFu), ppos = (int)(best >> 16);
                    const unsigned long long prbit = 1ull << (pr & 63);
                    const int prw = pr >> 6;
                    const bool bpr = (bvec[prw] & prbit) != 0ull;
                    for (int q = tid; q < m; q += T)                                          // rows with a one in this column get the pivot row added
                        if (U[uswz(q, prw, mw)] & prbit)
                            for (int w = 0; w < mw; w++) U[uswz(q, w, mw)] ^= (w == prw) ? (rc[w] & ~prbit) : rc[w];
                    __syncthreads();
                    if (tid < mw && bpr) bvec[tid] ^= (tid == prw) ? (rc[tid] & ~prbit) : rc[tid];
                    if (tid == 0) {
                        const int r0 = phys[row];
                        phys[row] = (uint16_t)pr; phys[ppos] = (uint16_t)r0; pos_of[pr] = (uint16_t)row; pos_of[r0] = (uint16_t)ppos;
                        pvrow[npiv] = (uint16_t)pr; pvcol[npiv] = (uint16_t)j;
                    }
                    row++; npiv++;
                    __syncthreads();
                    if (row >= P.rankH || row >= m) { finished = true; break; }
                }
            }
            if (L < K || (P.ordering && chunk_base >= n)) finished = true;
        }
        // ---- back-fill (osd.py:19-25): e[pivot col] = reduced rhs at the pivot row; solution = (hard + e) % 2 ----
        __syncthreads();
        if (sol != hard) for (int j = tid; j < n; j += T) sol[j] = hard[j];
        __syncthreads();
        for (int t = tid; t < npiv; t += T) {
            const int j = pvcol[t], r = pvrow[t];
            sol[j] = (int8_t)((hard[j] ^ (int8_t)((bvec[r >> 6] >> (r & 63)) & 1ull)) & 1);
        }
        __syncthreads();
    }
}

// rank of H over GF(2) (host, once per graph): the sweep above can stop as soon as this many pivots exist
static int host_rank(const qldpc_graph *g) {
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

static bool plan_osd_lds(const qldpc_graph *g, OsdLdsArgs &P, size_t &lds) {
    if (g->m > 1024 || g->n >= 65535 || g->m < 1) return false;
    P.m = g->m; P.n = g->n; P.mw = (g->m + 63) / 64; P.K = 1024;
    size_t off = (size_t)g->m * P.mw * 8;
    P.offKey = (int)off; off += (size_t)P.K * 8;
    P.offIdx = (int)off; off += (size_t)P.K * 2;
    P.offPhys = (int)off; off += round_up((size_t)g->m * 2, 8);
    P.offPos = (int)off; off += round_up((size_t)g->m * 2, 8);
    P.offPr = (int)off; off += round_up((size_t)g->m * 2, 8);
    P.offPc = (int)off; off += round_up((size_t)g->m * 2, 8);
    P.offRc = (int)off; off += (size_t)P.mw * 8;
    P.offB = (int)off; off += (size_t)P.mw * 8;
    P.offHist = (int)off; off += 256 * 4;
    P.offMisc = (int)off; off += 64;
    P.offScan = (int)off; off += 1024 * 4;
    lds = off + 16;
    return lds <= 160 * 1024;
}

int osd0_lds_launch(const qldpc_graph *g, const int32_t *d_list, const int32_t *d_count, const int8_t *d_synd, const double *d_llr,
                    const int8_t *d_hard, const int32_t *d_ordering, int8_t *d_solution, hipStream_t stream, bool &handled) {
    OsdLdsArgs P;
    size_t lds = 0;
    handled = false;
    if (getenv("QLDPC_OSD_GLOBAL") || !plan_osd_lds(g, P, lds)) return QLDPC_OK;
    if (g->gf2_rank < 0) g->gf2_rank = host_rank(g);      // callers hold g->mu
    P.rankH = g->gf2_rank;
    P.indptr = g->d_indptr; P.indices = g->d_indices; P.colptr = g->d_colptr; P.rowidx = g->d_rowidx;
    P.list = d_list; P.count = d_count; P.synd = d_synd; P.llr = d_llr; P.hard = d_hard; P.ordering = d_ordering; P.solution = d_solution;
    static bool attr_set = false;
    if (!attr_set) {
        QLDPC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(osd0_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int block = (int)std::min<int64_t>(1024, round_up(std::max(g->m, 256), 64));
    hipLaunchKernelGGL(osd0_lds_kernel, dim3(512), dim3(block), lds, stream, P);
    QLDPC_HIP_TRY(hipGetLastError());
    handled = true;
    return QLDPC_OK;
}

}  // namespace qldpc
