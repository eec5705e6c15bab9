// Single check-node passes on the CSR edge layout (a3 minsum_core_sparse, a5 bp_core) and the sum-product
// driver performBeliefPropagationFast (a5).  Edge messages are laid out [shot][edge] as in the reference call.
#include "common.h"
#include "mc_common.h"

#include <algorithm>
#include <cstring>

namespace qldpc {

// one thread per (shot, row).  BP = false: kernels.py:144-168; BP = true: kernels.py:176-192 (clip_val in `param`).
template <bool BP>
__global__ void check_pass_kernel(int64_t B, int m, int nnz, const int32_t *__restrict__ indptr, const double *__restrict__ Q,
                                  const double *__restrict__ ssign, const uint8_t *__restrict__ skip, double param,
                                  double *__restrict__ R) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * m) return;
    const int64_t b = t / m;
    const int i = (int)(t - b * m);
    if (skip && skip[b]) return;
    const int rs = indptr[i], re = indptr[i + 1];
    if (rs == re) return;
    const double *q = Q + b * nnz;
    double *r = R + b * nnz;
    const double ss = ssign[t];
    if (!BP) {
        double sign_prod = ss, min1 = INFINITY, min2 = INFINITY;
        int min1_pos = -1;
        for (int pos = rs; pos < re; pos++) {
            const double val = q[pos];
            sign_prod *= (val >= 0) ? 1.0 : -1.0;
            const double a = fabs(val);
            if (a < min1) { min2 = min1; min1 = a; min1_pos = pos; }
            else if (a < min2) { min2 = a; }
        }
        for (int pos = rs; pos < re; pos++) {
            const double val = q[pos];
            const double sign_j = (val >= 0) ? 1.0 : -1.0;
            const double mag = (pos == min1_pos) ? min2 : min1;
            r[pos] = param * (sign_prod * sign_j) * mag;
        }
    } else {
        double row_prod = 1.0;
        for (int pos = rs; pos < re; pos++) {
            double th = tanh(q[pos] * 0.5);
            if (fabs(th) < 1e-15) th = (th >= 0) ? 1e-15 : -1e-15;
            row_prod *= th;
        }
        for (int pos = rs; pos < re; pos++) {
            double th = tanh(q[pos] * 0.5);
            if (fabs(th) < 1e-15) th = (th >= 0) ? 1e-15 : -1e-15;
            double pc = (row_prod / th) * ss;
            if (pc < -param) pc = -param; else if (pc > param) pc = param;
            r[pos] = 2.0 * atanh(pc);
        }
    }
}

// R_sum[col] = ((0 + R[e1]) + R[e2]) + ... over the column's edges in ascending check order (kernels.py:168 / np.sum(R,axis=0));
// `prior` != NULL adds the prior afterwards (values = R_sum + initialBelief).  One thread per (shot, column).
__global__ void column_sum_kernel(int64_t B, int n, int nnz, const int32_t *__restrict__ colptr, const int32_t *__restrict__ csc2csr,
                                  const double *__restrict__ R, const double *__restrict__ prior, const uint8_t *__restrict__ skip,
                                  double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    const int64_t b = t / n;
    const int j = (int)(t - b * n);
    if (skip && skip[b]) return;
    const double *r = R + b * nnz;
    double s = 0.0;
    for (int k = colptr[j]; k < colptr[j + 1]; k++) s += r[csc2csr[k]];
    out[t] = prior ? s + prior[j] : s;
}

__global__ void bp_init_kernel(int64_t B, int m, int n, int nnz, const int32_t *__restrict__ indices, const double *__restrict__ prior,
                               const int8_t *__restrict__ synd, double *__restrict__ Q, double *__restrict__ ssign,
                               uint8_t *__restrict__ done, uint8_t *__restrict__ unsat) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B * nnz) Q[t] = prior[indices[t % nnz]];
    if (t < B * m) ssign[t] = (double)(1 - 2 * (int)synd[t]);
    if (t < B) { done[t] = 0; unsat[t] = 0; }
}

// Q = values[col] - R (dense.py:91) and the syndrome test (dense.py:92-94), one thread per (shot,row)
__global__ void bp_update_kernel(int64_t B, int m, int n, int nnz, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                 const double *__restrict__ values, const double *__restrict__ R, const int8_t *__restrict__ synd,
                                 const uint8_t *__restrict__ done, double *__restrict__ Q, uint8_t *__restrict__ unsat) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * m) return;
    const int64_t b = t / m;
    const int i = (int)(t - b * m);
    if (done[b]) return;
    const double *v = values + b * n;
    int s = 0;
    for (int pos = indptr[i]; pos < indptr[i + 1]; pos++) {
        const double x = v[indices[pos]];
        s ^= (x < 0) ? 1 : 0;
        Q[b * nnz + pos] = x - R[b * nnz + pos];
    }
    if (s != (int)synd[t]) unsat[b] = 1;
}

__global__ void bp_finalize_kernel(int64_t B, int n, int it, int last, const double *__restrict__ values, uint8_t *__restrict__ done,
                                   uint8_t *__restrict__ unsat, int8_t *__restrict__ out_err, double *__restrict__ out_llr,
                                   uint8_t *__restrict__ out_conv, int32_t *__restrict__ out_iter) {
    const int64_t b = (int64_t)blockIdx.x;
    if (b >= B || done[b]) return;
    const bool conv = !unsat[b];
    __syncthreads();
    if (conv || last) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const double x = values[b * n + j];
            out_llr[b * n + j] = x; out_err[b * n + j] = (x < 0) ? 1 : 0;
        }
        if (threadIdx.x == 0) { out_conv[b] = conv ? 1 : 0; out_iter[b] = it; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { if (conv) done[b] = 1; unsat[b] = 0; }
}

static unsigned blocks(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace qldpc

using namespace qldpc;

static int check_pass_host(const qldpc_graph *g, int64_t B, const double *Q, const double *ssign, double param, bool bp, double *R,
                           double *Rsum) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0, "negative batch");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(Q && ssign && R && Rsum, "NULL buffer");
    const size_t m = g->m, n = g->n, nnz = g->nnz;
    DevTmp dQ, dS, dR, dRs;
    if ((rc = dQ.alloc(B * nnz * 8)) || (rc = dS.alloc(B * m * 8)) || (rc = dR.alloc(B * nnz * 8)) || (rc = dRs.alloc(B * n * 8))) return rc;
    if (nnz) QLDPC_HIP_TRY(hipMemcpy(dQ.p, Q, B * nnz * 8, hipMemcpyHostToDevice));
    if (m) QLDPC_HIP_TRY(hipMemcpy(dS.p, ssign, B * m * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(zero_now(dR.p, B * nnz * 8 + (nnz ? 0 : 16)));
    if (B * m > 0) {
        if (bp) hipLaunchKernelGGL(check_pass_kernel<true>, dim3(blocks(B * m)), dim3(256), 0, nullptr, B, (int)m, (int)nnz, g->d_indptr,
                                   dQ.as<double>(), dS.as<double>(), (const uint8_t *)nullptr, param, dR.as<double>());
        else hipLaunchKernelGGL(check_pass_kernel<false>, dim3(blocks(B * m)), dim3(256), 0, nullptr, B, (int)m, (int)nnz, g->d_indptr,
                                dQ.as<double>(), dS.as<double>(), (const uint8_t *)nullptr, param, dR.as<double>());
    }
    if (B * n > 0)
        hipLaunchKernelGGL(column_sum_kernel, dim3(blocks(B * n)), dim3(256), 0, nullptr, B, (int)n, (int)nnz, g->d_colptr, g->d_csc2csr,
                           dR.as<double>(), (const double *)nullptr, (const uint8_t *)nullptr, dRs.as<double>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    if (nnz) QLDPC_HIP_TRY(hipMemcpy(R, dR.p, B * nnz * 8, hipMemcpyDeviceToHost));
    if (n) QLDPC_HIP_TRY(hipMemcpy(Rsum, dRs.p, B * n * 8, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_minsum_check_pass(const qldpc_graph *g, int64_t B, const double *Q, const double *syndrome_sign, double alpha,
                                         double *R, double *R_sum) {
    return check_pass_host(g, B, Q, syndrome_sign, alpha, false, R, R_sum);
}

QLDPC_EXPORT int qldpc_bp_check_pass(const qldpc_graph *g, int64_t B, const double *Q, const double *syndrome_sign, double clip_val,
                                     double *R, double *R_sum) {
    return check_pass_host(g, B, Q, syndrome_sign, clip_val, true, R, R_sum);
}

QLDPC_EXPORT int qldpc_bp_decode_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *prior, int max_iter,
                                       int8_t *out_err, double *out_llr, uint8_t *out_conv, int32_t *out_iter) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0 && max_iter >= 1, "bad batch / max_iter (performBeliefPropagationFast needs max_iter >= 1)");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(prior && out_err && out_llr && out_conv && out_iter && (syndromes || g->m == 0), "NULL buffer");
    const size_t m = g->m, n = g->n, nnz = g->nnz;
    DevTmp dsy, dpr, dQ, dR, dS, dV, ddone, dunsat, de, dl, dc, di;
    if ((rc = dsy.alloc(B * m)) || (rc = dpr.alloc(n * 8)) || (rc = dQ.alloc(B * nnz * 8)) || (rc = dR.alloc(B * nnz * 8)) ||
        (rc = dS.alloc(B * m * 8)) || (rc = dV.alloc(B * n * 8)) || (rc = ddone.alloc(B)) || (rc = dunsat.alloc(B)) ||
        (rc = de.alloc(B * n)) || (rc = dl.alloc(B * n * 8)) || (rc = dc.alloc(B)) || (rc = di.alloc(B * 4)))
        return rc;
    if (m) QLDPC_HIP_TRY(hipMemcpy(dsy.p, syndromes, B * m, hipMemcpyHostToDevice));
    if (n) QLDPC_HIP_TRY(hipMemcpy(dpr.p, prior, n * 8, hipMemcpyHostToDevice));
    size_t tot = B * nnz; if (B * m > tot) tot = B * m; if ((size_t)B > tot) tot = B;
    hipLaunchKernelGGL(bp_init_kernel, dim3(blocks(tot)), dim3(256), 0, nullptr, B, (int)m, (int)n, (int)nnz, g->d_indices, dpr.as<double>(),
                       dsy.as<int8_t>(), dQ.as<double>(), dS.as<double>(), ddone.as<uint8_t>(), dunsat.as<uint8_t>());
    QLDPC_HIP_TRY(zero_now(dR.p, B * nnz * 8 + (nnz ? 0 : 16)));
    for (int it = 0; it < max_iter; it++) {
        if (B * m > 0)
            hipLaunchKernelGGL(check_pass_kernel<true>, dim3(blocks(B * m)), dim3(256), 0, nullptr, B, (int)m, (int)nnz, g->d_indptr,
                               dQ.as<double>(), dS.as<double>(), ddone.as<uint8_t>(), 0.9999999, dR.as<double>());     // dense.py:84,87
        if (B * n > 0)
            hipLaunchKernelGGL(column_sum_kernel, dim3(blocks(B * n)), dim3(256), 0, nullptr, B, (int)n, (int)nnz, g->d_colptr, g->d_csc2csr,
                               dR.as<double>(), dpr.as<double>(), ddone.as<uint8_t>(), dV.as<double>());                // dense.py:88-89
        if (B * m > 0)
            hipLaunchKernelGGL(bp_update_kernel, dim3(blocks(B * m)), dim3(256), 0, nullptr, B, (int)m, (int)n, (int)nnz, g->d_indptr,
                               g->d_indices, dV.as<double>(), dR.as<double>(), dsy.as<int8_t>(), ddone.as<uint8_t>(), dQ.as<double>(),
                               dunsat.as<uint8_t>());
        hipLaunchKernelGGL(bp_finalize_kernel, dim3((unsigned)B), dim3(64), 0, nullptr, B, (int)n, it, it == max_iter - 1 ? 1 : 0,
                           dV.as<double>(), ddone.as<uint8_t>(), dunsat.as<uint8_t>(), de.as<int8_t>(), dl.as<double>(), dc.as<uint8_t>(),
                           di.as<int32_t>());
        QLDPC_HIP_TRY(hipGetLastError());
    }
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    if (n) QLDPC_HIP_TRY(hipMemcpy(out_err, de.p, B * n, hipMemcpyDeviceToHost));
    if (n) QLDPC_HIP_TRY(hipMemcpy(out_llr, dl.p, B * n * 8, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(out_conv, dc.p, B, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(out_iter, di.p, B * 4, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// f4: the trial loops of the alpha / beta estimators (reference src/decoding/alpha.py:119-137, 206-255 and
// src/decoding/scopt.py:80-134), batched.  The caller draws the error patterns (the reference uses the caller's numpy
// Generator, so the draws stay on the host and bit-identical); everything from the syndrome to the two histograms runs here.
// Samples stay resident in HBM between the range pass and the histogram pass ([trial][edge] or [trial][column] f64).
// ---------------------------------------------------------------------------------------------------------------------
namespace qldpc {

// alpha.py:226-244 / scopt.py:101-118, one thread per (trial, edge); the damping expression is evaluated even for damping == 1
__global__ void stats_q_update_kernel(int64_t B, int n, int nnz, const int32_t *__restrict__ indices, const double *__restrict__ values,
                                      const double *__restrict__ R, double damping, double clip, double *__restrict__ Q,
                                      double *__restrict__ Qold) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * nnz) return;
    const int64_t b = t / nnz;
    const int pos = (int)(t - b * nnz);
    double q = values[b * n + indices[pos]] - R[t];
    if (q != q) q = 0.0;
    else if (q > clip) q = clip;
    else if (q < -clip) q = -clip;
    double qd = damping * q + (1.0 - damping) * Qold[t];
    if (qd > clip) qd = clip;
    else if (qd < -clip) qd = -clip;
    Q[t] = qd;
    Qold[t] = qd;
}

// order-preserving map f64 -> u64 so that atomicMin / atomicMax on the key order the doubles
__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
static double key_f64(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double x;
    std::memcpy(&x, &u, 8);
    return x;
}

// red[0] = min key, red[1] = max key over the finite samples of both classes, red[2 + c] = number of finite samples of class c
__global__ __launch_bounds__(256) void stats_range_kernel(int64_t B, int L, int n, const int32_t *__restrict__ col_of, const double *__restrict__ X,
                                                          const int8_t *__restrict__ err, unsigned long long *__restrict__ red) {
    __shared__ unsigned long long s_min, s_max, s_cnt[2];
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    unsigned long long lo = ~0ull, hi = 0ull;
    unsigned c0 = 0, c1 = 0;
    const int64_t total = B * L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const double x = X[t];
        if (!isfinite(x)) continue;                                                          // alpha.py:23-24 / scopt.py:142-143
        const int64_t b = t / L;
        const int e = (int)(t - b * L);
        const int bit = err[b * n + (col_of ? col_of[e] : e)];                               // alpha.py:135-137 / scopt.py:132-134
        if (bit) c1++; else c0++;
        const unsigned long long k = f64_key(x);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    if (c0 | c1) {
        atomicMin(&s_min, lo); atomicMax(&s_max, hi);
        if (c0) atomicAdd(&s_cnt[0], (unsigned long long)c0);
        if (c1) atomicAdd(&s_cnt[1], (unsigned long long)c1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && (s_cnt[0] | s_cnt[1])) {
        atomicMin(&red[0], s_min); atomicMax(&red[1], s_max);
        if (s_cnt[0]) atomicAdd(&red[2], s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&red[3], s_cnt[1]);
    }
}

constexpr int kStatsMaxBins = 4096;

// np.histogram with explicit uniform edges: bin i holds edges[i] <= x < edges[i+1], the last bin also x == edges[bins];
// samples outside [edges[0], edges[bins]] and non-finite samples are dropped.  hist: [2][bins] (class-major).
__global__ __launch_bounds__(256) void stats_hist_kernel(int64_t B, int L, int n, const int32_t *__restrict__ col_of, const double *__restrict__ X,
                                                         const int8_t *__restrict__ err, const double *__restrict__ edges, int bins,
                                                         unsigned long long *__restrict__ hist) {
    extern __shared__ unsigned char smem[];
    double *s_edge = reinterpret_cast<double *>(smem);                        // [bins + 1]
    unsigned *s_hist = reinterpret_cast<unsigned *>(s_edge + bins + 1);       // [2][bins]; a block adds < 2^32 samples
    for (int i = threadIdx.x; i <= bins; i += blockDim.x) s_edge[i] = edges[i];
    for (int i = threadIdx.x; i < 2 * bins; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const double first = s_edge[0], last = s_edge[bins];
    const int64_t total = B * L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const double x = X[t];
        if (!isfinite(x) || x < first || x > last) continue;
        int lo = 0, hi = bins;                                                // invariant: edges[lo] <= x, and (hi == bins or x < edges[hi])
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (x >= s_edge[mid]) lo = mid; else hi = mid;
        }
        const int64_t b = t / L;
        const int e = (int)(t - b * L);
        const int bit = err[b * n + (col_of ? col_of[e] : e)] ? 1 : 0;
        atomicAdd(&s_hist[bit * bins + lo], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * bins; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

int build_alpha_table(int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len, std::vector<double> &tab);

}  // namespace qldpc

struct qldpc_msgstats {
    const qldpc_graph *g = nullptr;
    int64_t B = 0;
    int L = 0;                       // samples per trial: nnz (check messages) or n (posteriors)
    bool by_edge = false;
    DevBuf d_samples, d_err;
};

QLDPC_EXPORT void qldpc_msgstats_destroy(qldpc_msgstats *S) {
    if (!S) return;
    (void)hipSetDevice(S->g->device);
    S->d_samples.release();
    S->d_err.release();
    delete S;
}

QLDPC_EXPORT int qldpc_msgstats_create(const qldpc_graph *g, int64_t B, const int8_t *errors, const double *prior, int kind, int iters,
                                       int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len, double damping,
                                       double clip_llr, double *range, int64_t *finite, qldpc_msgstats **out) {
    QLDPC_REQUIRE(g != nullptr && out != nullptr && range != nullptr && finite != nullptr, "NULL argument");
    QLDPC_REQUIRE(kind == QLDPC_STATS_CHECK_MESSAGES || kind == QLDPC_STATS_POSTERIOR, "unknown statistics kind %d", kind);
    QLDPC_REQUIRE(B >= 0 && iters >= 0, "negative trial count / iteration count");
    QLDPC_REQUIRE(kind != QLDPC_STATS_POSTERIOR || iters >= 1, "maxIter must be > 0");                      // scopt.py:52-53
    QLDPC_REQUIRE(B == 0 || (errors != nullptr && (prior != nullptr || g->n == 0)), "NULL buffer");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    *out = nullptr;
    const size_t m = g->m, n = g->n, nnz = g->nnz;
    const bool by_edge = (kind == QLDPC_STATS_CHECK_MESSAGES);
    const size_t L = by_edge ? nnz : n;
    QLDPC_REQUIRE((double)B * (double)L * 8.0 < 200e9, "sample array of %lld x %zu doubles does not fit the device", (long long)B, L);
    std::vector<double> tab;
    if (by_edge && (rc = build_alpha_table(iters, alpha_mode, alpha_val, alpha_seq, alpha_len, tab)) != QLDPC_OK) return rc;

    qldpc_msgstats *S = new qldpc_msgstats;
    S->g = g; S->B = B; S->L = (int)L; S->by_edge = by_edge;
    auto fail = [&](int code) { qldpc_msgstats_destroy(S); return code; };
    DevTmp d_prior, d_red;
    if ((rc = S->d_samples.ensure((size_t)B * L * 8 + 16)) || (rc = S->d_err.ensure((size_t)B * n + 16)) || (rc = d_prior.alloc(n * 8)) ||
        (rc = d_red.alloc(32)))
        return fail(rc);
    hipError_t he = hipSuccess;
    if (B * n) he = hipMemcpy(S->d_err.p, errors, (size_t)B * n, hipMemcpyHostToDevice);
    if (he == hipSuccess && n) he = hipMemcpy(d_prior.p, prior, n * 8, hipMemcpyHostToDevice);
    if (he != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(he)); return fail(QLDPC_ERR_HIP); }

    // trials are independent: process them in chunks so the [trial][edge] temporaries stay below ~6 GB
    const size_t per_trial = by_edge ? (nnz * 16 + n * 8 + m * 9 + 2) : (m + n + 5);
    int64_t chunk = (int64_t)(((size_t)6 << 30) / (per_trial ? per_trial : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    DevTmp d_synd, d_Q, d_Qold, d_ss, d_val, d_done, d_unsat, d_hard, d_conv, d_iter;
    if (B > 0) {
        if ((rc = d_synd.alloc(chunk * m))) return fail(rc);
        if (by_edge) {
            if ((rc = d_Q.alloc(chunk * nnz * 8)) || (rc = d_Qold.alloc(chunk * nnz * 8)) || (rc = d_ss.alloc(chunk * m * 8)) ||
                (rc = d_val.alloc(chunk * n * 8)) || (rc = d_done.alloc(chunk)) || (rc = d_unsat.alloc(chunk)))
                return fail(rc);
        } else if ((rc = d_hard.alloc(chunk * n)) || (rc = d_conv.alloc(chunk)) || (rc = d_iter.alloc(chunk * 4))) {
            return fail(rc);
        }
    }
    for (int64_t off = 0; off < B; off += chunk) {
        const int64_t Bc = std::min<int64_t>(chunk, B - off);
        const int8_t *err_c = S->d_err.as<int8_t>() + off * n;
        double *smp_c = S->d_samples.as<double>() + off * L;
        if ((rc = gf2_spmv_launch(g, Bc, err_c, d_synd.as<int8_t>(), nullptr)) != QLDPC_OK) return fail(rc);      // alpha.py:121 / scopt.py:82
        if (by_edge) {
            size_t tot = Bc * nnz; if (Bc * m > tot) tot = Bc * m; if ((size_t)Bc > tot) tot = Bc;
            hipLaunchKernelGGL(bp_init_kernel, dim3(blocks(tot)), dim3(256), 0, nullptr, Bc, (int)m, (int)n, (int)nnz, g->d_indices,
                               d_prior.as<double>(), d_synd.as<int8_t>(), d_Q.as<double>(), d_ss.as<double>(), d_done.as<uint8_t>(),
                               d_unsat.as<uint8_t>());                                                           // alpha.py:122-124
            if (nnz && hipMemcpyAsync(d_Qold.p, d_Q.p, Bc * nnz * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) {
                set_error("device copy failed"); return fail(QLDPC_ERR_HIP);
            }
            if (hipMemsetAsync(smp_c, 0, Bc * nnz * 8 + (nnz ? 0 : 8), nullptr) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
            for (int k = 0; k <= iters; k++) {
                const double a = (k < iters) ? tab[k] : 1.0;                                                     // alpha.py:214-218, 247-249
                if (Bc * m > 0)
                    hipLaunchKernelGGL(check_pass_kernel<false>, dim3(blocks(Bc * m)), dim3(256), 0, nullptr, Bc, (int)m, (int)nnz, g->d_indptr,
                                       d_Q.as<double>(), d_ss.as<double>(), (const uint8_t *)nullptr, a, smp_c);
                if (k == iters) break;
                if (Bc * n > 0)
                    hipLaunchKernelGGL(column_sum_kernel, dim3(blocks(Bc * n)), dim3(256), 0, nullptr, Bc, (int)n, (int)nnz, g->d_colptr,
                                       g->d_csc2csr, smp_c, d_prior.as<double>(), (const uint8_t *)nullptr, d_val.as<double>());   // alpha.py:220
                if (Bc * nnz > 0)
                    hipLaunchKernelGGL(stats_q_update_kernel, dim3(blocks(Bc * nnz)), dim3(256), 0, nullptr, Bc, (int)n, (int)nnz, g->d_indices,
                                       d_val.as<double>(), smp_c, damping, clip_llr, d_Q.as<double>(), d_Qold.as<double>());
            }
            if (hipGetLastError() != hipSuccess) { set_error("kernel launch failed"); return fail(QLDPC_ERR_HIP); }
        } else {
            rc = qldpc_minsum_decode_batch_dev(g, Bc, d_synd.as<int8_t>(), d_prior.as<double>(), iters, alpha_mode, alpha_val, alpha_seq,
                                               alpha_len, damping, clip_llr, 0, d_hard.as<int8_t>(), smp_c, d_conv.as<uint8_t>(),
                                               d_iter.as<int32_t>(), nullptr);                                   // scopt.py:88-131
            if (rc != QLDPC_OK) return fail(rc);
        }
        if (hipDeviceSynchronize() != hipSuccess) { set_error("device synchronisation failed"); return fail(QLDPC_ERR_HIP); }
    }
    unsigned long long red[4] = {~0ull, 0ull, 0ull, 0ull};
    if (hipMemcpy(d_red.p, red, 32, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload failed"); return fail(QLDPC_ERR_HIP); }
    if (B * (int64_t)L > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(4096, (B * (int64_t)L + 255) / 256);
        hipLaunchKernelGGL(stats_range_kernel, dim3(grid), dim3(256), 0, nullptr, B, (int)L, (int)n, by_edge ? g->d_indices : (const int32_t *)nullptr,
                           S->d_samples.as<double>(), S->d_err.as<int8_t>(), d_red.as<unsigned long long>());
    }
    if (hipMemcpy(red, d_red.p, 32, hipMemcpyDeviceToHost) != hipSuccess) { set_error("range read-back failed"); return fail(QLDPC_ERR_HIP); }
    finite[0] = (int64_t)red[2]; finite[1] = (int64_t)red[3];
    const bool any = (red[2] | red[3]) != 0;
    range[0] = any ? key_f64(red[0]) : 0.0;
    range[1] = any ? key_f64(red[1]) : 0.0;
    *out = S;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_msgstats_histogram(qldpc_msgstats *S, const double *edges, int bins, int64_t *hist0, int64_t *hist1) {
    QLDPC_REQUIRE(S != nullptr && edges != nullptr && hist0 != nullptr && hist1 != nullptr, "NULL argument");
    QLDPC_REQUIRE(bins >= 1 && bins <= kStatsMaxBins, "bins must be in [1, %d]", kStatsMaxBins);
    for (int i = 0; i < bins; i++) QLDPC_REQUIRE(edges[i] < edges[i + 1], "bin edges must increase strictly");
    QLDPC_USE_DEVICE(S->g->device);
    int rc = QLDPC_OK; (void)rc;
    DevTmp d_edges, d_hist;
    if ((rc = d_edges.alloc((bins + 1) * 8)) || (rc = d_hist.alloc(2 * bins * 8))) return rc;
    QLDPC_HIP_TRY(hipMemcpy(d_edges.p, edges, (bins + 1) * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(zero_now(d_hist.p, 2 * bins * 8));
    const int64_t total = S->B * (int64_t)S->L;
    if (total > 0) {
        // each block counts into 32-bit LDS bins: keep its share of the samples below 2^31
        int64_t grid = std::min<int64_t>(4096, (total + 255) / 256);
        while (total / grid >= ((int64_t)1 << 31)) grid *= 2;
        const size_t lds = (size_t)(bins + 1) * 8 + (size_t)2 * bins * 4;
        hipLaunchKernelGGL(stats_hist_kernel, dim3((unsigned)grid), dim3(256), lds, nullptr, S->B, S->L, S->g->n,
                           S->by_edge ? S->g->d_indices : (const int32_t *)nullptr, S->d_samples.as<double>(), S->d_err.as<int8_t>(),
                           d_edges.as<double>(), bins, d_hist.as<unsigned long long>());
        QLDPC_HIP_TRY(hipGetLastError());
    }
    std::vector<unsigned long long> h(2 * (size_t)bins);
    QLDPC_HIP_TRY(hipMemcpy(h.data(), d_hist.p, 2 * (size_t)bins * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < bins; i++) { hist0[i] = (int64_t)h[i]; hist1[i] = (int64_t)h[bins + i]; }
    return QLDPC_OK;
}
