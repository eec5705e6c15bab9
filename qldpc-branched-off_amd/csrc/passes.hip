// Single check-node passes on the CSR edge layout (a3 minsum_core_sparse, a5 bp_core) and the sum-product
// driver performBeliefPropagationFast (a5).  Edge messages are laid out [shot][edge] as in the reference call.
#include "common.h"
#include "mc_common.h"

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
    int rc = use_device(g->device);
    if (rc != QLDPC_OK) return rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(Q && ssign && R && Rsum, "NULL buffer");
    const size_t m = g->m, n = g->n, nnz = g->nnz;
    DevTmp dQ, dS, dR, dRs;
    if ((rc = dQ.alloc(B * nnz * 8)) || (rc = dS.alloc(B * m * 8)) || (rc = dR.alloc(B * nnz * 8)) || (rc = dRs.alloc(B * n * 8))) return rc;
    if (nnz) QLDPC_HIP_TRY(hipMemcpy(dQ.p, Q, B * nnz * 8, hipMemcpyHostToDevice));
    if (m) QLDPC_HIP_TRY(hipMemcpy(dS.p, ssign, B * m * 8, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemset(dR.p, 0, B * nnz * 8 + (nnz ? 0 : 16)));
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
    int rc = use_device(g->device);
    if (rc != QLDPC_OK) return rc;
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
    QLDPC_HIP_TRY(hipMemset(dR.p, 0, B * nnz * 8 + (nnz ? 0 : 16)));
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
