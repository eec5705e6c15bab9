// HBM-streaming min-sum decoder: one lane per shot, messages laid out [edge][shot] so that every
// wave access is one contiguous 512-byte line across the batch dimension.  Works for any graph size
// (the circuit-level graphs of BASELINE config 5 do not fit in LDS); the LDS/register-resident kernel in
// minsum_resident.hip is the fast path for small code-capacity graphs.
//
// Algorithm = minsum_decoder_full (reference src/decoding/kernels.py:234-366) per lane, re-associated only
// where it is exact: the variable update of iteration k (kernels.py:323-345) is evaluated lazily inside the
// check pass of iteration k+1, so one message array is read and written once per iteration
// (16*nnz bytes/shot/iteration, the algorithmic figure of SURVEY 8d).  Floating-point expressions keep the
// reference's operand order; the file is compiled with -ffp-contract=off.
#include "common.h"
#include "minsum_common.h"

namespace qldpc {

struct StreamArgs {
    int m, n, max_iter, fixed, nfcheck;
    const int32_t *indptr, *indices;
    int64_t B, Bpad;
    const int8_t *synd;     // [B][m]
    const double *prior;    // [n]
    const double *alpha;    // [max_iter]
    double damping, clip;
    double *M, *Qold, *V0, *V1;  // [nnz][Bpad], [nnz][Bpad] (damping only), [n][Bpad] x2
    int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
};

template <bool DAMP>
__global__ __launch_bounds__(256) void minsum_stream_kernel(StreamArgs A) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < A.B;
    const int64_t S = A.Bpad;
    double *__restrict__ M = A.M + b;
    double *__restrict__ Qo = DAMP ? A.Qold + b : nullptr;
    double *vcur = A.V0 + b, *vnext = A.V1 + b;
    const int8_t *__restrict__ synd = A.synd + (valid ? b : 0) * A.m;
    const int m = A.m, n = A.n;
    const double clip = A.clip, damping = A.damping, one_minus_d = 1.0 - A.damping;

    bool done = !valid;       // outputs frozen
    for (int it = 0; it < A.max_iter; it++) {
        const bool run = A.fixed ? valid : !done;
        if (!A.fixed && !__any(run)) break;
        const double alpha = A.alpha[it];
        if (run) {
            for (int j = 0; j < n; j++) vnext[(int64_t)j * S] = 0.0;                       // kernels.py:278-279
            for (int i = 0; i < m; i++) {                                                    // kernels.py:282-316
                const int rs = A.indptr[i], re = A.indptr[i + 1];
                if (rs == re) continue;
                double sign_prod = 1.0 - 2.0 * (double)synd[i];                              // kernels.py:252,289
                double min1 = INFINITY, min2 = INFINITY;
                int min1_pos = -1;
                for (int pos = rs; pos < re; pos++) {
                    const int col = A.indices[pos];
                    double q;
                    if (it == 0) {
                        q = A.prior[col];                                                    // kernels.py:263-265 (not clipped)
                    } else {
                        q = clip_nan(vcur[(int64_t)col * S] - M[(int64_t)pos * S], clip);    // kernels.py:325-333
                        if (!DAMP && A.nfcheck && prior_not_finite(A.prior[col])) q = NAN;   // kernels.py:336 with Q_old = +-inf (see minsum_common.h)
                        if (DAMP) {
                            q = damping * q + one_minus_d * Qo[(int64_t)pos * S];           // kernels.py:336
                            q = clip_only(q, clip);                                          // kernels.py:339-342
                        }
                    }
                    if (DAMP) Qo[(int64_t)pos * S] = q;
                    if (!(q >= 0)) sign_prod = -sign_prod;                                   // sign_prod *= -1.0 (exact)
                    const double a = fabs(q);
                    if (a < min1) { min2 = min1; min1 = a; min1_pos = pos; }
                    else if (a < min2) { min2 = a; }
                }
                for (int pos = rs; pos < re; pos++) {
                    const int col = A.indices[pos];
                    double q;
                    if (DAMP) q = Qo[(int64_t)pos * S];
                    else if (it == 0) q = A.prior[col];
                    else {
                        q = clip_nan(vcur[(int64_t)col * S] - M[(int64_t)pos * S], clip);
                        if (A.nfcheck && prior_not_finite(A.prior[col])) q = NAN;
                    }
                    const double sign_j = (q >= 0) ? 1.0 : -1.0;
                    const double mag = (pos == min1_pos) ? min2 : min1;
                    const double msg = alpha * (sign_prod * sign_j) * mag;                   // kernels.py:312-314
                    M[(int64_t)pos * S] = msg;
                    vnext[(int64_t)col * S] += msg;                                          // kernels.py:316 (row order)
                }
            }
            for (int j = 0; j < n; j++) vnext[(int64_t)j * S] = vnext[(int64_t)j * S] + A.prior[j];   // kernels.py:319-320
        }
        double *t = vcur; vcur = vnext; vnext = t;
        if (run && !done) {
            bool ok = true;                                                                  // kernels.py:348-359
            for (int i = 0; i < m; i++) {
                int s = 0;
                for (int pos = A.indptr[i]; pos < A.indptr[i + 1]; pos++) s ^= (vcur[(int64_t)A.indices[pos] * S] < 0) ? 1 : 0;
                ok = ok && (s == (int)synd[i]);
            }
            if (ok) {                                                                        // kernels.py:361-364
                done = true;
                A.out_conv[b] = 1; A.out_iter[b] = it;
                for (int j = 0; j < n; j++) {
                    const double v = vcur[(int64_t)j * S];
                    A.out_llr[b * n + j] = v; A.out_err[b * n + j] = (v < 0) ? 1 : 0;
                }
            }
        }
    }
    if (valid && !done) {                                                                    // kernels.py:267,366
        A.out_conv[b] = 0; A.out_iter[b] = A.max_iter - 1;
        for (int j = 0; j < n; j++) {
            const double v = (A.max_iter > 0) ? vcur[(int64_t)j * S] : 0.0;
            A.out_llr[b * n + j] = v; A.out_err[b * n + j] = (v < 0) ? 1 : 0;
        }
    }
}

int minsum_stream_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                         const double *d_alpha, double damping, double clip, int flags, int8_t *d_err, double *d_llr,
                         uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    const bool damp = (damping != 1.0);
    const size_t per_shot = (size_t)g->nnz * 8 * (damp ? 2 : 1) + (size_t)g->n * 16;
    const size_t budget = (size_t)6 << 30;
    int64_t chunk = (int64_t)(budget / (per_shot ? per_shot : 1));
    chunk = chunk / 256 * 256;
    if (chunk < 256) chunk = 256;
    if (chunk > round_up(B, 256)) chunk = round_up(B, 256);
    int rc;
    if ((rc = g->ws_msg.ensure((size_t)g->nnz * 8 * chunk)) != QLDPC_OK) return rc;
    if (damp && (rc = g->ws_qold.ensure((size_t)g->nnz * 8 * chunk)) != QLDPC_OK) return rc;
    if ((rc = g->ws_vals.ensure((size_t)g->n * 16 * chunk)) != QLDPC_OK) return rc;
    for (int64_t off = 0; off < B; off += chunk) {
        const int64_t cnt = (B - off < chunk) ? (B - off) : chunk;
        StreamArgs A;
        A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
        A.nfcheck = (flags & QLDPC_FLAG_INTERNAL_PRIOR_FINITE) ? 0 : 1;
        A.indptr = g->d_indptr; A.indices = g->d_indices;
        A.B = cnt; A.Bpad = chunk;
        A.synd = d_synd + off * g->m; A.prior = d_prior; A.alpha = d_alpha;
        A.damping = damping; A.clip = clip;
        A.M = g->ws_msg.as<double>(); A.Qold = damp ? g->ws_qold.as<double>() : nullptr;
        A.V0 = g->ws_vals.as<double>(); A.V1 = A.V0 + (size_t)g->n * chunk;
        A.out_err = d_err + off * g->n; A.out_llr = d_llr + off * g->n; A.out_conv = d_conv + off; A.out_iter = d_iter + off;
        const unsigned grid = (unsigned)(round_up(cnt, 256) / 256);
        if (damp) hipLaunchKernelGGL(minsum_stream_kernel<true>, dim3(grid), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL(minsum_stream_kernel<false>, dim3(grid), dim3(256), 0, stream, A);
        QLDPC_HIP_TRY(hipGetLastError());
    }
    return QLDPC_OK;
}

}  // namespace qldpc
