/*
 * qldpc_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.
 *
 * A plain-C, strict IEEE-754 restatement of the reference's hot-path algorithms
 * (michelebanfi/qLDPC-branched-off, /root/reference).  Each function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the shipped package (qldpc-branched-off_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py)
 * against the tests/golden/ fixtures, which were produced by executing the reference's own Python
 * source in the build container (tests/golden/make_golden.py; numba is absent there, so
 * the source runs under CPython with an identity @njit = the strict-IEEE reading; the
 * reference's fastmath=True may only differ in NaN tests/FMA contraction).
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * The Monte-Carlo sampler at the bottom (Philox4x32-10 streams) is this project's own
 * definition of the synthetic workload (the reference uses legacy np.random, which cannot
 * be reproduced on a GPU); its sampling LAW follows src/decoding/alpha.py:127-128.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* alpha modes (src/decoding/sparse.py:18-29 decides which one applies) */
enum { ORC_ALPHA_CONST = 0, ORC_ALPHA_DYNAMIC = 1, ORC_ALPHA_SEQ = 2 };

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * a1 / a2: minsum_decoder_full (src/decoding/kernels.py:234-366) and
 * minsum_decoder_full_autoregressive (kernels.py:369-485).  One syndrome.
 * work: Q[nnz], Qold[nnz], R[nnz], Rsum[n] doubles (caller provided or NULL -> malloc).
 * Returns final_iter (kernels.py:267,362); *converged set.
 * ---------------------------------------------------------------------------------- */
static double alpha_at(int mode, int k, double alpha_val, const double *seq, int seq_len) {
    if (mode == ORC_ALPHA_DYNAMIC) return 1.0 - ldexp(1.0, -(k + 1)); /* kernels.py:273: 1.0 - 2.0**(-(k+1)) */
    if (mode == ORC_ALPHA_SEQ) return (k < seq_len) ? seq[k] : seq[seq_len - 1]; /* kernels.py:402-405 */
    return alpha_val;                                                  /* kernels.py:275 */
}

ORC_API int orc_minsum_decode(int m, int n, const int32_t *indptr, const int32_t *indices,
                              const int8_t *syndrome, const double *prior, int max_iter,
                              int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                              double damping, double clip,
                              int8_t *cand, double *values, uint8_t *converged, double *work) {
    const int nnz = indptr[m];
    double *own = NULL;
    if (!work) { own = (double *)malloc(sizeof(double) * (size_t)(3 * nnz + n + m + 1)); work = own; }
    double *Q = work, *Qold = Q + nnz, *R = Qold + nnz, *Rsum = R + nnz, *ssign = Rsum + n;
    for (int i = 0; i < m; i++) ssign[i] = 1.0 - 2.0 * (double)syndrome[i];          /* :250-252 */
    for (int j = 0; j < n; j++) { cand[j] = 0; Rsum[j] = 0.0; }                        /* :258,260 */
    for (int e = 0; e < nnz; e++) { Q[e] = prior[indices[e]]; Qold[e] = Q[e]; }        /* :263-265 */
    int final_iter = max_iter - 1;                                                      /* :267 */
    *converged = 0;
    for (int it = 0; it < max_iter; it++) {
        const double alpha = alpha_at(alpha_mode, it, alpha_val, alpha_seq, alpha_len); /* :272-275 */
        for (int j = 0; j < n; j++) Rsum[j] = 0.0;                                      /* :278-279 */
        for (int i = 0; i < m; i++) {                                                   /* :282-316 */
            const int rs = indptr[i], re = indptr[i + 1];
            if (rs == re) continue;
            double sign_prod = ssign[i], min1 = INFINITY, min2 = INFINITY;
            int min1_pos = -1;
            for (int pos = rs; pos < re; pos++) {
                const double val = Q[pos];
                if (val >= 0) sign_prod *= 1.0; else sign_prod *= -1.0;
                const double a = fabs(val);
                if (a < min1) { min2 = min1; min1 = a; min1_pos = pos; }
                else if (a < min2) { min2 = a; }
            }
            for (int pos = rs; pos < re; pos++) {
                const double val = Q[pos];
                const double sign_j = (val >= 0) ? 1.0 : -1.0;
                const double row_sign_excl_j = sign_prod * sign_j;
                const double mag = (pos == min1_pos) ? min2 : min1;
                const double msg = alpha * row_sign_excl_j * mag;      /* (alpha*sign)*mag, left to right */
                R[pos] = msg;
                Rsum[indices[pos]] += msg;                              /* scatter-add in row order */
            }
        }
        for (int j = 0; j < n; j++) values[j] = Rsum[j] + prior[j];                     /* :319-320 */
        for (int e = 0; e < nnz; e++) {                                                 /* :323-345 */
            double q = values[indices[e]] - R[e];
            if (q != q) q = 0.0; else if (q > clip) q = clip; else if (q < -clip) q = -clip;
            double qd = damping * q + (1.0 - damping) * Qold[e];
            if (qd > clip) qd = clip; else if (qd < -clip) qd = -clip;
            Q[e] = qd; Qold[e] = qd;
        }
        for (int j = 0; j < n; j++) cand[j] = (values[j] < 0) ? 1 : 0;                  /* :348-349 */
        int ok = 1;                                                                      /* :352-359 */
        for (int i = 0; i < m; i++) {
            int s = 0;
            for (int e = indptr[i]; e < indptr[i + 1]; e++) s ^= cand[indices[e]];
            if (s != syndrome[i]) { ok = 0; break; }
        }
        if (ok) { final_iter = it; *converged = 1; break; }                             /* :361-364 */
    }
    free(own);
    return final_iter;
}

/* batch driver (shots are independent; OpenMP over shots is the timed CPU baseline) */
ORC_API void orc_minsum_decode_batch(int m, int n, const int32_t *indptr, const int32_t *indices, int64_t B,
                                     const int8_t *syndromes, const double *prior, int max_iter,
                                     int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                                     double damping, double clip,
                                     int8_t *cand, double *values, uint8_t *converged, int32_t *iters, int threads) {
    const int nnz = indptr[m];
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        double *work = (double *)malloc(sizeof(double) * (size_t)(3 * nnz + n + m + 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t b = 0; b < B; b++)
            iters[b] = orc_minsum_decode(m, n, indptr, indices, syndromes + b * m, prior, max_iter, alpha_mode,
                                         alpha_val, alpha_seq, alpha_len, damping, clip, cand + b * n,
                                         values + b * n, converged + b, work);
        free(work);
    }
}

/* a3: minsum_core_sparse (kernels.py:138-169): one check-node pass on CSR */
ORC_API void orc_minsum_core_sparse(int m, int n, const int32_t *indptr, const int32_t *indices,
                                    const double *Q, const double *ssign, double alpha, double *R, double *Rsum) {
    const int nnz = indptr[m];
    for (int e = 0; e < nnz; e++) R[e] = 0.0;
    for (int j = 0; j < n; j++) Rsum[j] = 0.0;
    for (int i = 0; i < m; i++) {
        const int rs = indptr[i], re = indptr[i + 1];
        if (rs == re) continue;
        double sign_prod = ssign[i], min1 = INFINITY, min2 = INFINITY;
        int min1_pos = -1;
        for (int pos = rs; pos < re; pos++) {
            const double val = Q[pos];
            sign_prod *= (val >= 0) ? 1.0 : -1.0;
            const double a = fabs(val);
            if (a < min1) { min2 = min1; min1 = a; min1_pos = pos; }
            else if (a < min2) { min2 = a; }
        }
        for (int pos = rs; pos < re; pos++) {
            const double val = Q[pos];
            const double sign_j = (val >= 0) ? 1.0 : -1.0;
            const double mag = (pos == min1_pos) ? min2 : min1;
            const double msg = alpha * (sign_prod * sign_j) * mag;
            R[pos] = msg;
            Rsum[indices[pos]] += msg;
        }
    }
}

/* f4: the message update shared by both estimators (alpha.py:226-244, scopt.py:101-118) */
static void orc_estimator_q_update(int nnz, const int32_t *indices, const double *values, const double *R, double damping,
                                   double clip, double *Q, double *Qold) {
    for (int pos = 0; pos < nnz; pos++) {
        double q_new = values[indices[pos]] - R[pos];
        if (q_new != q_new) q_new = 0.0;
        else if (q_new > clip) q_new = clip;
        else if (q_new < -clip) q_new = -clip;
        double q_damped = damping * q_new + (1.0 - damping) * Qold[pos];
        if (q_damped > clip) q_damped = clip;
        else if (q_damped < -clip) q_damped = -clip;
        Q[pos] = q_damped;
        Qold[pos] = q_damped;
    }
}

/* f4: trial body of estimate_alpha_alvarado (alpha.py:119-137, n_prev = 0) and of the autoregressive estimator
 * (alpha.py:206-255): unscaled check messages R_flat[B][nnz] after n_prev iterations with the given alphas.
 * errors: int8[B][n]; scratch is allocated here. */
ORC_API void orc_alpha_messages(int m, int n, const int32_t *indptr, const int32_t *indices, int64_t B, const int8_t *errors,
                                const double *prior, const double *alpha_prev, int n_prev, double damping, double clip,
                                double *R_out) {
    const int nnz = indptr[m];
    double *Q = (double *)malloc(sizeof(double) * (nnz + 1)), *Qold = (double *)malloc(sizeof(double) * (nnz + 1));
    double *R = (double *)malloc(sizeof(double) * (nnz + 1)), *Rsum = (double *)malloc(sizeof(double) * (n + 1));
    double *ssign = (double *)malloc(sizeof(double) * (m + 1)), *values = (double *)malloc(sizeof(double) * (n + 1));
    for (int64_t b = 0; b < B; b++) {
        const int8_t *e = errors + b * n;
        for (int i = 0; i < m; i++) {
            int s = 0;
            for (int pos = indptr[i]; pos < indptr[i + 1]; pos++) s ^= e[indices[pos]] & 1;
            ssign[i] = 1.0 - 2.0 * (double)s;                                              /* alpha.py:121-122 */
        }
        for (int pos = 0; pos < nnz; pos++) { Q[pos] = prior[indices[pos]]; Qold[pos] = Q[pos]; }
        for (int k = 0; k < n_prev; k++) {
            orc_minsum_core_sparse(m, n, indptr, indices, Q, ssign, alpha_prev[k], R, Rsum);
            for (int j = 0; j < n; j++) values[j] = Rsum[j] + prior[j];
            orc_estimator_q_update(nnz, indices, values, R, damping, clip, Q, Qold);
        }
        orc_minsum_core_sparse(m, n, indptr, indices, Q, ssign, 1.0, R_out + b * nnz, Rsum);
    }
    free(Q); free(Qold); free(R); free(Rsum); free(ssign); free(values);
}

/* f4: trial body of estimate_scopt_beta (scopt.py:80-131): posterior values[B][n] at the iteration the loop leaves */
ORC_API void orc_scopt_values(int m, int n, const int32_t *indptr, const int32_t *indices, int64_t B, const int8_t *errors,
                              const double *prior, int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq,
                              int alpha_len, double damping, double clip, double *values_out) {
    const int nnz = indptr[m];
    double *Q = (double *)malloc(sizeof(double) * (nnz + 1)), *Qold = (double *)malloc(sizeof(double) * (nnz + 1));
    double *R = (double *)malloc(sizeof(double) * (nnz + 1)), *Rsum = (double *)malloc(sizeof(double) * (n + 1));
    double *ssign = (double *)malloc(sizeof(double) * (m + 1));
    int8_t *synd = (int8_t *)malloc(m + 1);
    for (int64_t b = 0; b < B; b++) {
        const int8_t *e = errors + b * n;
        double *values = values_out + b * n;
        for (int i = 0; i < m; i++) {
            int s = 0;
            for (int pos = indptr[i]; pos < indptr[i + 1]; pos++) s ^= e[indices[pos]] & 1;
            synd[i] = (int8_t)s;
            ssign[i] = 1.0 - 2.0 * (double)s;
        }
        for (int pos = 0; pos < nnz; pos++) { Q[pos] = prior[indices[pos]]; Qold[pos] = Q[pos]; }
        for (int j = 0; j < n; j++) values[j] = 0.0;
        for (int it = 0; it < max_iter; it++) {
            const double a = alpha_at(alpha_mode, it, alpha_val, alpha_seq, alpha_len);   /* scopt.py:89-94 */
            orc_minsum_core_sparse(m, n, indptr, indices, Q, ssign, a, R, Rsum);
            for (int j = 0; j < n; j++) values[j] = Rsum[j] + prior[j];
            orc_estimator_q_update(nnz, indices, values, R, damping, clip, Q, Qold);
            int ok = 1;                                                                   /* scopt.py:127-130 */
            for (int i = 0; i < m && ok; i++) {
                int s = 0;
                for (int pos = indptr[i]; pos < indptr[i + 1]; pos++) s ^= (values[indices[pos]] < 0) ? 1 : 0;
                ok = (s == synd[i]);
            }
            if (ok) break;
        }
    }
    free(Q); free(Qold); free(R); free(Rsum); free(ssign); free(synd);
}

/* a4: minsum_core (kernels.py:108-136): dense-mask twin.  H is unused by the reference body. */
ORC_API void orc_minsum_core_dense(int m, int n, const double *Q, const double *ssign, const uint8_t *mask,
                                   double alpha, double *R) {
    for (int i = 0; i < m; i++) {
        double sign_prod = ssign[i], min1 = INFINITY, min2 = INFINITY;
        int min1_idx = -1;
        for (int j = 0; j < n; j++) {
            R[(size_t)i * n + j] = 0.0;
            if (mask[(size_t)i * n + j]) {
                const double val = Q[(size_t)i * n + j];
                sign_prod *= (val >= 0) ? 1.0 : -1.0;
                const double a = fabs(val);
                if (a < min1) { min2 = min1; min1 = a; min1_idx = j; }
                else if (a < min2) { min2 = a; }
            }
        }
        for (int j = 0; j < n; j++) {
            if (mask[(size_t)i * n + j]) {
                const double val = Q[(size_t)i * n + j];
                const double sign_j = (val >= 0) ? 1.0 : -1.0;
                const double mag = (j == min1_idx) ? min2 : min1;
                R[(size_t)i * n + j] = alpha * (sign_prod * sign_j) * mag;
            }
        }
    }
}

/* a5: bp_core (kernels.py:171-193): tanh-product check update on a dense mask */
ORC_API void orc_bp_core_dense(int m, int n, const double *Q, const double *ssign, const uint8_t *mask,
                               double clip_val, double *R) {
    for (int i = 0; i < m; i++) {
        double row_prod = 1.0;
        for (int j = 0; j < n; j++) {
            R[(size_t)i * n + j] = 0.0;
            if (mask[(size_t)i * n + j]) {
                double t = tanh(Q[(size_t)i * n + j] * 0.5);
                if (fabs(t) < 1e-15) t = (t >= 0) ? 1e-15 : -1e-15;
                row_prod *= t;
            }
        }
        for (int j = 0; j < n; j++) {
            if (mask[(size_t)i * n + j]) {
                double t = tanh(Q[(size_t)i * n + j] * 0.5);
                if (fabs(t) < 1e-15) t = (t >= 0) ? 1e-15 : -1e-15;
                double pc = (row_prod / t) * ssign[i];
                if (pc < -clip_val) pc = -clip_val; else if (pc > clip_val) pc = clip_val;  /* np.clip */
                R[(size_t)i * n + j] = 2.0 * atanh(pc);
            }
        }
    }
}

/* a4 driver: performMinSum_Symmetric (src/decoding/dense.py:5-73).  alpha_estimation=1 returns the
 * unscaled iteration-0 messages in Rest[m*n] (dense.py:54-56) and final_iter 0. */
ORC_API int orc_minsum_dense_driver(int m, int n, const uint8_t *mask, const int8_t *syndrome, const double *prior,
                                    int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq,
                                    int alpha_len, double damping, double clip, int alpha_estimation,
                                    int8_t *cand, double *values, uint8_t *converged, double *Rest) {
    const size_t mn = (size_t)m * n;
    double *Q = (double *)malloc(sizeof(double) * (3 * mn + (size_t)m));
    double *Qold = Q + mn, *R = Qold + mn, *ssign = R + mn;
    for (int i = 0; i < m; i++) ssign[i] = (double)(1 - 2 * (int)syndrome[i]);
    for (size_t t = 0; t < mn; t++) { Q[t] = mask[t] ? prior[t % n] : 0.0; Qold[t] = Q[t]; }
    for (int j = 0; j < n; j++) { cand[j] = 0; values[j] = 0.0; }
    *converged = 0;
    int it = 0, last = 0;
    for (it = 0; it < max_iter; it++) {
        last = it;
        const double alpha = alpha_at(alpha_mode, it, alpha_val, alpha_seq, alpha_len);   /* dense.py:48-51 */
        orc_minsum_core_dense(m, n, Q, ssign, mask, alpha, R);
        if (alpha_estimation && it == 0) {                                                /* dense.py:54-56 */
            const double scale = (alpha != 0) ? alpha : 1.0;
            for (size_t t = 0; t < mn; t++) Rest[t] = R[t] / scale;
            for (int j = 0; j < n; j++) cand[j] = 0;
            free(Q);
            return 0;
        }
        for (int j = 0; j < n; j++) {                                   /* np.sum(R,axis=0): row-sequential */
            double s = R[j];
            for (int i = 1; i < m; i++) s += R[(size_t)i * n + j];
            values[j] = s + prior[j];
        }
        for (size_t t = 0; t < mn; t++) {                               /* dense.py:60-65 */
            double q = mask[t] ? (values[t % n] - R[t]) : 0.0;
            if (q != q) q = 0.0; else if (q == INFINITY) q = clip; else if (q == -INFINITY) q = -clip;
            double qd = damping * q + (1 - damping) * Qold[t];
            if (qd < -clip) qd = -clip; else if (qd > clip) qd = clip;
            Q[t] = qd; Qold[t] = qd;
        }
        int ok = 1;
        for (int j = 0; j < n; j++) cand[j] = (values[j] < 0) ? 1 : 0;
        for (int i = 0; i < m && ok; i++) {
            int s = 0;
            for (int j = 0; j < n; j++) if (mask[(size_t)i * n + j]) s ^= cand[j];
            if (s != syndrome[i]) ok = 0;
        }
        if (ok && !alpha_estimation) { *converged = 1; free(Q); return it; }              /* dense.py:70-71 */
    }
    free(Q);
    return last;                                                                           /* dense.py:73 */
}

/* a5 driver: performBeliefPropagationFast (dense.py:75-96): no clip / damping / NaN handling */
ORC_API int orc_bp_dense_driver(int m, int n, const uint8_t *mask, const int8_t *syndrome, const double *prior,
                                int max_iter, int8_t *cand, double *values, uint8_t *converged) {
    const size_t mn = (size_t)m * n;
    double *Q = (double *)malloc(sizeof(double) * (2 * mn + (size_t)m));
    double *R = Q + mn, *ssign = R + mn;
    for (int i = 0; i < m; i++) ssign[i] = (double)(1 - 2 * (int)syndrome[i]);
    for (size_t t = 0; t < mn; t++) Q[t] = mask[t] ? prior[t % n] : 0.0;
    *converged = 0;
    int last = 0;
    for (int it = 0; it < max_iter; it++) {
        last = it;
        orc_bp_core_dense(m, n, Q, ssign, mask, 0.9999999, R);
        for (int j = 0; j < n; j++) {
            double s = R[j];
            for (int i = 1; i < m; i++) s += R[(size_t)i * n + j];
            values[j] = s + prior[j];
        }
        for (size_t t = 0; t < mn; t++) Q[t] = mask[t] ? (values[t % n] - R[t]) : 0.0;
        int ok = 1;
        for (int j = 0; j < n; j++) cand[j] = (values[j] < 0) ? 1 : 0;
        for (int i = 0; i < m && ok; i++) {
            int s = 0;
            for (int j = 0; j < n; j++) if (mask[(size_t)i * n + j]) s ^= cand[j];
            if (s != syndrome[i]) ok = 0;
        }
        if (ok) { *converged = 1; free(Q); return it; }
    }
    free(Q);
    return last;
}

/* a6: syndrome_check (kernels.py:222-231) */
ORC_API void orc_syndrome_check(int m, const int32_t *indptr, const int32_t *indices, const int8_t *cand, int8_t *out) {
    for (int i = 0; i < m; i++) {
        int s = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; e++) s ^= cand[indices[e]];
        out[i] = (int8_t)s;
    }
}

/* a7: gf2_elimination (kernels.py:5-34) on a byte matrix (one 0/1 element per byte), in place */
ORC_API int orc_gf2_elimination(int m, int n, uint8_t *A, uint8_t *b, int64_t *pivot_rows, int64_t *pivot_cols) {
    int np_ = 0, row = 0;
    for (int col = 0; col < n; col++) {
        if (row >= m) break;
        int pr = -1;
        for (int r = row; r < m; r++) if (A[(size_t)r * n + col] == 1) { pr = r; break; }
        if (pr == -1) continue;
        if (pr != row) {
            for (int j = 0; j < n; j++) { uint8_t t = A[(size_t)row * n + j]; A[(size_t)row * n + j] = A[(size_t)pr * n + j]; A[(size_t)pr * n + j] = t; }
            uint8_t t = b[row]; b[row] = b[pr]; b[pr] = t;
        }
        pivot_rows[np_] = row; pivot_cols[np_] = col; np_++;
        for (int r = 0; r < m; r++)
            if (r != row && A[(size_t)r * n + col] == 1) {
                for (int j = 0; j < n; j++) A[(size_t)r * n + j] ^= A[(size_t)row * n + j];
                b[r] ^= b[row];
            }
        row++;
    }
    return np_;
}

/* a8: _pack_rows_uint64 (kernels.py:36-46): little-endian bit order, rows padded to 8 bytes */
ORC_API int orc_packed_words(int n) { return ((n + 7) / 8 + 7) / 8; }
ORC_API void orc_pack_rows_u64(int m, int n, const uint8_t *A, uint64_t *P) {
    const int nw = orc_packed_words(n);
    memset(P, 0, sizeof(uint64_t) * (size_t)m * nw);
    for (int r = 0; r < m; r++)
        for (int c = 0; c < n; c++)
            if (A[(size_t)r * n + c]) P[(size_t)r * nw + (c >> 6)] |= (uint64_t)1 << (c & 63);
}

/* a8: gf2_elimination_packed_core (kernels.py:48-96) */
ORC_API int orc_gf2_elimination_packed(int m, int n, int nwords, uint64_t *A, uint8_t *b, int64_t *pivot_rows, int64_t *pivot_cols) {
    int np_ = 0, row = 0;
    for (int col = 0; col < n; col++) {
        if (row >= m) break;
        const int w = col >> 6;
        const uint64_t bit = (uint64_t)1 << (col & 63);
        int pr = -1;
        for (int r = row; r < m; r++) if (A[(size_t)r * nwords + w] & bit) { pr = r; break; }
        if (pr == -1) continue;
        if (pr != row) {
            for (int k = 0; k < nwords; k++) { uint64_t t = A[(size_t)row * nwords + k]; A[(size_t)row * nwords + k] = A[(size_t)pr * nwords + k]; A[(size_t)pr * nwords + k] = t; }
            uint8_t t = b[row]; b[row] = b[pr]; b[pr] = t;
        }
        pivot_rows[np_] = row; pivot_cols[np_] = col; np_++;
        for (int r = 0; r < m; r++)
            if (r != row && (A[(size_t)r * nwords + w] & bit)) {
                for (int k = 0; k < nwords; k++) A[(size_t)r * nwords + k] ^= A[(size_t)row * nwords + k];
                b[r] ^= b[row];
            }
        row++;
    }
    return np_;
}

/* stable argsort of |llr| ascending, ties by ascending index.  osd.py:11-12 uses np.argsort's
 * default kind, whose tie order is implementation defined; callers that need the reference's
 * exact tie order pass `ordering` explicitly (tests do, from the golden capture). */
typedef struct { double key; int32_t idx; } orc_kv;
static int orc_kv_cmp(const void *a, const void *b) {
    const orc_kv *x = (const orc_kv *)a, *y = (const orc_kv *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
ORC_API void orc_argsort_abs(int n, const double *llr, int32_t *ordering) {
    orc_kv *kv = (orc_kv *)malloc(sizeof(orc_kv) * (size_t)n);
    for (int j = 0; j < n; j++) { double a = fabs(llr[j]); kv[j].key = (a != a) ? INFINITY : a; kv[j].idx = j; }
    qsort(kv, (size_t)n, sizeof(orc_kv), orc_kv_cmp);
    for (int j = 0; j < n; j++) ordering[j] = kv[j].idx;
    free(kv);
}

/* a9: performOSD_enhanced with order == 0 (osd.py:5-29).  H given as CSR.  `ordering` may be NULL
 * (-> stable argsort above).  solution[n] = (hard + e_correction) % 2. */
ORC_API void orc_osd0(int m, int n, const int32_t *indptr, const int32_t *indices, const int8_t *syndrome,
                      const double *llr, const int8_t *hard, const int32_t *ordering_in, int8_t *solution) {
    const int nw = orc_packed_words(n);
    int32_t *ordering = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (ordering_in) memcpy(ordering, ordering_in, sizeof(int32_t) * (size_t)n); else orc_argsort_abs(n, llr, ordering);
    for (int c = 0; c < n; c++) inv[ordering[c]] = c;
    uint64_t *A = (uint64_t *)calloc((size_t)m * nw, sizeof(uint64_t));
    uint8_t *b = (uint8_t *)malloc((size_t)m);
    for (int i = 0; i < m; i++) {                      /* residual syndrome osd.py:8-9 ; H[:, ordering] osd.py:13 */
        int s = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; e++) {
            const int j = indices[e];
            s ^= hard[j];
            const int c = inv[j];
            A[(size_t)i * nw + (c >> 6)] |= (uint64_t)1 << (c & 63);
        }
        b[i] = (uint8_t)((syndrome[i] + s) & 1);
    }
    const int maxp = m < n ? m : n;
    int64_t *pr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * maxp + 2)), *pc = pr + maxp + 1;
    const int np_ = orc_gf2_elimination_packed(m, n, nw, A, b, pr, pc);                /* osd.py:15-17 */
    for (int j = 0; j < n; j++) solution[j] = hard[j];
    for (int t = 0; t < np_; t++) {                                                     /* osd.py:19-25 */
        const int j = ordering[pc[t]];
        solution[j] = (int8_t)((hard[j] + b[pr[t]]) & 1);
    }
    free(pr); free(b); free(A); free(inv); free(ordering);
}

/* f1: performOSD_enhanced with order > 0 (osd.py:5-77), literal.  H as CSR.  `ordering` may be NULL (stable argsort of |llr|);
 * the second sort (osd.py:37-38, the non-pivot positions by |llr|) is stable with ties by ascending permuted position.
 * max_combinations <= 0 means "no limit" (Python None / 0).  solution[n]. */
static double orc_osd_metric(int n, const int8_t *sol, const double *llr_abs, int syndrome_weight) {
    double metric = (syndrome_weight > 0) ? 1e10 + syndrome_weight * 1e8 : 0.0;          /* kernels.py:197-200 */
    for (int i = 0; i < n; i++) metric += (double)sol[i] * llr_abs[i];                   /* kernels.py:201-202 */
    return metric;
}

ORC_API void orc_osdw(int m, int n, const int32_t *indptr, const int32_t *indices, const int8_t *syndrome, const double *llr,
                      const int8_t *hard, const int32_t *ordering_in, int order, int64_t max_combinations, int8_t *solution) {
    const int nw = orc_packed_words(n);
    int32_t *ordering = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1)), *inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    if (ordering_in) memcpy(ordering, ordering_in, sizeof(int32_t) * (size_t)n); else orc_argsort_abs(n, llr, ordering);
    for (int c = 0; c < n; c++) inv[ordering[c]] = c;
    double *llr_abs = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    for (int j = 0; j < n; j++) llr_abs[j] = fabs(llr[j]);
    uint64_t *A = (uint64_t *)calloc((size_t)m * nw + 1, sizeof(uint64_t));
    uint8_t *b = (uint8_t *)malloc((size_t)m + 1);
    for (int i = 0; i < m; i++) {
        int s = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; e++) { const int j = indices[e]; s ^= hard[j]; const int c = inv[j]; A[(size_t)i * nw + (c >> 6)] |= (uint64_t)1 << (c & 63); }
        b[i] = (uint8_t)((syndrome[i] + s) & 1);
    }
    const int maxp = m < n ? m : n;
    int64_t *pr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * maxp + 2)), *pc = pr + maxp + 1;
    const int np_ = orc_gf2_elimination_packed(m, n, nw, A, b, pr, pc);                  /* osd.py:15-17: b is s_reduced now */
    int8_t *e_perm = (int8_t *)calloc((size_t)n + 1, 1);
    for (int t = 0; t < np_; t++) e_perm[pc[t]] = (int8_t)b[pr[t]];                      /* osd.py:19-21 */
    for (int j = 0; j < n; j++) solution[j] = hard[j];
    for (int c = 0; c < n; c++) { const int j = ordering[c]; solution[j] = (int8_t)((hard[j] + e_perm[c]) & 1); }   /* osd.py:23-25 */
    int8_t *chk = (int8_t *)malloc((size_t)m + 1);
    orc_syndrome_check(m, indptr, indices, solution, chk);
    int w0 = 0;
    for (int i = 0; i < m; i++) w0 += (chk[i] != (syndrome[i] & 1));
    if (w0 == 0 || order == 0) goto done;                                                /* osd.py:27-29 */
    {
        uint8_t *is_pivot = (uint8_t *)calloc((size_t)n + 1, 1);
        for (int t = 0; t < np_; t++) is_pivot[pc[t]] = 1;
        orc_kv *np_list = (orc_kv *)malloc(sizeof(orc_kv) * (size_t)(n + 1));
        int nn = 0;
        for (int c = 0; c < n; c++) if (!is_pivot[c]) { double a = llr_abs[ordering[c]]; np_list[nn].key = (a != a) ? INFINITY : a; np_list[nn].idx = c; nn++; }   /* osd.py:31-36 */
        free(is_pivot);
        if (nn == 0) { free(np_list); goto done; }
        qsort(np_list, (size_t)nn, sizeof(orc_kv), orc_kv_cmp);                          /* osd.py:37-38 */
        const int K = nn < order + 10 ? nn : order + 10;                                 /* osd.py:40-41 */
        int8_t *best = (int8_t *)malloc((size_t)n + 1), *e_full = (int8_t *)malloc((size_t)n + 1), *test = (int8_t *)malloc((size_t)n + 1);
        memcpy(best, solution, (size_t)n);
        double best_metric = orc_osd_metric(n, solution, llr_abs, w0);                   /* osd.py:43-45 */
        int found_valid = 0;
        int64_t tested = 0;
        int comb[64];
        const int wmax = order < K ? order : K;
        for (int w = 1; w <= wmax && w < 64; w++) {                                      /* osd.py:48 */
            if (max_combinations > 0 && tested >= max_combinations) break;
            for (int t = 0; t < w; t++) comb[t] = t;
            for (;;) {                                                                   /* itertools.combinations order */
                if (max_combinations > 0 && tested >= max_combinations) break;
                memcpy(e_full, e_perm, (size_t)n);
                for (int t = 0; t < w; t++) e_full[np_list[comb[t]].idx] ^= 1;           /* osd.py:53-54 */
                for (int t = 0; t < np_; t++) {                                          /* recompute_solution kernels.py:205-219 on the ORIGINAL permuted H */
                    const int r = (int)pr[t], c = (int)pc[t];
                    int acc = 0;
                    for (int e = indptr[r]; e < indptr[r + 1]; e++) { const int col = inv[indices[e]]; if (col != c) acc ^= e_full[col]; }
                    e_full[c] = (int8_t)(b[r] ^ acc);
                }
                for (int c = 0; c < n; c++) { const int j = ordering[c]; test[j] = (int8_t)((hard[j] + e_full[c]) & 1); }   /* osd.py:57-59 */
                orc_syndrome_check(m, indptr, indices, test, chk);
                int wt = 0;
                for (int i = 0; i < m; i++) wt += (chk[i] != (syndrome[i] & 1));
                if (wt == 0) {                                                           /* osd.py:63-68 */
                    const double mt = orc_osd_metric(n, test, llr_abs, 0);
                    if (!found_valid || mt < best_metric) { memcpy(best, test, (size_t)n); best_metric = mt; found_valid = 1; }
                } else if (!found_valid) {                                               /* osd.py:69-73 */
                    const double mt = orc_osd_metric(n, test, llr_abs, wt);
                    if (mt < best_metric) { memcpy(best, test, (size_t)n); best_metric = mt; }
                }
                tested++;
                int t = w - 1;
                while (t >= 0 && comb[t] == K - w + t) t--;
                if (t < 0) break;
                comb[t]++;
                for (int u = t + 1; u < w; u++) comb[u] = comb[u - 1] + 1;
            }
        }
        memcpy(solution, best, (size_t)n);
        free(best); free(e_full); free(test); free(np_list);
    }
done:
    free(chk); free(e_perm); free(pr); free(b); free(A); free(llr_abs); free(inv); free(ordering);
}

/* a15: prior LLRs (src/simulation/engine.py:210-212): clip(nan_to_num(log((1-p)/p)), -50, 50) */
ORC_API void orc_prior_llrs(int n, const double *probs, double *llr) {
    for (int j = 0; j < n; j++) {
        double v = log((1.0 - probs[j]) / probs[j]);
        if (v != v) v = 0.0;                                   /* nan_to_num: nan -> 0, +-inf -> +-DBL_MAX */
        if (v > 50.0) v = 50.0; else if (v < -50.0) v = -50.0;
        llr[j] = v;
    }
}

/* ------------------------------------------------------------------------------------
 * Circuit-level noise (src/noise/kernels.py, op codes src/noise/constants.py:8-29)
 * ---------------------------------------------------------------------------------- */
enum { OP_CNOT = 1, OP_PREP_X = 2, OP_PREP_Z = 3, OP_MEAS_X = 4, OP_MEAS_Z = 5, OP_IDLE = 6,
       OP_X = 10, OP_Y = 11, OP_Z = 12,
       OP_XX = 20, OP_XY = 21, OP_XZ = 22, OP_YX = 23, OP_YY = 24, OP_YZ = 25, OP_ZX = 26, OP_ZY = 27, OP_ZZ = 28 };

/* a10: generate_noisy_circuit_jit (noise/kernels.py:175-353).  Returns out_len. */
ORC_API int64_t orc_generate_noisy_circuit(int64_t len, const int32_t *ops, const int32_t *q1, const int32_t *q2,
                                           double p, const double *rv, const int32_t *rp, const int32_t *rt,
                                           int32_t *oo, int32_t *o1, int32_t *o2) {
    /* two-qubit table kernels.py:283-343: err_type -> (op, first qubit selector 0=ctrl 1=tgt, two-qubit?) */
    static const int t_op[15] = { OP_X, OP_Y, OP_Z, OP_X, OP_Y, OP_Z, OP_XX, OP_YY, OP_ZZ, OP_XY, OP_YX, OP_YZ, OP_ZY, OP_XZ, OP_ZX };
    int64_t out = 0, ri = 0;
#define EMIT(a, b, c) do { oo[out] = (a); o1[out] = (b); o2[out] = (c); out++; } while (0)
    for (int64_t i = 0; i < len; i++) {
        const int op = ops[i], a = q1[i], b = q2[i];
        if (op == OP_MEAS_X) { if (rv[ri] < p) EMIT(OP_Z, a, -1); ri++; EMIT(op, a, b); }       /* :210-221 */
        else if (op == OP_MEAS_Z) { if (rv[ri] < p) EMIT(OP_X, a, -1); ri++; EMIT(op, a, b); }  /* :223-234 */
        else if (op == OP_PREP_X) { EMIT(op, a, b); if (rv[ri] < p) EMIT(OP_Z, a, -1); ri++; }  /* :236-246 */
        else if (op == OP_PREP_Z) { EMIT(op, a, b); if (rv[ri] < p) EMIT(OP_X, a, -1); ri++; }  /* :248-258 */
        else if (op == OP_IDLE) {                                                                 /* :260-272 */
            if (rv[ri] < p) { const int c = rp[ri]; EMIT(c == 0 ? OP_X : (c == 1 ? OP_Y : OP_Z), a, -1); }
            ri++;
        } else if (op == OP_CNOT) {                                                               /* :274-344 */
            EMIT(op, a, b);
            if (rv[ri] < p) {
                int t = rt[ri];
                if (t < 0 || t > 14) t = 14;                       /* the reference's final else: ZX */
                if (t < 3) EMIT(t_op[t], a, -1);
                else if (t < 6) EMIT(t_op[t], b, -1);
                else EMIT(t_op[t], a, b);
            }
            ri++;
        } else EMIT(op, a, b);                                                                    /* :346-351 */
    }
#undef EMIT
    return out;
}

/* a11: simulate_circuit_Z_jit (noise/kernels.py:13-91) */
ORC_API void orc_simulate_circuit_z(int64_t len, const int32_t *ops, const int32_t *q1, const int32_t *q2,
                                    int total_qubits, int max_syn, int8_t *hist, int8_t *state, int64_t *counts) {
    memset(state, 0, (size_t)total_qubits); memset(hist, 0, (size_t)max_syn);
    int64_t sc = 0, ec = 0;
    for (int64_t i = 0; i < len; i++) {
        const int op = ops[i], a = q1[i], b = q2[i];
        if (op == OP_CNOT) state[a] ^= state[b];
        else if (op == OP_PREP_X) state[a] = 0;
        else if (op == OP_MEAS_X) hist[sc++] = state[a];
        else if (op == OP_Z || op == OP_Y) { ec++; state[a] ^= 1; }
        else if (op == OP_ZX || op == OP_YX) { ec++; state[a] ^= 1; }
        else if (op == OP_XZ || op == OP_XY) { ec++; state[b] ^= 1; }
        else if (op == OP_ZZ || op == OP_YY || op == OP_YZ || op == OP_ZY) { ec++; state[a] ^= 1; state[b] ^= 1; }
    }
    counts[0] = sc; counts[1] = ec;
}

/* a11: simulate_circuit_X_jit (noise/kernels.py:94-172) */
ORC_API void orc_simulate_circuit_x(int64_t len, const int32_t *ops, const int32_t *q1, const int32_t *q2,
                                    int total_qubits, int max_syn, int8_t *hist, int8_t *state, int64_t *counts) {
    memset(state, 0, (size_t)total_qubits); memset(hist, 0, (size_t)max_syn);
    int64_t sc = 0, ec = 0;
    for (int64_t i = 0; i < len; i++) {
        const int op = ops[i], a = q1[i], b = q2[i];
        if (op == OP_CNOT) state[b] ^= state[a];
        else if (op == OP_PREP_Z) state[a] = 0;
        else if (op == OP_MEAS_Z) hist[sc++] = state[a];
        else if (op == OP_X || op == OP_Y) { ec++; state[a] ^= 1; }
        else if (op == OP_XZ || op == OP_YZ) { ec++; state[a] ^= 1; }
        else if (op == OP_ZX || op == OP_ZY) { ec++; state[b] ^= 1; }
        else if (op == OP_XX || op == OP_YY || op == OP_XY || op == OP_YX) { ec++; state[a] ^= 1; state[b] ^= 1; }
    }
    counts[0] = sc; counts[1] = ec;
}

/* a12: sparsify_syndrome_jit (noise/kernels.py:356-380): XOR with the RAW previous measurement */
ORC_API void orc_sparsify_syndrome(const int8_t *hist, int64_t syn_count, const int32_t *pos, const int32_t *ptrs,
                                   int num_checks, int8_t *out) {
    for (int64_t i = 0; i < syn_count; i++) out[i] = hist[i];
    for (int c = 0; c < num_checks; c++)
        for (int i = ptrs[c] + 1; i < ptrs[c + 1]; i++) {
            const int cur = pos[i], prev = pos[i - 1];
            if (cur < syn_count && prev < syn_count) out[cur] ^= hist[prev];
        }
}

/* a12: extract_data_state_jit (noise/kernels.py:383-393) */
ORC_API void orc_extract_data_state(const int8_t *state, const int32_t *idx, int n, int8_t *out) {
    for (int i = 0; i < n; i++) out[i] = state[idx[i]];
}

/* dense (k x n) binary matrix times vector mod 2: (Lx @ data_state) % 2, noise/simulation.py:81,99 */
ORC_API void orc_dense_matvec_mod2(int k, int n, const uint8_t *L, const int8_t *v, int8_t *out) {
    for (int r = 0; r < k; r++) {
        int s = 0;
        for (int j = 0; j < n; j++) s ^= (L[(size_t)r * n + j] & v[j] & 1);
        out[r] = (int8_t)s;
    }
}

/* ------------------------------------------------------------------------------------
 * Synthetic Monte-Carlo workload (this project's definition; law = alpha.py:127-128).
 * Philox4x32-10, key = (seed_lo, seed_hi), counter = (shot_lo, shot_hi, block, domain).
 * Code capacity: bit j of shot g is an error iff word (j & 3) of block (j >> 2), domain 0,
 * is < thr, thr = floor(p * 2^32).
 * ---------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
ORC_API uint32_t orc_bernoulli_threshold(double p) {
    if (p <= 0.0) return 0u;
    if (p >= 1.0) return 0xFFFFFFFFu;
    return (uint32_t)floor(p * 4294967296.0);
}

/* errors[n] for global shot g */
ORC_API void orc_cc_sample_errors(uint64_t seed, uint64_t g, int n, uint32_t thr, int8_t *err) {
    uint32_t o[4];
    for (int j = 0; j < n; j++) {
        if ((j & 3) == 0) philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(j >> 2), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        err[j] = (o[j & 3] < thr) ? 1 : 0;
    }
}

/* Code-capacity pipeline for shots [shot_begin, shot_begin+count): sample -> syndrome -> a1 decode ->
 * (OSD-0 if not converged and use_osd) -> logical compare (rule of engine.py:99-100: L.e_hat != L.e).
 * tally layout documented in include/qldpc_hip.h (QLDPC_TALLY_*). L is dense k x n. */
ORC_API void orc_cc_sample_decode_tally(int m, int n, const int32_t *indptr, const int32_t *indices,
                                        int k, const uint8_t *L, double p, uint64_t seed, int64_t shot_begin, int64_t count,
                                        int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                                        double damping, double clip, int use_osd, int threads, int64_t *tally) {
    const int nnz = indptr[m];
    const uint32_t thr = orc_bernoulli_threshold(p);
    double *prior = (double *)malloc(sizeof(double) * (size_t)n);
    for (int j = 0; j < n; j++) prior[j] = log((1.0 - p) / p);
    int64_t t_err = 0, t_conv = 0, t_osd = 0, t_it = 0, t_zero = 0, t_unsat = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads) reduction(+ : t_err, t_conv, t_osd, t_it, t_zero, t_unsat)
#endif
    {
        double *work = (double *)malloc(sizeof(double) * (size_t)(3 * nnz + n + m + 1));
        double *vals = (double *)malloc(sizeof(double) * (size_t)n);
        int8_t *e = (int8_t *)malloc((size_t)(3 * n + 2 * m));
        int8_t *cand = e + n, *sol = cand + n, *syn = sol + n, *chk = syn + m;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t t = 0; t < count; t++) {
            orc_cc_sample_errors(seed, (uint64_t)(shot_begin + t), n, thr, e);
            orc_syndrome_check(m, indptr, indices, e, syn);
            int nz = 0;
            for (int i = 0; i < m; i++) nz |= syn[i];
            uint8_t conv;
            const int fi = orc_minsum_decode(m, n, indptr, indices, syn, prior, max_iter, alpha_mode, alpha_val,
                                             alpha_seq, alpha_len, damping, clip, cand, vals, &conv, work);
            const int8_t *dec = cand;
            if (!conv && use_osd) { orc_osd0(m, n, indptr, indices, syn, vals, cand, NULL, sol); dec = sol; t_osd++; }
            orc_syndrome_check(m, indptr, indices, dec, chk);
            int unsat = 0;
            for (int i = 0; i < m; i++) unsat |= (chk[i] ^ syn[i]);
            int lerr = 0;
            for (int r = 0; r < k; r++) {
                int s = 0;
                for (int j = 0; j < n; j++) s ^= (L[(size_t)r * n + j] & (e[j] ^ dec[j]) & 1);
                lerr |= s;
            }
            t_err += lerr; t_conv += conv; t_it += fi + 1; t_zero += !nz; t_unsat += unsat;
        }
        free(e); free(vals); free(work);
    }
    free(prior);
    memset(tally, 0, sizeof(int64_t) * 16);
    tally[0] = count; tally[1] = t_err; tally[3] = t_err; tally[4] = t_conv; tally[6] = t_osd;
    tally[8] = t_it; tally[10] = t_zero; tally[12] = t_unsat;
}

/* ------------------------------------------------------------------------------------
 * Circuit-level Monte-Carlo trial (BASELINE config 5): run_trial_fast (src/noise/simulation.py:21-107) driven by
 * Philox instead of np.random, then _run_single_trial_fast (src/simulation/engine.py:68-122).
 * Random draws of trial g (this project's definition; mc_common.h):
 *   location l is faulty iff word (l & 3) of Philox(ctr = (g_lo, g_hi, l >> 2, 1)) < thr;
 *   its Pauli choice is word 0 of Philox(ctr = (g_lo, g_hi, l, 2)) mod 3 (IDLE) / mod 15 (CNOT).
 * The noisy circuit is then built and simulated literally with the a10-a12 functions above.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int64_t base_len, suffix_len;
    const int32_t *base_ops, *base_q1, *base_q2, *suffix_ops, *suffix_q1, *suffix_q2;
    int32_t total_qubits, num_x_checks, num_z_checks, n_data, k, pad_;
    const int32_t *x_syn_positions, *x_syn_ptrs, *z_syn_positions, *z_syn_ptrs, *data_qubit_indices;
    const uint8_t *Lx, *Lz;
} orc_circuit;

static int64_t orc_count_locs(const orc_circuit *c) {
    int64_t n = 0;
    for (int64_t i = 0; i < c->base_len; i++) n += (c->base_ops[i] >= OP_CNOT && c->base_ops[i] <= OP_IDLE);
    return n;
}

/* sparse_z[nsx], true_z[k], sparse_x[nsz], true_x[k] for global trial g */
ORC_API void orc_circuit_sample(const orc_circuit *c, double p, uint64_t seed, uint64_t g,
                                int8_t *sparse_z, int8_t *true_z, int8_t *sparse_x, int8_t *true_x) {
    const int64_t n_locs = orc_count_locs(c);
    const uint32_t thr = orc_bernoulli_threshold(p);
    const int64_t cap = c->base_len + n_locs + c->suffix_len;
    double *rv = (double *)malloc(sizeof(double) * (size_t)(n_locs + 1));
    int32_t *rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n_locs + 2)), *rt = rp + n_locs + 1;
    int32_t *oo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(3 * cap + 3)), *o1 = oo + cap + 1, *o2 = o1 + cap + 1;
    uint32_t o[4], w[4];
    for (int64_t l = 0; l < n_locs; l++) {
        if ((l & 3) == 0) philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(l >> 2), 1u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const int fault = o[l & 3] < thr;
        rv[l] = fault ? 0.0 : 1.0;                 /* rv < p  <=>  fault, for any p in (0,1) */
        rp[l] = 0; rt[l] = 0;
        if (fault) {
            philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)l, 2u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
            rp[l] = (int32_t)(w[0] % 3u); rt[l] = (int32_t)(w[0] % 15u);
        }
    }
    int64_t L = orc_generate_noisy_circuit(c->base_len, c->base_ops, c->base_q1, c->base_q2, p, rv, rp, rt, oo, o1, o2);
    memcpy(oo + L, c->suffix_ops, sizeof(int32_t) * (size_t)c->suffix_len);           /* simulation.py:55-69 */
    memcpy(o1 + L, c->suffix_q1, sizeof(int32_t) * (size_t)c->suffix_len);
    memcpy(o2 + L, c->suffix_q2, sizeof(int32_t) * (size_t)c->suffix_len);
    L += c->suffix_len;
    const int nsx = c->x_syn_ptrs[c->num_x_checks], nsz = c->z_syn_ptrs[c->num_z_checks];
    const int maxs = (nsx > nsz ? nsx : nsz) + 128;
    int8_t *hist = (int8_t *)malloc((size_t)maxs), *state = (int8_t *)malloc((size_t)c->total_qubits + 1);
    int8_t *data = (int8_t *)malloc((size_t)c->n_data + 1);
    int64_t counts[2];
    orc_simulate_circuit_z(L, oo, o1, o2, c->total_qubits, maxs, hist, state, counts);       /* simulation.py:72-88 */
    orc_extract_data_state(state, c->data_qubit_indices, c->n_data, data);
    orc_dense_matvec_mod2(c->k, c->n_data, c->Lx, data, true_z);
    orc_sparsify_syndrome(hist, counts[0], c->x_syn_positions, c->x_syn_ptrs, c->num_x_checks, sparse_z);
    orc_simulate_circuit_x(L, oo, o1, o2, c->total_qubits, maxs, hist, state, counts);       /* simulation.py:91-105 */
    orc_extract_data_state(state, c->data_qubit_indices, c->n_data, data);
    orc_dense_matvec_mod2(c->k, c->n_data, c->Lz, data, true_x);
    orc_sparsify_syndrome(hist, counts[0], c->z_syn_positions, c->z_syn_ptrs, c->num_z_checks, sparse_x);
    free(data); free(state); free(hist); free(oo); free(rp); free(rv);
}

typedef struct {            /* one decoding sector: Hdec (CSR), prior LLRs, logical rows of H_full as per-column bit masks */
    int32_t m, n;
    const int32_t *indptr, *indices;
    const double *prior;
    const uint64_t *logmask;
} orc_sector;

static int orc_decode_sector(const orc_sector *s, const int8_t *synd, const int8_t *true_log, int k, int max_iter, int alpha_mode,
                             double alpha_val, const double *alpha_seq, int alpha_len, double damping, double clip, int use_osd,
                             int8_t *cand, int8_t *sol, double *vals, double *work, int8_t *chk, int64_t *conv, int64_t *osd,
                             int64_t *iters, int64_t *zero, int64_t *unsat) {
    uint8_t cv;
    const int fi = orc_minsum_decode(s->m, s->n, s->indptr, s->indices, synd, s->prior, max_iter, alpha_mode, alpha_val, alpha_seq,
                                     alpha_len, damping, clip, cand, vals, &cv, work);                 /* engine.py:83-94 */
    const int8_t *det = cand;
    if (!cv && use_osd) { orc_osd0(s->m, s->n, s->indptr, s->indices, synd, vals, cand, NULL, sol); det = sol; (*osd)++; }   /* :96-97 */
    uint64_t lm = 0;
    for (int j = 0; j < s->n; j++) if (det[j] & 1) lm ^= s->logmask[j];                                  /* :99 dec = H_logical @ det */
    uint64_t tl = 0;
    for (int r = 0; r < k; r++) if (true_log[r] & 1) tl |= (uint64_t)1 << r;
    orc_syndrome_check(s->m, s->indptr, s->indices, det, chk);
    int nz = 0, bad = 0;
    for (int i = 0; i < s->m; i++) { nz |= synd[i]; bad |= (chk[i] ^ synd[i]); }
    *conv += cv; *iters += fi + 1; *zero += !nz; *unsat += (bad & 1);
    return lm != tl;                                                                                      /* :100 */
}

ORC_API void orc_circuit_sample_decode_tally(const orc_circuit *c, const orc_sector *sz, const orc_sector *sx, double p, uint64_t seed,
                                             int64_t trial_begin, int64_t count, int max_iter, int alpha_mode, double alpha_val,
                                             const double *alpha_seq, int alpha_len, double damping, double clip, int use_osd,
                                             int threads, int64_t *tally) {
    int64_t T[16];
    memset(T, 0, sizeof(T));
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        int64_t t[16];
        memset(t, 0, sizeof(t));
        const int nmax = sz->n > sx->n ? sz->n : sx->n, mmax = sz->m > sx->m ? sz->m : sx->m;
        const int nnzmax = sz->indptr[sz->m] > sx->indptr[sx->m] ? sz->indptr[sz->m] : sx->indptr[sx->m];
        double *work = (double *)malloc(sizeof(double) * (size_t)(3 * nnzmax + nmax + mmax + 1));
        double *vals = (double *)malloc(sizeof(double) * (size_t)nmax);
        int8_t *buf = (int8_t *)malloc((size_t)(2 * nmax + 3 * mmax + 2 * c->k + 8));
        int8_t *cand = buf, *sol = cand + nmax, *spz = sol + nmax, *spx = spz + mmax, *chk = spx + mmax, *tz = chk + mmax, *tx = tz + c->k;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t i = 0; i < count; i++) {
            orc_circuit_sample(c, p, seed, (uint64_t)(trial_begin + i), spz, tz, spx, tx);
            const int ze = orc_decode_sector(sz, spz, tz, c->k, max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, damping, clip, use_osd,
                                             cand, sol, vals, work, chk, &t[4], &t[6], &t[8], &t[10], &t[12]);
            const int xe = orc_decode_sector(sx, spx, tx, c->k, max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, damping, clip, use_osd,
                                             cand, sol, vals, work, chk, &t[5], &t[7], &t[9], &t[11], &t[13]);
            t[0]++; t[1] += ze; t[2] += xe; t[3] += (ze | xe);                                          /* engine.py:122, 450-457 */
        }
        free(buf); free(vals); free(work);
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int q = 0; q < 16; q++) T[q] += t[q];
    }
    memcpy(tally, T, sizeof(T));
}
