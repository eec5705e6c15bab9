// LDS/register-resident min-sum decoder for small Tanner graphs (code-capacity BB codes, Steane, ...).
//
// Mapping (MI355X-first, not a translation of the reference's serial loops):
//   * a persistent workgroup decodes S shots at a time; each shot is owned by a TEAM of TS threads;
//   * every thread plays two roles.  As a CHECK thread it owns CPT rows of H for the whole decode and keeps
//     that row's check->variable messages R (and Q_old when damping != 1) in REGISTERS across iterations; the
//     row's column indices are loaded once per kernel.  As a VARIABLE thread it owns VPT columns and keeps the
//     prior and the last posterior in registers;
//   * the only per-iteration exchange goes through LDS: R[row][k] (written contiguously by the check thread,
//     gathered by the variable threads in ASCENDING CHECK ORDER = the reference's scatter-add order,
//     kernels.py:316) and the posteriors V[col] (written contiguously, gathered by the check threads).
//     Rows are padded to an odd number of doubles so the gathers are bank-conflict free for regular codes;
//   * the variable->check message Q = clip(V[col] - R) (kernels.py:323-345) is evaluated by the check thread
//     from its register-resident R, so Q is never stored anywhere;
//   * the GF(2) syndrome test of iteration k (kernels.py:348-359) is a by-product of the V gather of
//     iteration k+1; a team freezes its outputs at the first satisfied iteration (early exit per shot) and the
//     workgroup leaves the loop when all its shots are frozen (unless QLDPC_FLAG_FIXED_ITERS).
// No global-memory traffic inside the iteration loop: HBM sees syndromes in, (err, llr, conv, iter) out.
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"
#include "minsum_f64.h"

#include <cstdlib>

namespace qldpc {

struct ResidentArgs {
    int m, n, max_iter, fixed;
    int S, TS;                       // shots per workgroup, threads per shot
    const int32_t *indptr, *indices, *colptr, *rowidx, *csc2csr;
    int64_t B;
    const int8_t *synd; const double *prior; const double *alpha;
    double damping, clip;
    int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
    // Monte-Carlo mode (MC = true): the sampler, the syndrome, the judge and the tally fused around the decode, as in minsum_regular.hip
    uint32_t seed_lo, seed_hi, thr; int use_osd;
    int64_t shot_begin;
    const uint64_t *Lmask;
    const RegCold *cold;
};

template <int CDEG> struct RowStride { static constexpr int value = (CDEG % 2 == 0) ? CDEG + 1 : CDEG; };

// MC = true: a shot is sampled here (Philox4x32-10 stream of mc_common.h keyed by the global shot index, 4 columns per block -- the stream of
// the regular fused kernel and of the CPU checker), its syndrome is the GF(2) product with the rows (a6), a converged (or, without OSD-0, any)
// shot is judged against the logical rows and tallied in the kernel, and a BP failure is exported as a record (error, syndrome, hard decision,
// posteriors) for OSD-0 + the judge of failures -- the irregular small graphs (Steane: BASELINE config 1) get the one-launch pipeline too.
template <int CDEG, int VDEG, int CPT, int VPT, bool DAMP, bool MC = false>
__global__ __launch_bounds__(1024) void minsum_resident_kernel(ResidentArgs A) {
    extern __shared__ double smem[];
    constexpr int RST = RowStride<CDEG>::value;
    const int m = A.m, n = A.n, S = A.S, TS = A.TS, max_iter = A.max_iter;
    const int slot = threadIdx.x / TS, member = threadIdx.x - slot * TS;
    const bool in_team = slot < S;
    double *Rl = smem + (size_t)(in_team ? slot : 0) * (m * RST);           // R[row][k], this team's shot
    double *Vl = smem + (size_t)S * (m * RST) + (size_t)(in_team ? slot : 0) * n;   // V[col]
    int *flags = reinterpret_cast<int *>(smem + (size_t)S * (m * RST) + (size_t)S * n);
    int *unsat = flags + 2 * (in_team ? slot : 0);                          // [2], indexed by iteration parity
    int *active = flags + 2 * S;                                            // [0] active shots, [1] (MC) some shot has a record to export
    // Monte-Carlo state behind the flags: per slot (conv, final iteration, non-zero syndrome, failure index), the logical-row accumulator,
    // the block tally, the sampled error bytes
    const int nq = (n + 3) >> 2;
    int *sres = active + 2 + 4 * (in_team ? slot : 0);
    unsigned long long *lacc_base = reinterpret_cast<unsigned long long *>(flags + ((2 * S + 2 + 4 * S + 1) & ~1));
    unsigned long long *lacc = lacc_base + (in_team ? slot : 0);
    unsigned long long *Tl = lacc_base + S;
    uint8_t *Eb = reinterpret_cast<uint8_t *>(Tl + 6) + (size_t)(in_team ? slot : 0) * (nq * 4);
    if (MC && threadIdx.x < 6) Tl[threadIdx.x] = 0ull;
    const double clip = A.clip, damping = A.damping, one_minus_d = 1.0 - A.damping;

    // ---- per-thread graph slices, loaded once (registers) ----
    int crow[CPT], cdeg[CPT], ccol[CPT][CDEG];
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int i = member + c * TS;
        crow[c] = (in_team && i < m) ? i : -1;
        cdeg[c] = 0;
#pragma unroll
        for (int k = 0; k < CDEG; k++) ccol[c][k] = 0;
        if (crow[c] >= 0) {
            const int rs = A.indptr[i];
            cdeg[c] = A.indptr[i + 1] - rs;
#pragma unroll
            for (int k = 0; k < CDEG; k++) if (k < cdeg[c]) ccol[c][k] = A.indices[rs + k];
        }
    }
    unsigned cnf[CPT];                                      // bit k: the prior of the row's k-th column is not finite (see minsum_common.h)
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        cnf[c] = 0u;
#pragma unroll
        for (int k = 0; k < CDEG; k++) if (crow[c] >= 0 && k < cdeg[c] && prior_not_finite(A.prior[ccol[c][k]])) cnf[c] |= 1u << k;
    }
    int vcol[VPT], vdeg[VPT], vrpos[VPT][VDEG];
    double vprior[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) {
        const int j = member + v * TS;
        vcol[v] = (in_team && j < n) ? j : -1;
        vdeg[v] = 0; vprior[v] = 0.0;
#pragma unroll
        for (int d = 0; d < VDEG; d++) vrpos[v][d] = 0;
        if (vcol[v] >= 0) {
            const int cs = A.colptr[j];
            vdeg[v] = A.colptr[j + 1] - cs;
            vprior[v] = A.prior[j];
#pragma unroll
            for (int d = 0; d < VDEG; d++)
                if (d < vdeg[v]) {
                    const int row = A.rowidx[cs + d];
                    vrpos[v][d] = row * RST + (A.csc2csr[cs + d] - A.indptr[row]);
                }
        }
    }

    // ---- persistent loop over groups of S shots ----
    for (int64_t base = (int64_t)blockIdx.x * S; base < A.B; base += (int64_t)gridDim.x * S) {
        const int64_t b = base + slot;
        const bool valid = in_team && b < A.B;
        int csyn[CPT];
        if (MC) {
            // ---- sample e ~ Bernoulli(p)^n (4 columns per Philox block), s = H e ----
            if (valid)
                for (int blk = member; blk < nq; blk += TS) {
                    const uint64_t gsh = (uint64_t)(A.shot_begin + b);
                    uint32_t o[4];
                    philox4x32_10((uint32_t)gsh, (uint32_t)(gsh >> 32), (uint32_t)blk, 0u, A.seed_lo, A.seed_hi, o);
#pragma unroll
                    for (int t = 0; t < 4; t++) Eb[4 * blk + t] = (4 * blk + t < n && o[t] < A.thr) ? 1 : 0;
                }
            if (in_team && member == 0) { sres[0] = 0; sres[1] = 0; sres[2] = 0; sres[3] = -1; *lacc = 0ull; }
            if (threadIdx.x == 0) active[1] = 0;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                csyn[c] = 0;
                if (valid && crow[c] >= 0) {
                    int sy = 0;
#pragma unroll
                    for (int k = 0; k < CDEG; k++) if (k < cdeg[c]) sy ^= Eb[ccol[c][k]];
                    csyn[c] = sy & 1;
                    if (csyn[c]) sres[2] = 1;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPT; c++) csyn[c] = (valid && crow[c] >= 0) ? (int)A.synd[b * m + crow[c]] : 0;
        }
        double Rprev[CPT][CDEG], Qold[CPT][CDEG], vval[VPT];
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int k = 0; k < CDEG; k++) { Rprev[c][k] = 0.0; Qold[c][k] = 0.0; }
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            vval[v] = 0.0;
            if (vcol[v] >= 0) Vl[vcol[v]] = vprior[v];                       // Q_{-1} = prior[col] (kernels.py:263-265)
        }
        if (in_team && member == 0) { unsat[0] = 0; unsat[1] = 0; }
        if (threadIdx.x == 0) {
            int64_t left = A.B - base;
            *active = (int)(left < S ? left : S);
        }
        bool done = !valid;
        __syncthreads();

        for (int it = 0; it <= max_iter; it++) {
            // ================= check phase: R_it from Q_{it-1}; syndrome test of values_{it-1} =================
            const bool run = A.fixed ? valid : !done;
            if (run) {
                const double alpha = (it < max_iter) ? A.alpha[it] : 0.0;
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    if (crow[c] < 0) continue;
                    double q[CDEG];
                    int par = 0;
#pragma unroll
                    for (int k = 0; k < CDEG; k++) {
                        q[k] = 0.0;
                        if (k < cdeg[c]) {
                            const double v = Vl[ccol[c][k]];
                            par ^= (v < 0) ? 1 : 0;                                         // kernels.py:349,356
                            q[k] = v;
                        }
                    }
                    if (it >= 1 && !done && par != csyn[c]) unsat[it & 1] = 1;            // kernels.py:357-359
                    if (it < max_iter && cdeg[c] > 0) {                                     // kernels.py:285-286
                        double sign_prod = 1.0 - 2.0 * (double)csyn[c];                     // kernels.py:252,289
                        double min1 = INFINITY, min2 = INFINITY;
                        int min1_k = -1;
#pragma unroll
                        for (int k = 0; k < CDEG; k++) {
                            if (k < cdeg[c]) {
                                double x = q[k];
                                if (it > 0) {
                                    x = clip_nan(x - Rprev[c][k], clip);                    // kernels.py:325-333
                                    if (!DAMP && ((cnf[c] >> k) & 1u)) x = NAN;             // kernels.py:336 with Q_old = +-inf
                                    if (DAMP) x = clip_only(damping * x + one_minus_d * Qold[c][k], clip);   // :336-342
                                }
                                if (DAMP) Qold[c][k] = x;
                                q[k] = x;
                                if (!(x >= 0)) sign_prod = -sign_prod;                      // kernels.py:296-299
                                const double a = fabs(x);
                                if (a < min1) { min2 = min1; min1 = a; min1_k = k; }        // kernels.py:301-306
                                else if (a < min2) { min2 = a; }
                            }
                        }
#pragma unroll
                        for (int k = 0; k < CDEG; k++) {
                            if (k < cdeg[c]) {
                                const double sign_j = (q[k] >= 0) ? 1.0 : -1.0;
                                const double mag = (k == min1_k) ? min2 : min1;
                                const double msg = alpha * (sign_prod * sign_j) * mag;      // kernels.py:311-314
                                Rprev[c][k] = msg;
                                Rl[crow[c] * RST + k] = msg;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            // ================= variable phase: freeze test, then values_it =================
            if (valid && !done) {
                const bool conv = (it >= 1) && (unsat[it & 1] == 0);                        // kernels.py:361-364
                if (conv || it == max_iter) {
                    done = true;
                    if (MC) {
                        const bool exportit = !conv && A.use_osd;
                        if (!exportit) {                                                     // judged here: residual error against the logical rows
                            unsigned long long lm = 0ull;
#pragma unroll
                            for (int v = 0; v < VPT; v++)
                                if (vcol[v] >= 0 && ((Eb[vcol[v]] ^ ((it >= 1 && vval[v] < 0) ? 1 : 0)) & 1)) lm ^= A.Lmask[vcol[v]];
                            if (lm) atomicXor(lacc, lm);
                        } else if (member == 0) {
                            active[1] = 1;
                        }
                        if (member == 0) { sres[0] = conv ? 1 : 0; sres[1] = conv ? it - 1 : max_iter - 1; atomicSub(active, 1); }
                    } else {
#pragma unroll
                        for (int v = 0; v < VPT; v++)
                            if (vcol[v] >= 0) {
                                const double x = (it >= 1) ? vval[v] : 0.0;
                                A.out_llr[b * n + vcol[v]] = x;
                                A.out_err[b * n + vcol[v]] = (x < 0) ? 1 : 0;                // kernels.py:349
                            }
                        if (member == 0) {
                            A.out_conv[b] = conv ? 1 : 0;
                            A.out_iter[b] = conv ? it - 1 : max_iter - 1;                    // kernels.py:267,362
                            atomicSub(active, 1);
                        }
                    }
                }
            }
            if (in_team && member == 0) unsat[(it + 1) & 1] = 0;
            if (it < max_iter && (A.fixed ? valid : !done)) {
#pragma unroll
                for (int v = 0; v < VPT; v++)
                    if (vcol[v] >= 0) {
                        double s = 0.0;                                                      // kernels.py:279
#pragma unroll
                        for (int d = 0; d < VDEG; d++) if (d < vdeg[v]) s += Rl[vrpos[v][d]];   // kernels.py:316, ascending check order
                        const double x = s + vprior[v];                                      // kernels.py:320
                        vval[v] = x;
                        Vl[vcol[v]] = x;
                    }
            }
            __syncthreads();
            if (!A.fixed && *active == 0) break;
        }
        __syncthreads();   // *active / unsat are re-initialised by the next group
        if (MC) {
            if (active[1]) {                        // block-uniform: some shot needs OSD-0 -> export its record
                const RegCold C = *A.cold;
                if (valid && member == 0 && sres[0] == 0 && A.use_osd) {
                    const int f = atomicAdd(C.fail_count, 1);
                    sres[3] = f;
                    C.fail_list[f] = f;
                }
                __syncthreads();
                const int f = valid ? sres[3] : -1;
                if (f >= 0) {
#pragma unroll
                    for (int c = 0; c < CPT; c++) if (crow[c] >= 0) C.f_synd[(int64_t)f * m + crow[c]] = (int8_t)csyn[c];
#pragma unroll
                    for (int v = 0; v < VPT; v++)
                        if (vcol[v] >= 0) {
                            const double xo = (max_iter >= 1) ? vval[v] : 0.0;               // values of the last iteration (what the decode entry point returns)
                            C.f_llr[(int64_t)f * n + vcol[v]] = xo;
                            C.f_hard[(int64_t)f * n + vcol[v]] = (xo < 0) ? 1 : 0;
                            C.f_err[(int64_t)f * n + vcol[v]] = (int8_t)Eb[vcol[v]];
                        }
                }
                __syncthreads();
            }
            if ((int)threadIdx.x < S && base + threadIdx.x < A.B) {
                const int *r = active + 2 + 4 * threadIdx.x;
                const unsigned long long lm = lacc_base[threadIdx.x];
                const bool exported = (r[0] == 0) && A.use_osd;
                atomicAdd(&Tl[0], 1ull);
                if (r[0]) atomicAdd(&Tl[2], 1ull);
                atomicAdd(&Tl[3], (unsigned long long)(r[1] + 1));
                if (!r[2]) atomicAdd(&Tl[4], 1ull);
                if (!exported) { if (lm) atomicAdd(&Tl[1], 1ull); if (!r[0]) atomicAdd(&Tl[5], 1ull); }
            }
            __syncthreads();
        }
    }
    if (MC && threadIdx.x < 6 && Tl[threadIdx.x]) {
        unsigned long long *tally = A.cold->tally;
        const unsigned long long v = Tl[threadIdx.x];
        const int slotmap[6] = {QLDPC_TALLY_TRIALS, QLDPC_TALLY_Z_ERR, QLDPC_TALLY_BP_CONV_Z, QLDPC_TALLY_ITERS_Z, QLDPC_TALLY_ZERO_SYND_Z, QLDPC_TALLY_UNSAT_Z};
        atomicAdd(&tally[slotmap[threadIdx.x]], v);
        if (threadIdx.x == 1) atomicAdd(&tally[QLDPC_TALLY_TOTAL_ERR], v);
    }
}

struct ResidentPlan { int cdeg, vdeg, cpt, vpt, TS, S; size_t lds; };

static bool plan_resident(const qldpc_graph *g, ResidentPlan &P) {
    if (g->m <= 0 || g->n <= 0 || g->nnz <= 0) return false;
    if (g->max_row_deg > 8 || g->max_col_deg > 4) return false;
    P.cdeg = g->max_row_deg <= 6 ? 6 : 8;
    P.vdeg = g->max_col_deg <= 3 ? 3 : 4;
    if (P.cdeg == 8 || P.vdeg == 4) { P.cdeg = 8; P.vdeg = 4; }
    const int rst = (P.cdeg % 2 == 0) ? P.cdeg + 1 : P.cdeg;
    const size_t per_slot = ((size_t)g->m * rst + g->n) * 8 + 8 + 16 + 8 + (size_t)((g->n + 3) / 4) * 4;      // R, V, flags + Monte-Carlo state (results, logical accumulator, error bytes)
    if (per_slot > 60 * 1024) return false;
    // one check / two variables per thread when the team then fits 1024 threads, else two / four
    P.cpt = 1; P.vpt = 2;
    int ts = std::max(g->m, (g->n + 1) / 2);
    if (ts > 256) { P.cpt = 2; P.vpt = 4; ts = std::max((g->m + 1) / 2, (g->n + 3) / 4); }
    if (ts > 1024) return false;
    P.TS = ts;
    int S = 1024 / ts;
    const size_t lds_budget = 64 * 1024;   // two workgroups per CU
    while (S > 1 && (size_t)S * per_slot + 96 > lds_budget) S--;
    if (S < 1) return false;
    P.S = S;
    P.lds = (size_t)S * per_slot + 16 + 16 + 48 + 16;      // + active[2], the block tally, alignment
    return true;
}

bool resident_supported(const qldpc_graph *g, double damping) {
    (void)damping;
    ResidentPlan P;
    return plan_resident(g, P);
}

template <int CDEG, int VDEG, int CPT, int VPT>
static int launch_mc_t(const ResidentArgs &A, unsigned grid, unsigned block, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((minsum_resident_kernel<CDEG, VDEG, CPT, VPT, false, true>), dim3(grid), dim3(block), lds, stream, A);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

template <int CDEG, int VDEG, int CPT, int VPT>
static int launch_t(const ResidentArgs &A, bool damp, unsigned grid, unsigned block, size_t lds, hipStream_t stream) {
    if (damp) hipLaunchKernelGGL((minsum_resident_kernel<CDEG, VDEG, CPT, VPT, true>), dim3(grid), dim3(block), lds, stream, A);
    else hipLaunchKernelGGL((minsum_resident_kernel<CDEG, VDEG, CPT, VPT, false>), dim3(grid), dim3(block), lds, stream, A);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

int minsum_resident_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                           const double *d_alpha, double damping, double clip, int flags, int8_t *d_err, double *d_llr,
                           uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    ResidentPlan P;
    if (!plan_resident(g, P)) { set_error("graph not supported by the resident kernel"); return QLDPC_ERR_UNSUPPORTED; }
    ResidentArgs A;
    A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
    A.S = P.S; A.TS = P.TS;
    A.indptr = g->d_indptr; A.indices = g->d_indices; A.colptr = g->d_colptr; A.rowidx = g->d_rowidx; A.csc2csr = g->d_csc2csr;
    A.B = B; A.synd = d_synd; A.prior = d_prior; A.alpha = d_alpha; A.damping = damping; A.clip = clip;
    A.out_err = d_err; A.out_llr = d_llr; A.out_conv = d_conv; A.out_iter = d_iter;
    const unsigned block = (unsigned)round_up((int64_t)P.S * P.TS, 64);
    int64_t groups = (B + P.S - 1) / P.S;
    const int64_t max_grid = 256 * 2 * 8;   // persistent: a few waves of workgroups per CU
    const unsigned grid = (unsigned)(groups < max_grid ? groups : max_grid);
    const bool damp = damping != 1.0;
    if (P.cdeg == 6 && P.cpt == 1) return launch_t<6, 3, 1, 2>(A, damp, grid, block, P.lds, stream);
    if (P.cdeg == 6 && P.cpt == 2) return launch_t<6, 3, 2, 4>(A, damp, grid, block, P.lds, stream);
    if (P.cdeg == 8 && P.cpt == 1) return launch_t<8, 4, 1, 2>(A, damp, grid, block, P.lds, stream);
    return launch_t<8, 4, 2, 4>(A, damp, grid, block, P.lds, stream);
}

// the fused Monte-Carlo form (damping == 1): the arguments of mc_regular_launch (minsum_regular.hip), the same record / tally conventions
int mc_resident_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                       uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask, void *d_cold, hipStream_t stream) {
    ResidentPlan P;
    if (!plan_resident(g, P)) { set_error("graph not supported by the resident kernel"); return QLDPC_ERR_UNSUPPORTED; }
    ResidentArgs A{};
    A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
    A.S = P.S; A.TS = P.TS;
    A.indptr = g->d_indptr; A.indices = g->d_indices; A.colptr = g->d_colptr; A.rowidx = g->d_rowidx; A.csc2csr = g->d_csc2csr;
    A.B = B; A.prior = d_prior; A.alpha = d_alpha; A.damping = 1.0; A.clip = clip;
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.thr = thr; A.use_osd = use_osd; A.shot_begin = shot_begin;
    A.Lmask = d_Lmask; A.cold = reinterpret_cast<const RegCold *>(d_cold);
    const unsigned block = (unsigned)round_up((int64_t)P.S * P.TS, 64);
    const int64_t groups = (B + P.S - 1) / P.S, max_grid = 256 * 2 * 8;
    const unsigned grid = (unsigned)(groups < max_grid ? (groups > 0 ? groups : 1) : max_grid);
    if (P.cdeg == 6 && P.cpt == 1) return launch_mc_t<6, 3, 1, 2>(A, grid, block, P.lds, stream);
    if (P.cdeg == 6 && P.cpt == 2) return launch_mc_t<6, 3, 2, 4>(A, grid, block, P.lds, stream);
    if (P.cdeg == 8 && P.cpt == 1) return launch_mc_t<8, 4, 1, 2>(A, grid, block, P.lds, stream);
    return launch_mc_t<8, 4, 2, 4>(A, grid, block, P.lds, stream);
}

}  // namespace qldpc
