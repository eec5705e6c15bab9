// Branch-free LDS/register-resident min-sum kernel for REGULAR Tanner graphs (every check of degree CDEG, every
// variable of degree VDEG -- all bivariate-bicycle codes: CDEG = 6, VDEG = 3), optionally fused with the whole
// code-capacity Monte-Carlo step.
//
// Thread mapping as in minsum_resident.hip (a team of TS threads owns one shot; a thread is the CHECK thread of one
// row -- its check->variable messages R stay in registers for the whole decode -- and the VARIABLE thread of two
// columns), but with compile-time degrees the check update is straight-line code:
//   * Q = clip(V[col] - R) as v_min_f64/v_max_f64 (the NaN test of kernels.py:328 is kept unless the launcher proved
//     that no NaN can arise: finite prior/clip/alpha and every check degree >= 2);
//   * min1 / min2 = the two smallest magnitudes WITH multiplicity, from a 14-op v_min/v_max network (kernels.py:301-306);
//     the position that receives min2 is selected by |q| == min1: if the minimum is attained twice, min2 == min1, so
//     this equals the reference's "first strict minimum" rule at every position;
//   * signs are boolean masks (x < 0; note -0.0 counts as >= 0 exactly like `val >= 0`, kernels.py:296), the message is
//     (+-alpha) * mag, one rounding, equal to the reference's (alpha * sign) * mag.
// MC = true fuses the sampler (Philox4x32-10 stream of mc_common.h), the GF(2) syndrome (a6), the decode, the logical
// comparison L (e xor e_hat) (engine.py:99-100) and the tally (engine.py:450-457) into the same launch: errors and
// syndromes live in LDS, the posterior in registers; only shots BP fails on are written out (for the OSD-0 stage).
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"
#include "minsum_f64.h"

#include <cstdlib>

namespace qldpc {

struct RegArgs {
    int m, n, max_iter, fixed, S, TS;
    const int32_t *indptr, *indices, *colptr, *rowidx, *csc2csr;
    int64_t B;
    const double *prior, *alpha;
    double damping, clip;
    // decode mode
    const int8_t *synd; int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
    // Monte-Carlo mode
    uint32_t seed_lo, seed_hi, thr; int use_osd;
    int64_t shot_begin;
    const uint64_t *Lmask;
    const int32_t *shot_list, *shot_count;   // Monte-Carlo mode: decode only the listed shots (offsets from shot_begin; device-resident count), else NULL
    const struct RegCold *cold;     // rarely used pointers live in device memory to keep scalar registers free
    // LDS carve (byte offsets)
    int offV, offE, offL, offI, offA, offT;
};

#ifndef QLDPC_REG_SIGNBITS
#define QLDPC_REG_SIGNBITS 0      // 1: clean inputs take signs, parities and the message sign from the high words (integer XORs) instead of f64 compares --
                                  // 12 fewer 4-cycle compares per check, measured 1.1 % SLOWER (12.74 / 12.78 vs 12.60 / 12.63 ms, same box, profiles/r03_experiments.txt)
#endif
#ifndef QLDPC_LB_T
#define QLDPC_LB_T 512
#define QLDPC_LB_W 8
#endif
// FIXED (all max_iter iterations executed for every shot, outputs frozen at convergence) is a template parameter: the fixed-work and the
// early-exit forms are distinct kernel symbols, so a rocprofv3 --stats summary lists them separately.
template <int CDEG, int VDEG, bool DAMP, bool NANFREE, bool MC, bool FIXED>
__global__ __launch_bounds__(QLDPC_LB_T, QLDPC_LB_W) void minsum_regular_kernel(RegArgs A) {
    extern __shared__ unsigned char lds[];
    constexpr int RST = (CDEG % 2 == 0) ? CDEG + 1 : CDEG;
    const int m = A.m, n = A.n, S = A.S, TS = A.TS, max_iter = A.max_iter;
    const int nq = (n + 3) >> 2;
    const int slot = threadIdx.x / TS, member = threadIdx.x - slot * TS;
    const bool in_team = slot < S;
    const int sl = in_team ? slot : 0;
    double *Rl = reinterpret_cast<double *>(lds) + (size_t)sl * m * RST;
    double *Vl = reinterpret_cast<double *>(lds + A.offV) + (size_t)sl * n;
    uint32_t *El = reinterpret_cast<uint32_t *>(lds + A.offE) + (size_t)sl * nq;
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(lds + A.offL) + sl;
    int *I = reinterpret_cast<int *>(lds + A.offI);
    int *unsat = I + 2 * sl;            // [2] by iteration parity
    int *active = I + 2 * S;            // [0] active shots, [1] any failure to export
    int *sres = I + 2 * S + 2 + 4 * sl; // conv, final_iter, nonzero syndrome, failure index
    const double *Al = reinterpret_cast<const double *>(lds + A.offA);                  // alpha_k staged in LDS
    unsigned long long *Tl = reinterpret_cast<unsigned long long *>(lds + A.offT);      // block tally (MC)
    const double clip = A.clip, nclip = -A.clip, damping = A.damping, one_minus_d = 1.0 - A.damping;
    for (int k = threadIdx.x; k < max_iter; k += blockDim.x) reinterpret_cast<double *>(lds + A.offA)[k] = A.alpha[k];
    if (MC && threadIdx.x < 6) Tl[threadIdx.x] = 0ull;
    const double prior0 = (MC && A.n > 0) ? A.prior[0] : 0.0;                                // Monte-Carlo plans: the uniform prior
    unsigned long long *clkbuf = MC ? A.cold->clk : nullptr;
    const ClkStamp clk0 = clk_begin(clkbuf);

    // ---- per-thread graph slices (registers, loaded once) ----
    const bool has_check = in_team && member < m;
    int coff[CDEG];
#pragma unroll
    for (int k = 0; k < CDEG; k++) coff[k] = has_check ? A.indices[A.indptr[member] + k] : 0;
    unsigned cnf = 0u;                                     // bit k: the prior of the row's k-th column is not finite (see minsum_common.h)
    if (!NANFREE && !DAMP) {
#pragma unroll
        for (int k = 0; k < CDEG; k++) if (has_check && prior_not_finite(A.prior[coff[k]])) cnf |= 1u << k;
    }
    const int roff = has_check ? member * RST : 0;
    bool has_var[2];
    int vj[2], voff[2][VDEG];
    double vprior[2];
#pragma unroll
    for (int v = 0; v < 2; v++) {
        const int j = member + v * TS;
        has_var[v] = in_team && j < n;
        vj[v] = has_var[v] ? j : 0;
        vprior[v] = has_var[v] ? A.prior[j] : 0.0;
#pragma unroll
        for (int d = 0; d < VDEG; d++) {
            voff[v][d] = 0;
            if (has_var[v]) {
                const int k = A.colptr[j] + d, row = A.rowidx[k];
                voff[v][d] = row * RST + (A.csc2csr[k] - A.indptr[row]);
            }
        }
    }

    // a shot list (the shots the bit-sliced first iteration, mc_first.hip, did not finish) replaces the contiguous range
    const int64_t nshots = (MC && A.shot_list) ? (int64_t)*A.shot_count : A.B;
    for (int64_t base = (int64_t)blockIdx.x * S; base < nshots; base += (int64_t)gridDim.x * S) {
        const int64_t b = base + slot;
        const bool valid = in_team && b < nshots;
        bool csyn = false;
        if (MC) {
            // ---- sample e ~ Bernoulli(p)^n (4 bits per Philox block), s = H e ----
            if (valid && member < nq) {
                const uint64_t g = (uint64_t)(A.shot_begin + (A.shot_list ? (int64_t)A.shot_list[b] : b));
                uint32_t o[4];
                philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)member, 0u, A.seed_lo, A.seed_hi, o);
                uint32_t w = 0;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (4 * member + t < n && o[t] < A.thr) w |= 1u << (8 * t);
                El[member] = w;
            }
            if (in_team && member == 0) { sres[0] = 0; sres[1] = 0; sres[2] = 0; sres[3] = -1; *lacc = 0ull; }
            if (threadIdx.x == 0) active[1] = 0;
            __syncthreads();
            const uint8_t *Eb = reinterpret_cast<const uint8_t *>(El);
            if (valid && has_check) {
                int s = 0;
#pragma unroll
                for (int k = 0; k < CDEG; k++) s ^= Eb[coff[k]];
                csyn = s & 1;
                if (csyn) sres[2] = 1;
            }
        } else {
            csyn = (valid && has_check) ? (A.synd[b * m + member] & 1) : false;
        }
        double Rprev[CDEG], Qold[CDEG];
#pragma unroll
        for (int k = 0; k < CDEG; k++) { Rprev[k] = 0.0; Qold[k] = 0.0; }
#pragma unroll
        for (int v = 0; v < 2; v++) if (has_var[v]) Vl[vj[v]] = vprior[v];             // Q_{-1} = prior[col] (kernels.py:263-265)
        if (in_team && member == 0) { unsat[0] = 0; unsat[1] = 0; }
        if (threadIdx.x == 0) { const int64_t left = nshots - base; active[0] = (int)(left < S ? left : S); }
        bool done = !valid;
        __syncthreads();

        for (int it = 0; it <= max_iter; it++) {
            // ======== check phase: syndrome test of values_{it-1}, then R_it from Q_{it-1} ========
            const bool in_check = (FIXED ? valid : !done) && has_check;
            double x[CDEG];
            if (in_check) {
                bool par = csyn;
                if (NANFREE && QLDPC_REG_SIGNBITS) {                  // clean inputs: a posterior is never -0.0 (minsum_common.h), so x < 0 <=> its sign bit: one XOR per edge
                    unsigned ph = 0u;
#pragma unroll
                    for (int k = 0; k < CDEG; k++) { x[k] = Vl[coff[k]]; ph ^= (unsigned)__double2hiint(x[k]); }
                    par ^= (ph >> 31) != 0u;
                } else {
#pragma unroll
                    for (int k = 0; k < CDEG; k++) { x[k] = Vl[coff[k]]; par ^= (x[k] < 0.0); }   // kernels.py:349,356
                }
                if (it >= 1 && !done && par) unsat[it & 1] = 1;                               // kernels.py:357-359
            }
            // Reference semantics, iteration 1: at BASELINE's error rates ~97 % of the shots pass the syndrome test of values_0 here, and the
            // messages of iteration 1 computed alongside the test (the fused form below) are thrown away for them -- a quarter of all the
            // instructions a shot costs (PMC: 372 VALU wave-instructions per shot, profiles/r02a_pmc.txt).  One extra barrier at it == 1 lets
            // a team see its verdict first and skip them; later iterations keep the fused form (a team still running there is rarely done).
            bool skip_msg = false;
            if (!FIXED && it == 1) {
                __syncthreads();
                skip_msg = (unsat[1] == 0);
            }
            if (MC && NANFREE && !DAMP && it == 0) {
                // Monte-Carlo plans decode against a UNIFORM prior p0 (code capacity, alpha.py:119-120), so iteration 0 is a closed form: every
                // variable-to-check message is p0 (kernels.py:263-265, unclipped), hence min1 = min2 = |p0| and the sign of an edge's
                // message is the syndrome sign times the signs of the other CDEG - 1 inputs: msg = +-(alpha_0 * |p0|), the same single
                // rounding as the general form below.  Saves the gather, the min network and the selects: ~60 of the ~290 instructions a
                // converging shot costs under reference semantics.
                if (in_check && max_iter > 0) {
                    const double p0 = prior0;
                    const bool sneg = csyn != ((((CDEG - 1) & 1) != 0) && (p0 < 0.0));
                    const double mag = Al[0] * fabs(p0);
                    const double msg = sneg ? -mag : mag;
#pragma unroll
                    for (int k = 0; k < CDEG; k++) { Rprev[k] = msg; Rl[roff + k] = msg; }
                }
            } else if (in_check && !skip_msg) {
                if (it < max_iter) {
                    const double alpha = Al[it];
                    if (it > 0) {
#pragma unroll
                        for (int k = 0; k < CDEG; k++) {
                            double t = x[k] - Rprev[k];                                        // kernels.py:325
                            if (!NANFREE) t = (t != t) ? 0.0 : t;                              // kernels.py:328-329
                            t = vmax(vmin(t, clip), nclip);                                    // kernels.py:330-333
                            if (DAMP) {                                                        // kernels.py:336-342
                                const double qd = damping * t + one_minus_d * Qold[k];
                                t = NANFREE ? vmax(vmin(qd, clip), nclip) : clip_only(qd, clip);       // a NaN (from a NaN Q_old) must survive the clip
                            }
                            if (!NANFREE && !DAMP && ((cnf >> k) & 1u)) t = NAN;               // kernels.py:336 with Q_old = +-inf
                            x[k] = t;
                        }
                    }
                    bool neg[CDEG];
                    bool sp = csyn;                                                            // sign of 1 - 2 s (kernels.py:252,289)
                    unsigned spw = csyn ? 0x80000000u : 0u;                                    // the same in the sign bit of a word (NANFREE: Q is never -0.0)
#pragma unroll
                    for (int k = 0; k < CDEG; k++) {
                        if (DAMP) Qold[k] = x[k];
                        if (NANFREE && !DAMP && QLDPC_REG_SIGNBITS) { spw ^= (unsigned)__double2hiint(x[k]); neg[k] = false; }
                        else {
                            neg[k] = NANFREE ? (x[k] < 0.0) : !(x[k] >= 0.0);                  // kernels.py:296-299 (-0.0 counts as >= 0)
                            sp ^= neg[k];
                        }
                    }
                    double min1, min2;
                    if (NANFREE) {
                        two_smallest_abs<CDEG>(x, min1, min2);                                 // kernels.py:301-306
                    } else {                                   // NaN-tolerant form: NaN magnitudes are ignored like `abs_val < min1`
                        min1 = INFINITY; min2 = INFINITY;
#pragma unroll
                        for (int k = 0; k < CDEG; k++) {
                            const double a = fabs(x[k]);
                            if (a < min1) { min2 = min1; min1 = a; } else if (a < min2) { min2 = a; }
                        }
                    }
                    // the first minimum gets min2, everybody else min1 (kernels.py:313).  If the minimum is attained twice,
                    // min2 == min1, so selecting on |q| == min1 gives the same value at every position.
                    const double p1 = alpha * min1, p2 = alpha * min2;                         // (+-alpha)*mag == +-(alpha*mag)
                    const int p1lo = __double2loint(p1), p1hi = __double2hiint(p1), p2lo = __double2loint(p2), p2hi = __double2hiint(p2);
#pragma unroll
                    for (int k = 0; k < CDEG; k++) {
                        const bool eq = (fabs(x[k]) == min1);
                        const int lo = eq ? p2lo : p1lo;
                        const int hi = (NANFREE && !DAMP && QLDPC_REG_SIGNBITS) ? (int)(((unsigned)(eq ? p2hi : p1hi)) ^ ((spw ^ (unsigned)__double2hiint(x[k])) & 0x80000000u))
                                                          : ((eq ? p2hi : p1hi) ^ ((sp != neg[k]) ? (int)0x80000000 : 0));   // kernels.py:311-314
                        const double msg = __hiloint2double(hi, lo);
                        Rprev[k] = msg;
                        Rl[roff + k] = msg;
                    }
                }
            }
            __syncthreads();
            // ======== variable phase: freeze test, then values_it ========
            if (valid && !done) {
                const bool conv = (it >= 1) && (unsat[it & 1] == 0);                           // kernels.py:361-364
                if (conv || it == max_iter) {
                    done = true;
                    if (MC) {
                        const bool exportit = !conv && A.use_osd;
                        if (!exportit) {
                            const uint8_t *Eb = reinterpret_cast<const uint8_t *>(El);
                            unsigned long long lm = 0ull;
#pragma unroll
                            for (int v = 0; v < 2; v++)
                                if (has_var[v] && ((Eb[vj[v]] ^ ((it >= 1 && Vl[vj[v]] < 0.0) ? 1 : 0)) & 1)) lm ^= A.Lmask[vj[v]];   // V still holds values_{it-1}
                            if (lm) atomicXor(lacc, lm);
                        } else if (member == 0) {
                            active[1] = 1;
                        }
                        if (member == 0) { sres[0] = conv ? 1 : 0; sres[1] = conv ? it - 1 : max_iter - 1; atomicSub(&active[0], 1); }
                    } else {
#pragma unroll
                        for (int v = 0; v < 2; v++)
                            if (has_var[v]) {
                                const double xo = (it >= 1) ? Vl[vj[v]] : 0.0;        // V still holds values_{it-1}
                                A.out_llr[b * n + vj[v]] = xo;
                                A.out_err[b * n + vj[v]] = (xo < 0.0) ? 1 : 0;                 // kernels.py:349
                            }
                        if (member == 0) {
                            A.out_conv[b] = conv ? 1 : 0;
                            A.out_iter[b] = conv ? it - 1 : max_iter - 1;                      // kernels.py:267,362
                            atomicSub(&active[0], 1);
                        }
                    }
                }
            }
            if (in_team && member == 0) unsat[(it + 1) & 1] = 0;
            if (it < max_iter && (FIXED ? valid : !done)) {
#pragma unroll
                for (int v = 0; v < 2; v++)
                    if (has_var[v]) {
                        double s = 0.0;                                                        // kernels.py:279
#pragma unroll
                        for (int d = 0; d < VDEG; d++) s += Rl[voff[v][d]];                    // kernels.py:316, ascending check order
                        const double xv = s + vprior[v];                                       // kernels.py:320
                        Vl[vj[v]] = xv;
                    }
            }
            __syncthreads();
            if (!FIXED && active[0] == 0) break;
        }
        __syncthreads();
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
                    const uint8_t *Eb = reinterpret_cast<const uint8_t *>(El);
                    if (has_check) C.f_synd[(int64_t)f * m + member] = csyn ? 1 : 0;
#pragma unroll
                    for (int v = 0; v < 2; v++)
                        if (has_var[v]) {
                            const double xo = (max_iter >= 1) ? Vl[vj[v]] : 0.0;
                            C.f_llr[(int64_t)f * n + vj[v]] = xo;
                            C.f_hard[(int64_t)f * n + vj[v]] = (xo < 0.0) ? 1 : 0;
                            C.f_err[(int64_t)f * n + vj[v]] = (int8_t)Eb[vj[v]];
                        }
                }
                __syncthreads();
            }
            if ((int)threadIdx.x < S && base + threadIdx.x < nshots) {
                const int *r = I + 2 * S + 2 + 4 * threadIdx.x;
                const unsigned long long lm = *(reinterpret_cast<unsigned long long *>(lds + A.offL) + threadIdx.x);
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
    if (MC) clk_end(clkbuf, clk0);
    if (MC && threadIdx.x < 6 && Tl[threadIdx.x]) {
        unsigned long long *tally = A.cold->tally;
        const unsigned long long v = Tl[threadIdx.x];
        const int slotmap[6] = {QLDPC_TALLY_TRIALS, QLDPC_TALLY_Z_ERR, QLDPC_TALLY_BP_CONV_Z, QLDPC_TALLY_ITERS_Z, QLDPC_TALLY_ZERO_SYND_Z, QLDPC_TALLY_UNSAT_Z};
        atomicAdd(&tally[slotmap[threadIdx.x]], v);
        if (threadIdx.x == 1) atomicAdd(&tally[QLDPC_TALLY_TOTAL_ERR], v);
    }
}

// judge of the exported failures after OSD-0: logical error / unsat / OSD count (32 lanes per record, device-side count)
// reset != 0: the LAST workgroup to finish zeroes the piece's counters (count[0] failures, count[2] shots the first iteration handed on; count[3] counts
// finished workgroups) -- every workgroup reads `total` before it can be counted as finished, so the next piece on this lane needs no memset from the host
__global__ __launch_bounds__(256) void cc_judge_failed_kernel(int32_t *__restrict__ count, int reset, int *__restrict__ osd_queue, int m, int n,
                                                              const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                              const uint64_t *__restrict__ Lmask, const int8_t *__restrict__ err,
                                                              const int8_t *__restrict__ synd, const int8_t *__restrict__ dec,
                                                              unsigned long long *__restrict__ tally) {
    const int total = *count;
    const int lane = threadIdx.x & 31;
    unsigned long long nerr = 0, nbad = 0;
    for (int64_t b = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); b < total; b += (int64_t)gridDim.x * 8) {
        const int8_t *e = err + b * n, *d = dec + b * n, *s = synd + b * m;
        uint64_t lm = 0;
        for (int j = lane; j < n; j += 32)
            if ((e[j] ^ d[j]) & 1) lm ^= Lmask[j];
        int bad = 0;
        for (int i = lane; i < m; i += 32) {
            int p = 0;
            for (int k = indptr[i]; k < indptr[i + 1]; k++) p ^= d[indices[k]];
            bad |= ((p ^ s[i]) & 1);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { lm ^= __shfl_xor(lm, off, 32); bad |= __shfl_xor(bad, off, 32); }
        if (lane == 0) { nerr += lm ? 1 : 0; nbad += bad ? 1 : 0; }
    }
    if (lane == 0) {
        if (nerr) { atomicAdd(&tally[QLDPC_TALLY_Z_ERR], nerr); atomicAdd(&tally[QLDPC_TALLY_TOTAL_ERR], nerr); }
        if (nbad) atomicAdd(&tally[QLDPC_TALLY_UNSAT_Z], nbad);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && total) atomicAdd(&tally[QLDPC_TALLY_OSD_Z], (unsigned long long)total);
    if (reset) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(&count[3], 1) == (int)gridDim.x - 1) { count[0] = 0; count[2] = 0; count[3] = 0; if (osd_queue) *osd_queue = 0; __threadfence(); }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
struct RegPlan { int cdeg, vdeg, TS, S, offV, offE, offL, offI, offA, offT; size_t lds; unsigned block; };
static const int kMaxIterLds = 1024;

static int g_list_shots = 0;          // qldpc_set_option("mc_list_shots"): shots per workgroup of the shot-list launch (0 = as the full launch)
void regular_set_list_shots(int s) { g_list_shots = s; }

static bool plan_regular(const qldpc_graph *g, int max_iter, RegPlan &P, int force_S = 0) {
    if (g->m <= 0 || g->n <= 0 || max_iter > kMaxIterLds) return false;
    const int cdeg = g->max_row_deg, vdeg = g->max_col_deg;
    if (!((cdeg == 6 && vdeg == 3) || (cdeg == 4 && vdeg == 2) || (cdeg == 8 && vdeg == 4))) return false;
    for (int i = 0; i < g->m; i++) if (g->indptr[i + 1] - g->indptr[i] != cdeg) return false;
    for (int j = 0; j < g->n; j++) if (g->colptr[j + 1] - g->colptr[j] != vdeg) return false;
    const int ts = std::max(g->m, (g->n + 1) / 2);
    if (ts > QLDPC_LB_T) return false;
    const int rst = (cdeg % 2 == 0) ? cdeg + 1 : cdeg;
    const int nq = (g->n + 3) / 4;
    int S = QLDPC_LB_T / ts;                           // 4 blocks per CU fill its 32 wave slots
    if (force_S > 0 && force_S < S) S = force_S;
    auto layout = [&](int s) {
        P.offV = s * g->m * rst * 8;
        P.offE = P.offV + s * g->n * 8;
        P.offL = (P.offE + s * nq * 4 + 7) / 8 * 8;
        P.offI = P.offL + s * 8;
        P.offA = (P.offI + (6 * s + 2) * 4 + 7) / 8 * 8;
        P.offT = P.offA + (max_iter > 0 ? max_iter : 1) * 8;
        P.lds = (size_t)P.offT + 6 * 8 + 16;
    };
    layout(S);
    while (S > 1 && P.lds > 39 * 1024) { S--; layout(S); }      // 4 blocks per CU within 160 KiB
    if (P.lds > 150 * 1024) return false;
    P.cdeg = cdeg; P.vdeg = vdeg; P.TS = ts; P.S = S;
    P.block = (unsigned)round_up((int64_t)S * ts, 64);
    return true;
}

bool regular_supported(const qldpc_graph *g, double clip, int max_iter) {
    RegPlan P;
    return clip >= 0.0 && plan_regular(g, max_iter, P);
}

template <int CDEG, int VDEG, bool MC, bool FIXED>
static int launch_reg2(const RegArgs &A, bool damp, bool nanfree, unsigned grid, unsigned block, size_t lds, hipStream_t stream) {
    if (damp) {
        if (MC) return QLDPC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((minsum_regular_kernel<CDEG, VDEG, true, false, false, FIXED>), dim3(grid), dim3(block), lds, stream, A);
    } else if (nanfree) {
        hipLaunchKernelGGL((minsum_regular_kernel<CDEG, VDEG, false, true, MC, FIXED>), dim3(grid), dim3(block), lds, stream, A);
    } else {
        hipLaunchKernelGGL((minsum_regular_kernel<CDEG, VDEG, false, false, MC, FIXED>), dim3(grid), dim3(block), lds, stream, A);
    }
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

template <int CDEG, int VDEG, bool MC>
static int launch_reg(const RegArgs &A, bool damp, bool nanfree, unsigned grid, unsigned block, size_t lds, hipStream_t stream) {
    return A.fixed ? launch_reg2<CDEG, VDEG, MC, true>(A, damp, nanfree, grid, block, lds, stream)
                   : launch_reg2<CDEG, VDEG, MC, false>(A, damp, nanfree, grid, block, lds, stream);
}

template <bool MC>
static int dispatch_reg(const RegPlan &P, const RegArgs &A, bool damp, bool nanfree, unsigned grid, hipStream_t stream) {
    if (P.cdeg == 6) return launch_reg<6, 3, MC>(A, damp, nanfree, grid, P.block, P.lds, stream);
    if (P.cdeg == 4) return launch_reg<4, 2, MC>(A, damp, nanfree, grid, P.block, P.lds, stream);
    return launch_reg<8, 4, MC>(A, damp, nanfree, grid, P.block, P.lds, stream);
}

static void fill_common(const qldpc_graph *g, const RegPlan &P, RegArgs &A, int64_t B, const double *d_prior, int max_iter,
                        const double *d_alpha, double damping, double clip, int flags) {
    A = RegArgs{};
    A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.fixed = (flags & QLDPC_FLAG_FIXED_ITERS) ? 1 : 0;
    A.S = P.S; A.TS = P.TS;
    A.indptr = g->d_indptr; A.indices = g->d_indices; A.colptr = g->d_colptr; A.rowidx = g->d_rowidx; A.csc2csr = g->d_csc2csr;
    A.B = B; A.prior = d_prior; A.alpha = d_alpha; A.damping = damping; A.clip = clip;
    A.offV = P.offV; A.offE = P.offE; A.offL = P.offL; A.offI = P.offI; A.offA = P.offA; A.offT = P.offT;
}

static unsigned persistent_grid(int64_t B, int S) {
    const int64_t groups = (B + S - 1) / S;
    const int64_t cap = 256 * 12;
    return (unsigned)(groups < cap ? (groups > 0 ? groups : 1) : cap);
}

int minsum_regular_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                          const double *d_alpha, double damping, double clip, int flags, bool nanfree, int8_t *d_err, double *d_llr,
                          uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    RegPlan P;
    if (!plan_regular(g, max_iter, P)) { set_error("graph is not regular (6,3)/(4,2)/(8,4)"); return QLDPC_ERR_UNSUPPORTED; }
    RegArgs A;
    fill_common(g, P, A, B, d_prior, max_iter, d_alpha, damping, clip, flags);
    A.synd = d_synd; A.out_err = d_err; A.out_llr = d_llr; A.out_conv = d_conv; A.out_iter = d_iter;
    return dispatch_reg<false>(P, A, damping != 1.0, nanfree, persistent_grid(B, P.S), stream);
}

int mc_regular_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                      bool nanfree, uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask,
                      void *d_cold, hipStream_t stream, const int32_t *d_shot_list, const int32_t *d_shot_count) {
    RegPlan P;
    // a shot list holds the few shots the first iteration did not finish; their iteration counts vary from 2 to max_iter, and a workgroup waits
    // for its slowest shot: fewer shots per workgroup there (mc_list_shots)
    if (!plan_regular(g, max_iter, P, d_shot_list ? g_list_shots : 0)) { set_error("graph is not regular (6,3)/(4,2)/(8,4)"); return QLDPC_ERR_UNSUPPORTED; }
    RegArgs A;
    fill_common(g, P, A, B, d_prior, max_iter, d_alpha, 1.0, clip, flags);
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.thr = thr; A.use_osd = use_osd; A.shot_begin = shot_begin;
    A.Lmask = d_Lmask; A.cold = reinterpret_cast<const RegCold *>(d_cold);
    A.shot_list = d_shot_list; A.shot_count = d_shot_count;
    // listed shots: a few percent of the batch at BASELINE's error rates -- a grid for B / 16 shots covers them in one or two rounds
    const unsigned grid = d_shot_list ? persistent_grid(std::max<int64_t>(B / 16, 1), P.S) * (unsigned)(g_list_shots > 0 ? 4 : 1) : persistent_grid(B, P.S);
    return dispatch_reg<true>(P, A, false, nanfree, grid, stream);
}

// Fills the device-resident cold-argument block of a Monte-Carlo plan (done once per plan).
int mc_regular_fill_cold(void *d_cold, unsigned long long *d_tally, int32_t *d_fail_count, int32_t *d_fail_list, int8_t *f_synd,
                         int8_t *f_err, int8_t *f_hard, double *f_llr, unsigned long long *d_clk) {
    RegCold C{d_tally, d_fail_count, d_fail_list, f_synd, f_err, f_hard, f_llr, d_clk};
    QLDPC_HIP_TRY(hipMemcpy(d_cold, &C, sizeof(C), hipMemcpyHostToDevice));
    return QLDPC_OK;
}
size_t mc_regular_cold_bytes() { return sizeof(RegCold); }

int judge_failed_launch(const qldpc_graph *g, int32_t *d_count, bool reset_counters, int *d_osd_queue, const uint64_t *d_Lmask, const int8_t *f_err, const int8_t *f_synd,
                        const int8_t *f_dec, unsigned long long *d_tally, hipStream_t stream) {
    hipLaunchKernelGGL(cc_judge_failed_kernel, dim3(64), dim3(256), 0, stream, d_count, reset_counters ? 1 : 0, d_osd_queue, g->m, g->n, g->d_indptr, g->d_indices, d_Lmask, f_err,
                       f_synd, f_dec, d_tally);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

}  // namespace qldpc
