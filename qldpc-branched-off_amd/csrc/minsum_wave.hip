// Wave-private min-sum kernel for REGULAR Tanner graphs (every check of degree CDEG, every variable of degree VDEG: the
// bivariate-bicycle codes), decode and fused Monte-Carlo forms.  Same arithmetic as minsum_regular.hip (the reference loop nest
// src/decoding/kernels.py:282-359 in its operand order), different mapping of a shot onto the machine:
//
//   * a TEAM of LPS = ceil(m / CPL) lanes of ONE wave owns a shot: lane l is the check thread of rows l*CPL .. l*CPL+CPL-1 (their
//     check->variable messages R stay in registers for the whole decode) and the variable thread of the 4*VB columns whose Philox
//     blocks it draws (column j belongs to block j >> 2; lane l owns blocks l, l + LPS, ...).  [[144,12,12]]: CPL = 6 -> 12 lanes per
//     shot, 5 shots per wave; [[72,12,6]]: 6 lanes, 10 shots; [[288,12,18]]: CPL = 9 -> 16 lanes, 4 shots.
//   * messages cross lanes through LDS (R[row][k] written by the check thread, gathered by the variable threads; posteriors V[col]
//     the other way), but only between lanes of the SAME wave: a wave's LDS operations execute in order, so there is NO barrier in
//     the kernel (the 72-thread teams of minsum_regular.hip straddle waves and pay two workgroup barriers per iteration: 38 % of the
//     wave cycles parked, profiles/r02h_pmc.txt).  A workgroup is one wave; nothing is shared between waves.
//   * reference semantics (per-shot early exit, kernels.py:361-364): every team runs its own iteration counter and takes its next
//     shot as soon as its current one has converged -- a team never waits for the slowest shot of a block.  The syndrome test of an
//     iteration reads only the sign words of the posteriors; messages are computed only for teams that go on.
//   * fixed-work mode (QLDPC_FLAG_FIXED_ITERS): all max_iter iterations for every shot, outputs frozen at the first converged one.
//
// Inputs must be "clean" (host-verified: every prior finite and not -0.0, |prior| <= clip, clip finite > 0, every alpha finite > 0,
// damping == 1): then no message or posterior is NaN, no posterior or variable-to-check message is -0.0 (minsum_common.h), so
//   x < 0  <=>  sign bit of x           (signs, parities and hard decisions are integer XORs of the high words),
//   clip(prior - 0.0) == prior          (iteration 0 needs no special case: Q_{-1} = prior, kernels.py:263-265),
// and the NaN test of kernels.py:328 can never fire.  Everything else goes to minsum_regular.hip.
// Monte-Carlo plans decode against a UNIFORM prior p0 (alpha.py:119-120): iteration 0 is the closed form
// msg = +-(alpha_0 |p0|) (see minsum_regular.hip), and the sampled error e_j is parked in V[j] as -1.0 / +1.0 so that the
// syndrome s = H e (a6) is the same sign-parity gather as the convergence test.
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"
#include "minsum_f64.h"

#include <atomic>
#include <cstring>

namespace qldpc {

struct WaveArgs {
    int m, n, max_iter, LPS, SPW, team_bytes, offV, offM, use_osd;
    const int32_t *indptr, *indices, *colptr, *rowidx, *csc2csr;
    int64_t B, shot_begin;
    const double *prior, *alpha;
    double clip;
    // decode mode
    const int8_t *synd; int8_t *out_err; double *out_llr; uint8_t *out_conv; int32_t *out_iter;
    // Monte-Carlo mode
    uint32_t seed_lo, seed_hi, thr;
    const uint64_t *Lmask;
    const RegCold *cold;
};

#define QLDPC_WAVE_ORDER() asm volatile("" ::: "memory")      // compiler-only: LDS operations of a wave execute in issue order

__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)__double2hiint(x); }

// tally slots a team leader accumulates in registers (one launch handles < 2^31 shot-iterations per team)
struct WaveTally { unsigned trials = 0, conv = 0, iters = 0, zsyn = 0, zerr = 0, unsat = 0; };

// one row's messages R_it[k] from the gathered posteriors x[k] = V[col_k] and the row's previous messages (kernels.py:282-316 for one
// check, kernels.py:323-345 for its edges); x is overwritten with Q_{it-1}; returns the XOR of the high words of the posteriors
template <int CDEG>
__device__ __forceinline__ unsigned check_row(double (&x)[CDEG], double (&Rp)[CDEG], unsigned synbit, double alpha, double clip, double nclip) {
    unsigned s0 = 0u, sp = synbit;                                                                   // sign of 1 - 2 s (kernels.py:252,289)
#pragma unroll
    for (int k = 0; k < CDEG; k++) {
        s0 ^= hi32(x[k]);
        const double t = vmax(vmin(x[k] - Rp[k], clip), nclip);                                       // kernels.py:325, 330-333 (no NaN: clean inputs)
        x[k] = t;
        sp ^= hi32(t);                                                                                // kernels.py:296-299
    }
    double min1, min2;
    two_smallest_abs<CDEG>(x, min1, min2);                                                            // kernels.py:301-306
    const double p1 = alpha * min1, p2 = alpha * min2;                                                // (+-alpha) * mag == +-(alpha * mag)
    const int p1lo = __double2loint(p1), p1hi = __double2hiint(p1), p2lo = __double2loint(p2), p2hi = __double2hiint(p2);
#pragma unroll
    for (int k = 0; k < CDEG; k++) {
        const bool eq = (fabs(x[k]) == min1);                                                         // the first minimum gets min2 (kernels.py:313)
        const int lo = eq ? p2lo : p1lo;
        const unsigned hi = ((unsigned)(eq ? p2hi : p1hi) & 0x7FFFFFFFu) | ((sp ^ hi32(x[k])) & 0x80000000u);   // kernels.py:311-314
        Rp[k] = __hiloint2double((int)hi, lo);
    }
    return s0;
}

template <int CDEG, int VDEG, int CPL, int VB, int RST, bool MC, bool FIXED>
__global__ __launch_bounds__(64, (CPL > 6) ? 1 : 2) void minsum_wave_kernel(WaveArgs A) {
    extern __shared__ unsigned char lds[];
    constexpr int VPL = 4 * VB;
    static_assert(CPL <= 16 && VPL <= 32, "bit masks per lane");
    const int lane = threadIdx.x, LPS = A.LPS, SPW = A.SPW, m = A.m, n = A.n, max_iter = A.max_iter;
    const int team = lane / LPS, ell = lane - team * LPS;
    const bool in_team = team < SPW;
    unsigned char *T = lds + (size_t)(in_team ? team : 0) * A.team_bytes;
    double *Rl = reinterpret_cast<double *>(T);
    double *Vl = reinterpret_cast<double *>(T + A.offV);
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(T + A.offM);
    int *fidx = reinterpret_cast<int *>(T + A.offM + 8);
    const unsigned long long tmask = in_team ? ((LPS >= 64 ? ~0ull : ((1ull << LPS) - 1ull)) << (team * LPS)) : 0ull;
    const bool leader = in_team && ell == 0;
    const int nq = (n + 3) >> 2;
    const double clip = A.clip, nclip = -A.clip;
    unsigned long long *clkbuf = MC ? A.cold->clk : nullptr;
    const ClkStamp clk0 = clk_begin(clkbuf);

    // ---- per-lane graph slices (registers, loaded once) ----
    unsigned cmask = 0u;                                   // bit c: this lane owns a row in slot c
    const double *vp[CPL][CDEG];                           // LDS address of the posterior of the row's k-th column
    const int row0 = ell * CPL;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const bool ok = in_team && row0 + c < m;
        if (ok) cmask |= 1u << c;
#pragma unroll
        for (int k = 0; k < CDEG; k++) vp[c][k] = Vl + (ok ? A.indices[A.indptr[row0 + c] + k] : 0);
    }
    double *rw = Rl + (size_t)(in_team ? row0 : 0) * RST;  // R[row0 + c][k] = rw[c * RST + k]
    unsigned vmask = 0u;                                   // bit v: this lane owns a column in slot v
    const double *rp[VPL][VDEG];                           // LDS address of the message of the column's d-th check (ascending rows)
    double *vw[VB];                                        // V[4 q .. 4 q + 3] of the lane's t-th block
    double vprior[MC ? 1 : VPL];
#pragma unroll
    for (int t = 0; t < VB; t++) {
        const int q = ell + LPS * t;
        const bool qok = in_team && q < nq;
        vw[t] = Vl + (qok ? 4 * q : 0);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int v = 4 * t + w, j = 4 * q + w;
            const bool ok = qok && j < n;
            if (ok) vmask |= 1u << v;
            if (!MC) vprior[v] = ok ? A.prior[j] : 0.0;
#pragma unroll
            for (int d = 0; d < VDEG; d++) {
                int off = 0;
                if (ok) { const int kk = A.colptr[j] + d, r = A.rowidx[kk]; off = r * RST + (A.csc2csr[kk] - A.indptr[r]); }
                rp[v][d] = Rl + off;
            }
        }
    }
    const double prior0 = (MC && n > 0) ? A.prior[0] : 0.0;
    if (MC) vprior[0] = prior0;
    const int64_t TT = (int64_t)gridDim.x * SPW;           // teams in the launch: team g takes shots g, g + TT, ...
    int64_t b = (int64_t)blockIdx.x * SPW + team;
    unsigned csyn = 0u, ebits = 0u;                        // bit c: syndrome of the row in slot c; bit v: sampled error of the column in slot v
    bool zsyn = false;
    double Rprev[CPL][CDEG];
    WaveTally tl;

    // sign-parity of the gathered posteriors per owned row (bit c): the syndrome test H e_hat == s (kernels.py:352-359) and, on the
    // +-1.0 error image, the syndrome s = H e itself
    auto sign_parity = [&]() -> unsigned {
        unsigned par = 0u;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            unsigned s = 0u;
#pragma unroll
            for (int k = 0; k < CDEG; k++) s ^= reinterpret_cast<const unsigned *>(vp[c][k])[1];
            par |= (s >> 31) << c;
        }
        return par & cmask;
    };
    // hard decisions of the owned columns from the posteriors in LDS (kernels.py:349)
    auto own_hard = [&]() -> unsigned {
        unsigned hard = 0u;
#pragma unroll
        for (int t = 0; t < VB; t++)
#pragma unroll
            for (int w = 0; w < 4; w++) hard |= (reinterpret_cast<const unsigned *>(vw[t] + w)[1] >> 31) << (4 * t + w);
        return hard & vmask;
    };

    // a team starts shot b: sample (MC) or load the syndrome, V = error image (MC) / prior (kernels.py:263-265); lanes with `ini`
    auto start_shot = [&](bool ini) {
        if (MC) {
            // e ~ Bernoulli(p)^n, 4 bits per Philox block (mc_common.h), parked as V[j] = e_j ? -1.0 : +1.0
            if (ini) {
                ebits = 0u;
                const uint64_t g = (uint64_t)(A.shot_begin + b);
#pragma unroll
                for (int t = 0; t < VB; t++) {
                    const int q = ell + LPS * t;
                    if (q < nq) {
                        uint32_t o[4];
                        philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)q, 0u, A.seed_lo, A.seed_hi, o);
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const bool own = (vmask >> (4 * t + w)) & 1u, e = own && o[w] < A.thr;
                            if (e) ebits |= 1u << (4 * t + w);
                            if (own) vw[t][w] = e ? -1.0 : 1.0;
                        }
                    }
                }
            }
            QLDPC_WAVE_ORDER();
            const unsigned s = sign_parity();                                                         // s = H e (a6, kernels.py:222-231)
            if (ini) csyn = s;
        } else if (ini) {
            csyn = 0u;
#pragma unroll
            for (int c = 0; c < CPL; c++)
                if ((cmask >> c) & 1u) csyn |= (unsigned)(A.synd[b * m + row0 + c] & 1) << c;
#pragma unroll
            for (int t = 0; t < VB; t++)
#pragma unroll
                for (int w = 0; w < 4; w++)
                    if ((vmask >> (4 * t + w)) & 1u) vw[t][w] = vprior[MC ? 0 : 4 * t + w];
        }
        const unsigned long long anys = __ballot(ini && csyn != 0u);
        if (ini) zsyn = (anys & tmask) == 0ull;
        QLDPC_WAVE_ORDER();
    };

    // outputs of a shot whose decode ended at iteration it_end with hard decisions `hard` of the owned columns; the posteriors of that
    // iteration are in V (always for reference semantics; in fixed-work mode only when the shot did not converge); lanes with `fin`
    auto finish_shot = [&](bool fin, bool conv, int it_end, unsigned hard) {
        const int final_iter = conv ? it_end - 1 : max_iter - 1;                                      // kernels.py:267,362
        if (MC) {
            const bool exportit = !conv && A.use_osd;
            if (fin && leader) { *lacc = 0ull; *fidx = -1; }
            QLDPC_WAVE_ORDER();
            if (fin && !exportit) {                                                                   // L (e xor e_hat), engine.py:99-100
                unsigned diff = (ebits ^ hard) & vmask;
                unsigned long long lm = 0ull;
                while (diff) {
                    const int v = __builtin_ctz(diff);
                    diff &= diff - 1u;
                    lm ^= A.Lmask[4 * (ell + LPS * (v >> 2)) + (v & 3)];
                }
                if (lm) atomicXor(lacc, lm);
            }
            if (fin && exportit && leader) {
                const RegCold C = *A.cold;
                const int f = atomicAdd(C.fail_count, 1);
                C.fail_list[f] = f;
                *fidx = f;
            }
            QLDPC_WAVE_ORDER();
            if (fin && exportit) {                                                                    // record for the OSD-0 stage
                const RegCold C = *A.cold;
                const int64_t f = *fidx;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if ((cmask >> c) & 1u) C.f_synd[f * m + row0 + c] = (int8_t)((csyn >> c) & 1u);
#pragma unroll
                for (int t = 0; t < VB; t++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int v = 4 * t + w;
                        if ((vmask >> v) & 1u) {
                            const int64_t j = 4 * (ell + LPS * t) + w;
                            C.f_llr[f * n + j] = (it_end >= 1) ? vw[t][w] : 0.0;
                            C.f_hard[f * n + j] = (int8_t)((hard >> v) & 1u);
                            C.f_err[f * n + j] = (int8_t)((ebits >> v) & 1u);
                        }
                    }
            }
            if (fin && leader) {
                tl.trials++;
                if (conv) tl.conv++;
                tl.iters += (unsigned)(final_iter + 1);
                if (zsyn) tl.zsyn++;
                if (!exportit) { if (*lacc) tl.zerr++; if (!conv) tl.unsat++; }
            }
            QLDPC_WAVE_ORDER();
        } else if (fin) {
#pragma unroll
            for (int t = 0; t < VB; t++)
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int v = 4 * t + w;
                    if ((vmask >> v) & 1u) A.out_err[b * n + 4 * (ell + LPS * t) + w] = (int8_t)((hard >> v) & 1u);
                }
            if (leader) { A.out_conv[b] = conv ? 1 : 0; A.out_iter[b] = final_iter; }
        }
    };
    // decode mode: the posteriors of the owned columns as they stand in V (values_{it-1}), or zeros before the first iteration
    auto write_llr = [&](bool on, bool zeros) {
        if (!MC && on) {
#pragma unroll
            for (int t = 0; t < VB; t++)
#pragma unroll
                for (int w = 0; w < 4; w++)
                    if ((vmask >> (4 * t + w)) & 1u) A.out_llr[b * n + 4 * (ell + LPS * t) + w] = zeros ? 0.0 : vw[t][w];
        }
    };
    // iteration 0 of a uniform-prior Monte-Carlo plan: msg = +-(alpha_0 |p0|), sign = syndrome sign times the signs of the other CDEG - 1
    // (equal) inputs -- the same single rounding as the general form (see minsum_regular.hip)
    auto closed_form_it0 = [&](bool on) {
        const double mag = A.alpha[0] * fabs(prior0);
        const bool flip = (((CDEG - 1) & 1) != 0) && (prior0 < 0.0);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const bool sneg = (((csyn >> c) & 1u) != 0u) != flip;
            const double msg = sneg ? -mag : mag;
#pragma unroll
            for (int k = 0; k < CDEG; k++) {
                if (on) Rprev[c][k] = msg;
                if (on && ((cmask >> c) & 1u)) rw[c * RST + k] = msg;
            }
        }
    };
    // values_it = prior + the column's messages in ascending row order (kernels.py:316-320)
    auto variable_phase = [&](bool on) {
#pragma unroll
        for (int t = 0; t < VB; t++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int v = 4 * t + w;
                double s = 0.0;                                                                       // kernels.py:279
#pragma unroll
                for (int d = 0; d < VDEG; d++) s += *rp[v][d];
                const double xv = s + vprior[MC ? 0 : v];                                             // kernels.py:320
                if (on && ((vmask >> v) & 1u)) vw[t][w] = xv;
            }
    };

    if (FIXED) {
        // ================= fixed-work mode: the teams of a wave move in lockstep, `it` is wave-uniform =================
        for (; __ballot(in_team && b < A.B) != 0ull; b += TT) {
            const bool live = in_team && b < A.B;
            start_shot(live);
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int k = 0; k < CDEG; k++) Rprev[c][k] = 0.0;                                     // x - 0.0 == x: iteration 0 reads the prior
            bool frozen = false, conv_f = false;
            int it_f = 0;
            unsigned hard_f = 0u;
            // freeze test: the first converged iteration, or max_iter, fixes the outputs; the arithmetic goes on (kernels.py:361-364)
            auto freeze_test = [&](int it, unsigned par) {
                const unsigned long long ub = __ballot(((par ^ csyn) & cmask) != 0u);
                const bool conv = it >= 1 && (ub & tmask) == 0ull;
                const bool fin = live && !frozen && (conv || it >= max_iter);
                if (__ballot(fin) != 0ull) {
                    const unsigned h = (it >= 1) ? own_hard() : 0u;
                    if (fin) { frozen = true; conv_f = conv; it_f = it; hard_f = h; }
                    write_llr(fin, it < 1);
                }
            };
            // the loop body is the same straight-line code for every iteration 0 < it < max_iter (iteration 0 of a Monte-Carlo plan and the
            // final syndrome test are peeled): the row messages live in the same registers throughout, no copies at control-flow joins
            int it = 0;
            if (MC && max_iter > 0) {
                closed_form_it0(live);
                QLDPC_WAVE_ORDER();
                variable_phase(live);
                QLDPC_WAVE_ORDER();
                it = 1;
            }
            for (; it < max_iter; it++) {
                // ---- check phase: syndrome test of values_{it-1} fused with R_it from Q_{it-1} (kernels.py:282-316, 352-359) ----
                unsigned par = 0u;
                const double alpha = A.alpha[it];
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    double x[CDEG];
#pragma unroll
                    for (int k = 0; k < CDEG; k++) x[k] = *vp[c][k];
                    const unsigned s0 = check_row<CDEG>(x, Rprev[c], ((csyn >> c) & 1u) << 31, alpha, clip, nclip);
                    par |= (s0 >> 31) << c;
                    if ((cmask >> c) & 1u) {
#pragma unroll
                        for (int k = 0; k < CDEG; k++) rw[c * RST + k] = Rprev[c][k];
                    }
                }
                QLDPC_WAVE_ORDER();
                freeze_test(it, par);
                variable_phase(live);                                                                 // ---- variable phase ----
                QLDPC_WAVE_ORDER();
            }
            freeze_test(max_iter, max_iter >= 1 ? sign_parity() : 0u);                                 // syndrome test of values_{max_iter-1}
            finish_shot(live, conv_f, it_f, hard_f);
        }
    } else {
        // ================= reference semantics: every team has its own iteration counter and refills as soon as it is done =================
        bool live = in_team && b < A.B, need_init = live;
        int it = 0;
        while (__ballot(live) != 0ull) {
            // ---- A: teams that start a shot ----
            if (__ballot(live && need_init) != 0ull) {
                const bool ini = live && need_init;
                start_shot(ini);
                if (ini) {
                    it = 0; need_init = false;
#pragma unroll
                    for (int c = 0; c < CPL; c++)
#pragma unroll
                        for (int k = 0; k < CDEG; k++) Rprev[c][k] = 0.0;
                }
            }
            // ---- B: syndrome test of values_{it-1} on the sign words; converged / exhausted teams finish and take their next shot ----
            const bool testing = live && (it >= 1 || max_iter == 0);
            if (__ballot(testing) != 0ull) {
                const unsigned par = (sign_parity() ^ csyn) & cmask;                                  // kernels.py:352-359
                const unsigned long long ub = __ballot(testing && it >= 1 && par != 0u);
                const bool conv = testing && it >= 1 && (ub & tmask) == 0ull;                          // kernels.py:361-364
                const bool fin = testing && (conv || it >= max_iter);
                if (__ballot(fin) != 0ull) {
                    const unsigned h = own_hard();
                    write_llr(fin, it < 1);
                    finish_shot(fin, conv, it, (it >= 1) ? h : 0u);
                    if (fin) { b += TT; live = b < A.B; need_init = live; }
                }
            }
            // ---- C: check phase for the teams that go on -- R_it from Q_{it-1} (kernels.py:282-316) ----
            const bool run = live && !need_init;             // it < max_iter here: exhausted teams were finished in B
            if (__ballot(run) == 0ull) continue;
            if (MC && __ballot(run && it != 0) == 0ull) {
                closed_form_it0(run);
            } else {
                double alpha = 0.0;
                if (run) alpha = A.alpha[it];
                const bool it0mc = MC && it == 0;            // a team at iteration 0 beside older teams: V holds its error image, Q_{-1} = p0
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    double x[CDEG], Rn[CDEG];
#pragma unroll
                    for (int k = 0; k < CDEG; k++) { x[k] = it0mc ? prior0 : *vp[c][k]; Rn[k] = it0mc ? 0.0 : Rprev[c][k]; }
                    (void)check_row<CDEG>(x, Rn, ((csyn >> c) & 1u) << 31, alpha, clip, nclip);
#pragma unroll
                    for (int k = 0; k < CDEG; k++) {
                        if (run) Rprev[c][k] = Rn[k];
                        if (run && ((cmask >> c) & 1u)) rw[c * RST + k] = Rn[k];
                    }
                }
            }
            QLDPC_WAVE_ORDER();
            // ---- D: variable phase ----
            variable_phase(run);
            QLDPC_WAVE_ORDER();
            if (run) it++;
        }
    }
    if (MC) {
        clk_end(clkbuf, clk0);
        // wave sum of the leaders' counters, one atomic per slot and wave
        unsigned v6[6] = {tl.trials, tl.zerr, tl.conv, tl.iters, tl.zsyn, tl.unsat};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            unsigned x = leader ? v6[i] : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
            v6[i] = x;
        }
        if (lane == 0) {
            unsigned long long *tally = A.cold->tally;
            const int slotmap[6] = {QLDPC_TALLY_TRIALS, QLDPC_TALLY_Z_ERR, QLDPC_TALLY_BP_CONV_Z, QLDPC_TALLY_ITERS_Z, QLDPC_TALLY_ZERO_SYND_Z, QLDPC_TALLY_UNSAT_Z};
#pragma unroll
            for (int i = 0; i < 6; i++) if (v6[i]) atomicAdd(&tally[slotmap[i]], (unsigned long long)v6[i]);
            if (v6[1]) atomicAdd(&tally[QLDPC_TALLY_TOTAL_ERR], (unsigned long long)v6[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
// shape switches of this kernel (qldpc_set_option, options.hip)
std::atomic<int> g_opt_wave_cpl{0}, g_opt_wave_rst{0}, g_opt_wave_grid{0};

struct WavePlan { int cpl, vb, rst, LPS, SPW, team_bytes, offV, offM; size_t lds; int waves_per_cu; };

static bool plan_wave(const qldpc_graph *g, WavePlan &P) {
    if (g->m <= 0 || g->n <= 0) return false;
    if (!(g->max_row_deg == 6 && g->max_col_deg == 3)) return false;
    for (int i = 0; i < g->m; i++) if (g->indptr[i + 1] - g->indptr[i] != 6) return false;
    for (int j = 0; j < g->n; j++) if (g->colptr[j + 1] - g->colptr[j] != 3) return false;
    const int nq = (g->n + 3) / 4;
    static const int cands[][2] = {{6, 3}, {5, 3}, {9, 5}, {4, 2}};       // instantiated (CPL, VB) pairs
    double best = -1.0;
    const int force_cpl = g_opt_wave_cpl.load(), force_rst = g_opt_wave_rst.load();
    for (const auto &cd : cands) {
        const int cpl = cd[0], vb = cd[1];
        if (force_cpl && cpl != force_cpl) continue;
        const int LPS = (g->m + cpl - 1) / cpl;
        if (LPS > 64 || (nq + LPS - 1) / LPS > vb) continue;
        const int SPW = 64 / LPS;
        for (int rst : {6, 7}) {
            if (force_rst && rst != force_rst) continue;
            const int offV = g->m * rst * 8, offM = offV + g->n * 8, team_bytes = offM + 16;
            const size_t lds = (size_t)SPW * team_bytes;
            if (lds > 64 * 1024) continue;
            const int limit = cpl > 6 ? 4 : 8;                             // one or two waves per SIMD (register budget of the instantiation)
            const int wpc = (int)std::min<size_t>(limit, (160 * 1024) / (lds + 256));
            if (wpc < 1) continue;
            // lanes doing useful check work per issued lane, times how far the CU's issue slots are covered (two waves per SIMD hide each
            // other's LDS latency; measured on [[144,12,12]]: profiles/r03_wave_kernel.txt)
            const double util = (double)g->m * SPW / (64.0 * cpl) * std::min(1.0, wpc / 7.0);
            if (util > best) { best = util; P = WavePlan{cpl, vb, rst, LPS, SPW, team_bytes, offV, offM, lds, wpc}; }
        }
    }
    return best > 0.0;
}

bool wave_supported(const qldpc_graph *g, double damping, bool clean) {
    WavePlan P;
    if (wave_kernel_choice() == 1) return false;
    return clean && damping == 1.0 && plan_wave(g, P);
}

template <int CPL, int VB, int RST, bool MC>
static int launch_wave3(const WaveArgs &A, bool fixed, unsigned grid, size_t lds, int device, hipStream_t stream) {
    int rc;
    if (fixed) {
        if ((rc = ensure_max_lds(device, reinterpret_cast<const void *>(minsum_wave_kernel<6, 3, CPL, VB, RST, MC, true>), 64 * 1024)) != QLDPC_OK) return rc;
        hipLaunchKernelGGL((minsum_wave_kernel<6, 3, CPL, VB, RST, MC, true>), dim3(grid), dim3(64), lds, stream, A);
    } else {
        if ((rc = ensure_max_lds(device, reinterpret_cast<const void *>(minsum_wave_kernel<6, 3, CPL, VB, RST, MC, false>), 64 * 1024)) != QLDPC_OK) return rc;
        hipLaunchKernelGGL((minsum_wave_kernel<6, 3, CPL, VB, RST, MC, false>), dim3(grid), dim3(64), lds, stream, A);
    }
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

template <bool MC>
static int launch_wave(const WavePlan &P, const WaveArgs &A, bool fixed, unsigned grid, int device, hipStream_t stream) {
#define QLDPC_WAVE_CASE(C, V)                                                                                     \
    if (P.cpl == C && P.vb == V)                                                                                  \
        return P.rst == 6 ? launch_wave3<C, V, 6, MC>(A, fixed, grid, P.lds, device, stream) : launch_wave3<C, V, 7, MC>(A, fixed, grid, P.lds, device, stream);
    QLDPC_WAVE_CASE(6, 3) QLDPC_WAVE_CASE(5, 3) QLDPC_WAVE_CASE(9, 5) QLDPC_WAVE_CASE(4, 2)
#undef QLDPC_WAVE_CASE
    set_error("wave kernel: no instantiation for CPL=%d VB=%d", P.cpl, P.vb);
    return QLDPC_ERR_UNSUPPORTED;
}

static void fill_wave(const qldpc_graph *g, const WavePlan &P, WaveArgs &A, int64_t B, const double *d_prior, int max_iter, const double *d_alpha,
                      double clip) {
    A = WaveArgs{};
    A.m = g->m; A.n = g->n; A.max_iter = max_iter; A.LPS = P.LPS; A.SPW = P.SPW; A.team_bytes = P.team_bytes; A.offV = P.offV; A.offM = P.offM;
    A.indptr = g->d_indptr; A.indices = g->d_indices; A.colptr = g->d_colptr; A.rowidx = g->d_rowidx; A.csc2csr = g->d_csc2csr;
    A.B = B; A.prior = d_prior; A.alpha = d_alpha; A.clip = clip;
}

static unsigned wave_grid(const WavePlan &P, int64_t B) {
    const int64_t groups = (B + P.SPW - 1) / P.SPW;
    const int per_cu = g_opt_wave_grid.load() > 0 ? g_opt_wave_grid.load() : P.waves_per_cu;
    const int64_t cap = (int64_t)256 * per_cu;               // persistent: exactly the waves the chip holds at once
    return (unsigned)std::max<int64_t>(1, std::min(groups, cap));
}

int minsum_wave_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter, const double *d_alpha,
                       double clip, int flags, int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream) {
    WavePlan P;
    if (!plan_wave(g, P)) { set_error("graph is not (6,3)-regular"); return QLDPC_ERR_UNSUPPORTED; }
    WaveArgs A;
    fill_wave(g, P, A, B, d_prior, max_iter, d_alpha, clip);
    A.synd = d_synd; A.out_err = d_err; A.out_llr = d_llr; A.out_conv = d_conv; A.out_iter = d_iter;
    return launch_wave<false>(P, A, (flags & QLDPC_FLAG_FIXED_ITERS) != 0, wave_grid(P, B), g->device, stream);
}

int mc_wave_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                   uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask, void *d_cold, hipStream_t stream) {
    WavePlan P;
    if (!plan_wave(g, P)) { set_error("graph is not (6,3)-regular"); return QLDPC_ERR_UNSUPPORTED; }
    WaveArgs A;
    fill_wave(g, P, A, B, d_prior, max_iter, d_alpha, clip);
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.thr = thr; A.use_osd = use_osd; A.shot_begin = shot_begin;
    A.Lmask = d_Lmask; A.cold = reinterpret_cast<const RegCold *>(d_cold);
    return launch_wave<true>(P, A, (flags & QLDPC_FLAG_FIXED_ITERS) != 0, wave_grid(P, B), g->device, stream);
}

}  // namespace qldpc

