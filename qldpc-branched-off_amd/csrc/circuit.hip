// Circuit-level Monte-Carlo (BASELINE config 5): run_trial_fast (reference src/noise/simulation.py:21-107) and the
// per-trial pipeline _run_single_trial_fast (src/simulation/engine.py:68-122), batched on the device.
//
// MI355X-first formulation.  The reference builds a noisy op list per trial and walks ~20k ops twice.  Pauli-frame
// propagation is linear over GF(2) in the inserted faults, so here the effect of ONE flip at every (error location,
// qubit slot) -- its detector flips and logical flips, per sector -- is computed once per plan by a batched frame
// simulation (one lane per single-fault circuit; the same thing the reference's builder.py:37-51 does to build H).
// A trial is then: draw the faulty locations (Philox), XOR the signatures of their Pauli components into a bit-set in
// LDS.  Equality with the literal simulation is asserted against the CPU checker on identical random draws.
// Random draws of trial g: location l faulty iff word (l&3) of Philox(g, block l>>2, domain 1) < thr; its Pauli choice
// is word 0 of Philox(g, l, domain 2) mod 3 (IDLE, noise/kernels.py:260-272) or mod 15 (CNOT, :274-344).
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace qldpc {

enum { C_OP_CNOT = 1, C_OP_PREP_X = 2, C_OP_PREP_Z = 3, C_OP_MEAS_X = 4, C_OP_MEAS_Z = 5, C_OP_IDLE = 6 };   // noise/constants.py:8-14

// ---- single-fault frame simulation: lane = (location, slot); ops shared; frames laid out [qubit][lane] ----
template <bool XSECTOR>
__global__ void fault_signature_kernel(int nl, const int32_t *__restrict__ lane_pos, const int8_t *__restrict__ lane_after,
                                       const int32_t *__restrict__ lane_qubit, int64_t len, const int32_t *__restrict__ ops,
                                       const int32_t *__restrict__ q1, const int32_t *__restrict__ q2, int tq, int nsyn,
                                       int8_t *__restrict__ state, int8_t *__restrict__ hist) {
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= nl) return;
    for (int q = 0; q < tq; q++) state[(size_t)q * nl + lane] = 0;
    const int pos = lane_pos[lane], fq = lane_qubit[lane];
    const bool after = lane_after[lane] != 0;
    int sc = 0;
    for (int64_t i = 0; i < len; i++) {
        const int op = ops[i], a = q1[i], c = q2[i];
        if (i == pos && !after && fq >= 0) state[(size_t)fq * nl + lane] ^= 1;       // Meas: flip BEFORE (kernels.py:210-221)
        if (op == C_OP_CNOT) {
            if (!XSECTOR) state[(size_t)a * nl + lane] ^= state[(size_t)c * nl + lane];   // Z: target -> control (kernels.py:55-57)
            else state[(size_t)c * nl + lane] ^= state[(size_t)a * nl + lane];            // X: control -> target (kernels.py:136-138)
        } else if (op == (XSECTOR ? C_OP_PREP_Z : C_OP_PREP_X)) {
            state[(size_t)a * nl + lane] = 0;
        } else if (op == (XSECTOR ? C_OP_MEAS_Z : C_OP_MEAS_X)) {
            if (sc < nsyn) hist[(size_t)sc * nl + lane] = state[(size_t)a * nl + lane];
            sc++;
        }
        if (i == pos && after && fq >= 0) state[(size_t)fq * nl + lane] ^= 1;        // Prep/CNOT: flip AFTER (kernels.py:236-258,274)
    }
}

struct SigTab {           // signatures of one sector; entry e = 2 * location + slot
    const int32_t *ptr;   // [2*n_locs + 1]
    const uint16_t *idx;  // detector indices
    const uint64_t *log;  // logical flips, bit r = logical row r
};

// component masks: bit0 = X on slot 0, bit1 = X on slot 1, bit2 = Z on slot 0, bit3 = Z on slot 1
__constant__ uint8_t c_cnot_comp[15] = {1, 5, 4, 2, 10, 8, 3, 15, 12, 11, 7, 13, 14, 9, 6};   // order of noise/kernels.py:283-343
__constant__ uint8_t c_idle_comp[3] = {1, 5, 4};                                               // X, Y, Z (kernels.py:263-268)

__device__ __forceinline__ void xor_signature(const SigTab &T, int e, uint32_t *bits, unsigned long long *logacc) {
    for (int k = T.ptr[e]; k < T.ptr[e + 1]; k++) { const int d = T.idx[k]; atomicXor(&bits[d >> 5], 1u << (d & 31)); }
    const unsigned long long lm = T.log[e];
    if (lm) atomicXor(logacc, lm);
}

// one workgroup per trial (grid-stride): draw faults, accumulate both sectors' detector bit-sets in LDS, write them out
__global__ __launch_bounds__(256) void circuit_sample_kernel(int64_t B, int64_t trial_begin, uint32_t seed_lo, uint32_t seed_hi, uint32_t thr,
                                                             int n_locs, const uint8_t *__restrict__ loc_type, SigTab Z, SigTab X, int nsx,
                                                             int nsz, int8_t *__restrict__ syn_z, int8_t *__restrict__ syn_x,
                                                             unsigned long long *__restrict__ true_z, unsigned long long *__restrict__ true_x,
                                                             int32_t *__restrict__ fail_counts) {
    extern __shared__ uint32_t sm[];
    if (fail_counts && blockIdx.x == 0 && threadIdx.x < 8) fail_counts[threadIdx.x] = 0;      // [0] Z, [4] X BP failures of this batch (no memset launches)
    const int wz = (nsx + 31) >> 5, wx = (nsz + 31) >> 5;
    uint32_t *bz = sm, *bx = sm + wz;
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(sm + ((wz + wx + 1) & ~1));
    const int nblk = (n_locs + 3) >> 2;
    for (int64_t t = blockIdx.x; t < B; t += gridDim.x) {
        for (int w = threadIdx.x; w < wz + wx; w += blockDim.x) sm[w] = 0;
        if (threadIdx.x < 2) lacc[threadIdx.x] = 0ull;
        __syncthreads();
        const uint64_t g = (uint64_t)(trial_begin + t);
        for (int blk = threadIdx.x; blk < nblk; blk += blockDim.x) {
            uint32_t o[4];
            philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)blk, 1u, seed_lo, seed_hi, o);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int l = 4 * blk + w;
                if (l < n_locs && o[w] < thr) {
                    const int ty = loc_type[l];
                    int comp;
                    if (ty == C_OP_MEAS_X || ty == C_OP_PREP_X) comp = 4;            // Z flip (kernels.py:211-216, 241-246)
                    else if (ty == C_OP_MEAS_Z || ty == C_OP_PREP_Z) comp = 1;       // X flip (kernels.py:224-229, 253-258)
                    else {
                        uint32_t r[4];
                        philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)l, 2u, seed_lo, seed_hi, r);
                        comp = (ty == C_OP_IDLE) ? c_idle_comp[r[0] % 3u] : c_cnot_comp[r[0] % 15u];
                    }
                    if (comp & 1) xor_signature(X, 2 * l, bx, &lacc[1]);
                    if (comp & 2) xor_signature(X, 2 * l + 1, bx, &lacc[1]);
                    if (comp & 4) xor_signature(Z, 2 * l, bz, &lacc[0]);
                    if (comp & 8) xor_signature(Z, 2 * l + 1, bz, &lacc[0]);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nsx; i += blockDim.x) syn_z[t * nsx + i] = (bz[i >> 5] >> (i & 31)) & 1;
        for (int i = threadIdx.x; i < nsz; i += blockDim.x) syn_x[t * nsz + i] = (bx[i >> 5] >> (i & 31)) & 1;
        if (threadIdx.x == 0) { true_z[t] = lacc[0]; true_x[t] = lacc[1]; }
        __syncthreads();
    }
}

// per trial (32 lanes): decoded logical action H_logical @ det (engine.py:99,119) vs the true logical flips, both sectors
struct JudgeSector {
    int m, n;
    const int32_t *indptr, *indices, *colptr, *rowidx;
    const uint64_t *logmask;
    const int8_t *synd, *det;
    const uint8_t *conv;
    const int32_t *iters;
    const unsigned long long *true_log;
};

__device__ __forceinline__ void judge_sector(const JudgeSector &S, int64_t b, int lane, bool &err, bool &nz, bool &bad) {
    const int8_t *d = S.det + b * S.n, *s = S.synd + b * S.m;
    uint64_t lm = 0;
    for (int j = lane; j < S.n; j += 32) if (d[j] & 1) lm ^= S.logmask[j];
    int bd = 0, z = 0;
    for (int i = lane; i < S.m; i += 32) {
        int p = 0;
        for (int k = S.indptr[i]; k < S.indptr[i + 1]; k++) p ^= d[S.indices[k]];
        bd |= ((p ^ s[i]) & 1);
        z |= (s[i] & 1);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { lm ^= __shfl_xor(lm, off, 32); bd |= __shfl_xor(bd, off, 32); z |= __shfl_xor(z, off, 32); }
    err = (lm != S.true_log[b]);
    nz = z != 0;
    bad = bd != 0;
}

// The same verdicts from the ONES of the correction: a decoded error has ~100 ones among ~8 800 columns, so H @ det is ~600 parity flips (the columns' rows, CSC) into a bit
// set of the trial in LDS instead of the ~31 000 byte gathers per sector of the row-wise form above (1.15 ms per 16 384-trial batch of config 5); the scan of det -- aligned
// dwords, the ragged head and tail as bytes -- is what is left.  `par`: m bits of LDS owned by the trial's 32 lanes.
__device__ __forceinline__ void judge_sector_sparse(const JudgeSector &S, int64_t b, int lane, uint32_t *par, bool &err, bool &nz, bool &bad) {
    const uint8_t *d = reinterpret_cast<const uint8_t *>(S.det) + b * S.n;
    const int8_t *s = S.synd + b * S.m;
    const int mwords = (S.m + 31) >> 5;
    for (int w = lane; w < mwords; w += 32) par[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    uint64_t lm = 0;
    auto one = [&](int j) {
        lm ^= S.logmask[j];
        for (int e = S.colptr[j]; e < S.colptr[j + 1]; e++) { const int r = S.rowidx[e]; atomicXor(&par[r >> 5], 1u << (r & 31)); }
    };
    const int head = (int)((4 - (reinterpret_cast<uintptr_t>(d) & 3)) & 3), nhead = head < S.n ? head : S.n;
    if (lane < nhead && (d[lane] & 1)) one(lane);
    const int nw = (S.n - nhead) >> 2;
    const uint32_t *dw = reinterpret_cast<const uint32_t *>(d + nhead);
    for (int w = lane; w < nw; w += 32) {
        uint32_t x = dw[w] & 0x01010101u;
        while (x) {                                                        // (rare: a correction is sparse)
            const int byte = (__builtin_ctz(x)) >> 3;
            x &= x - 1u;
            one(nhead + 4 * w + byte);
        }
    }
    const int tail0 = nhead + 4 * nw;
    if (tail0 + lane < S.n && (d[tail0 + lane] & 1)) one(tail0 + lane);
    __builtin_amdgcn_wave_barrier();
    int bd = 0, z = 0;
    for (int i = lane; i < S.m; i += 32) {
        const int si = s[i] & 1;
        bd |= (int)((par[i >> 5] >> (i & 31)) & 1u) ^ si;
        z |= si;
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { lm ^= __shfl_xor(lm, off, 32); bd |= __shfl_xor(bd, off, 32); z |= __shfl_xor(z, off, 32); }
    err = (lm != S.true_log[b]);
    nz = z != 0;
    bad = bd != 0;
}

template <bool SPARSE>
__global__ __launch_bounds__(256) void circuit_judge_kernel(int64_t B, JudgeSector Z, JudgeSector X, unsigned long long *__restrict__ tally,
                                                            uint8_t *__restrict__ outcome, const int32_t *__restrict__ fail_counts) {
    __shared__ unsigned long long acc[QLDPC_TALLY_SLOTS];
    __shared__ uint32_t parbits[SPARSE ? 8 * 2 * 128 : 1];                      // per trial of the workgroup: the two sectors' parity bits (m <= 4096)
    if (threadIdx.x < QLDPC_TALLY_SLOTS) acc[threadIdx.x] = 0ull;
    __syncthreads();
    if (fail_counts && blockIdx.x == 0 && threadIdx.x == 0) {                 // OSD-0 calls of this batch = its BP failures per sector
        acc[QLDPC_TALLY_OSD_Z] = (unsigned long long)fail_counts[0];
        acc[QLDPC_TALLY_OSD_X] = (unsigned long long)fail_counts[4];
    }
    __syncthreads();
    const int lane = threadIdx.x & 31;
    const int64_t b = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (b < B) {
        bool ze, zn, zb, xe, xn, xb;
        if (SPARSE) {
            uint32_t *par = parbits + (threadIdx.x >> 5) * 256;
            judge_sector_sparse(Z, b, lane, par, ze, zn, zb);
            judge_sector_sparse(X, b, lane, par + 128, xe, xn, xb);
        } else {
            judge_sector(Z, b, lane, ze, zn, zb);
            judge_sector(X, b, lane, xe, xn, xb);
        }
        if (lane == 0) {
            if (outcome) outcome[b] = (uint8_t)((ze ? 1 : 0) | (xe ? 2 : 0));               // (z_err, x_err) of engine.py:117-122
            atomicAdd(&acc[QLDPC_TALLY_TRIALS], 1ull);
            if (ze) atomicAdd(&acc[QLDPC_TALLY_Z_ERR], 1ull);
            if (xe) atomicAdd(&acc[QLDPC_TALLY_X_ERR], 1ull);
            if (ze || xe) atomicAdd(&acc[QLDPC_TALLY_TOTAL_ERR], 1ull);                       // engine.py:122
            if (Z.conv[b]) atomicAdd(&acc[QLDPC_TALLY_BP_CONV_Z], 1ull);
            if (X.conv[b]) atomicAdd(&acc[QLDPC_TALLY_BP_CONV_X], 1ull);
            atomicAdd(&acc[QLDPC_TALLY_ITERS_Z], (unsigned long long)(Z.iters[b] + 1));
            atomicAdd(&acc[QLDPC_TALLY_ITERS_X], (unsigned long long)(X.iters[b] + 1));
            if (!zn) atomicAdd(&acc[QLDPC_TALLY_ZERO_SYND_Z], 1ull);
            if (!xn) atomicAdd(&acc[QLDPC_TALLY_ZERO_SYND_X], 1ull);
            if (zb) atomicAdd(&acc[QLDPC_TALLY_UNSAT_Z], 1ull);
            if (xb) atomicAdd(&acc[QLDPC_TALLY_UNSAT_X], 1ull);
        }
    }
    __syncthreads();
    if (threadIdx.x < QLDPC_TALLY_SLOTS && acc[threadIdx.x]) atomicAdd(&tally[threadIdx.x], acc[threadIdx.x]);
}

__global__ void collect_failed2_kernel(int64_t B, const uint8_t *__restrict__ conv, int32_t *__restrict__ list, int32_t *__restrict__ count) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && !conv[b]) list[atomicAdd(count, 1)] = (int32_t)b;
}

}  // namespace qldpc

using namespace qldpc;

struct qldpc_circuit_plan {
    const qldpc_graph *gz = nullptr, *gx = nullptr;
    int device = 0, k = 0, n_locs = 0, nsx = 0, nsz = 0, max_iter = 0, use_osd = 0, flags = 0;
    double p = 0, damping = 1, clip = 20;
    uint32_t thr = 0;
    int64_t batch = 0;
    bool nanfree = false;
    std::vector<double> h_prior_z, h_prior_x;      // the priors on the host: what lets the decode dispatch pick the LDS-resident workgroup kernel (minsum_wg2.hip)
    DevBuf d_loc_type, d_zptr, d_zidx, d_zlog, d_xptr, d_xidx, d_xlog;
    DevBuf d_alpha_z, d_alpha_x, d_prior_z, d_prior_x, d_lm_z, d_lm_x;
    DevBuf d_syn_z, d_syn_x, d_true_z, d_true_x, d_det_z, d_det_x, d_llr_z, d_llr_x, d_conv_z, d_conv_x, d_iter_z, d_iter_x;
    DevBuf d_list_z, d_list_x, d_count, d_tally, d_outcome, d_clk;    // d_count: [0] Z failures, [4] X failures (int32, 16 bytes apart)
    // The two sectors are independent between the sampler and the judge: sector X runs on the plan's own stream so its decode fills the CUs the
    // tail of sector Z's OSD launch leaves idle (and vice versa); QLDPC_FLAG_MC_UNFUSED keeps everything on the caller's stream.
    hipStream_t side = nullptr;
    hipEvent_t ev_sampled = nullptr, ev_x_done = nullptr;
    // hipEvent brackets of the phases of every batch not yet read by qldpc_circuit_plan_phase_times
    struct Bracket { int phase; hipEvent_t a, b; };
    std::vector<Bracket> pending;
    std::vector<hipEvent_t> pool;
    double phase_ms[QLDPC_CIRCUIT_PHASES] = {0, 0, 0, 0, 0, 0};
    int64_t batches = 0;
    std::vector<DevBuf *> all() {
        return {&d_loc_type, &d_zptr, &d_zidx, &d_zlog, &d_xptr, &d_xidx, &d_xlog, &d_alpha_z, &d_alpha_x, &d_prior_z, &d_prior_x, &d_lm_z, &d_lm_x,
                &d_syn_z, &d_syn_x, &d_true_z, &d_true_x, &d_det_z, &d_det_x, &d_llr_z, &d_llr_x, &d_conv_z, &d_conv_x, &d_iter_z, &d_iter_x,
                &d_list_z, &d_list_x, &d_count, &d_tally, &d_outcome, &d_clk};
    }
};

static hipEvent_t plan_event(qldpc_circuit_plan *P) {
    if (!P->pool.empty()) { hipEvent_t e = P->pool.back(); P->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
// brackets `phase` on stream s: call once before (open = true) and once after the launches
static int phase_mark(qldpc_circuit_plan *P, int phase, hipStream_t s, bool open) {
    hipEvent_t e = plan_event(P);
    if (!e) { set_error("hipEventCreate failed"); return QLDPC_ERR_HIP; }
    QLDPC_HIP_TRY(hipEventRecord(e, s));
    if (open) P->pending.push_back({phase, e, nullptr});
    else {
        for (auto it = P->pending.rbegin(); it != P->pending.rend(); ++it)
            if (it->phase == phase && it->b == nullptr) { it->b = e; return QLDPC_OK; }
        P->pool.push_back(e);
    }
    return QLDPC_OK;
}

// Folds finished phase brackets into phase_ms and hands their events back to the pool; every entry leaves `pending` before its events enter the
// pool (an entry is never in both).  wait = false: only brackets whose closing event has already completed (no host wait).
static void drain_phases(qldpc_circuit_plan *P, bool wait) {
    size_t keep = 0;
    for (size_t i = 0; i < P->pending.size(); i++) {
        const qldpc_circuit_plan::Bracket br = P->pending[i];
        if (!br.b) {                                       // an opener whose launches failed before the closing mark
            if (wait) P->pool.push_back(br.a); else P->pending[keep++] = br;
            continue;
        }
        const bool done = wait ? (hipEventSynchronize(br.b) == hipSuccess) : (hipEventQuery(br.b) == hipSuccess);
        if (!done && !wait) { P->pending[keep++] = br; continue; }
        float t = 0;
        if (done && hipEventElapsedTime(&t, br.a, br.b) == hipSuccess) P->phase_ms[br.phase] += t;
        P->pool.push_back(br.a); P->pool.push_back(br.b);
    }
    P->pending.resize(keep);
    (void)hipGetLastError();                               // hipEventQuery's hipErrorNotReady is not an error of the caller
}

template <class T>
static int up(DevBuf &b, const std::vector<T> &v) {
    int rc = b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (rc != QLDPC_OK) return rc;
    if (!v.empty()) QLDPC_HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return QLDPC_OK;
}

// Builds the per-(location, slot) signatures of one sector with the batched single-fault frame simulation.
static int build_signatures(const qldpc_circuit_desc *D, bool xsector, const std::vector<int32_t> &loc_op, const std::vector<int32_t> &ops,
                            const std::vector<int32_t> &q1, const std::vector<int32_t> &q2, std::vector<int32_t> &ptr, std::vector<uint16_t> &idx,
                            std::vector<uint64_t> &logm) {
    const int n_locs = (int)loc_op.size(), nl = 2 * n_locs, tq = D->total_qubits;
    const int32_t *spos = xsector ? D->z_syn_positions : D->x_syn_positions, *sptr = xsector ? D->z_syn_ptrs : D->x_syn_ptrs;
    const int nchk = xsector ? D->num_z_checks : D->num_x_checks;
    const int nsyn = sptr[nchk];
    const uint8_t *L = xsector ? D->Lz : D->Lx;
    std::vector<int32_t> lane_pos(nl), lane_q(nl);
    std::vector<int8_t> lane_after(nl);
    for (int l = 0; l < n_locs; l++) {
        const int i = loc_op[l], op = D->base_ops[i];
        for (int s = 0; s < 2; s++) {
            const int e = 2 * l + s;
            lane_pos[e] = i;
            lane_after[e] = (op == C_OP_PREP_X || op == C_OP_PREP_Z || op == C_OP_CNOT) ? 1 : 0;      // Meas: before; Idle: either
            lane_q[e] = (s == 0) ? D->base_q1[i] : (op == C_OP_CNOT ? D->base_q2[i] : -1);
        }
    }
    DevTmp dpos, dafter, dq, dops, dq1, dq2, dstate, dhist;
    int rc;
    const size_t len = ops.size();
    if ((rc = dpos.alloc(nl * 4)) || (rc = dafter.alloc(nl)) || (rc = dq.alloc(nl * 4)) || (rc = dops.alloc(len * 4)) || (rc = dq1.alloc(len * 4)) ||
        (rc = dq2.alloc(len * 4)) || (rc = dstate.alloc((size_t)tq * nl)) || (rc = dhist.alloc((size_t)nsyn * nl)))
        return rc;
    QLDPC_HIP_TRY(hipMemcpy(dpos.p, lane_pos.data(), nl * 4, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dafter.p, lane_after.data(), nl, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dq.p, lane_q.data(), nl * 4, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dops.p, ops.data(), len * 4, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dq1.p, q1.data(), len * 4, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dq2.p, q2.data(), len * 4, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(zero_now(dhist.p, (size_t)nsyn * nl));
    const unsigned grid = (unsigned)((nl + 255) / 256);
    if (xsector)
        hipLaunchKernelGGL(fault_signature_kernel<true>, dim3(grid), dim3(256), 0, nullptr, nl, dpos.as<int32_t>(), dafter.as<int8_t>(), dq.as<int32_t>(),
                           (int64_t)len, dops.as<int32_t>(), dq1.as<int32_t>(), dq2.as<int32_t>(), tq, nsyn, dstate.as<int8_t>(), dhist.as<int8_t>());
    else
        hipLaunchKernelGGL(fault_signature_kernel<false>, dim3(grid), dim3(256), 0, nullptr, nl, dpos.as<int32_t>(), dafter.as<int8_t>(), dq.as<int32_t>(),
                           (int64_t)len, dops.as<int32_t>(), dq1.as<int32_t>(), dq2.as<int32_t>(), tq, nsyn, dstate.as<int8_t>(), dhist.as<int8_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    std::vector<int8_t> hist((size_t)nsyn * nl), state((size_t)tq * nl);
    QLDPC_HIP_TRY(hipMemcpy(hist.data(), dhist.p, hist.size(), hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(state.data(), dstate.p, state.size(), hipMemcpyDeviceToHost));
    // detectors = XOR of consecutive RAW measurements of the same check (sparsify_syndrome_jit, kernels.py:356-380)
    ptr.assign(nl + 1, 0); idx.clear(); logm.assign(nl, 0);
    std::vector<int32_t> prev(nsyn, -1);
    for (int c = 0; c < nchk; c++)
        for (int i = sptr[c] + 1; i < sptr[c + 1]; i++) prev[spos[i]] = spos[i - 1];
    for (int e = 0; e < nl; e++) {
        if (lane_q[e] >= 0)
            for (int s = 0; s < nsyn; s++) {
                int v = hist[(size_t)s * nl + e];
                if (prev[s] >= 0) v ^= hist[(size_t)prev[s] * nl + e];
                if (v & 1) idx.push_back((uint16_t)s);
            }
        ptr[e + 1] = (int32_t)idx.size();
        uint64_t lm = 0;
        if (lane_q[e] >= 0)
            for (int r = 0; r < D->k; r++) {                          // true logical = L @ data_state (simulation.py:80-81, 98-99)
                int s = 0;
                for (int j = 0; j < D->n_data; j++) s ^= (L[(size_t)r * D->n_data + j] & state[(size_t)D->data_qubit_indices[j] * nl + e] & 1);
                if (s) lm |= (uint64_t)1 << r;
            }
        logm[e] = lm;
    }
    return QLDPC_OK;
}

static int validate_desc(const qldpc_circuit_desc *D) {
    QLDPC_REQUIRE(D != nullptr, "circuit descriptor is NULL");
    QLDPC_REQUIRE(D->base_len >= 0 && D->suffix_len >= 0 && D->total_qubits > 0, "bad circuit sizes");
    QLDPC_REQUIRE(D->base_ops && D->base_q1 && D->base_q2 && (D->suffix_len == 0 || (D->suffix_ops && D->suffix_q1 && D->suffix_q2)), "NULL op array");
    QLDPC_REQUIRE(D->x_syn_positions && D->x_syn_ptrs && D->z_syn_positions && D->z_syn_ptrs && D->data_qubit_indices && D->Lx && D->Lz, "NULL table");
    QLDPC_REQUIRE(D->k >= 0 && D->k <= 64, "k out of range (0..64)");
    for (int pass = 0; pass < 2; pass++) {
        const int64_t len = pass ? D->suffix_len : D->base_len;
        const int32_t *o = pass ? D->suffix_ops : D->base_ops, *a = pass ? D->suffix_q1 : D->base_q1, *c = pass ? D->suffix_q2 : D->base_q2;
        for (int64_t i = 0; i < len; i++) {
            QLDPC_REQUIRE(o[i] >= C_OP_CNOT && o[i] <= C_OP_IDLE, "op %d at %lld is not a base-circuit gate", o[i], (long long)i);
            QLDPC_REQUIRE(a[i] >= 0 && a[i] < D->total_qubits, "q1 out of range at op %lld", (long long)i);
            QLDPC_REQUIRE(o[i] != C_OP_CNOT || (c[i] >= 0 && c[i] < D->total_qubits), "q2 out of range at op %lld", (long long)i);
        }
    }
    for (int j = 0; j < D->n_data; j++) QLDPC_REQUIRE(D->data_qubit_indices[j] >= 0 && D->data_qubit_indices[j] < D->total_qubits, "data qubit index out of range");
    return QLDPC_OK;
}

// Single-fault signatures of one sector (what the reference's builder simulates fault by fault, src/noise/builder.py:37-66).
// Entry e = 2 * base_op_index + slot (slot 0 = the gate's first qubit, slot 1 = the CNOT target).  ptr int32[2*base_len+1],
// idx uint16[idx_cap] (detector indices, ascending per entry), logmask uint64[2*base_len].  *idx_needed receives the total
// number of detector entries; QLDPC_ERR_INVALID if idx_cap is too small (call again with a larger buffer).
QLDPC_EXPORT int qldpc_circuit_fault_signatures(const qldpc_circuit_desc *D, int sector_is_x, int32_t *ptr, uint16_t *idx, int64_t idx_cap,
                                                uint64_t *logmask, int64_t *idx_needed) {
    int rc = validate_desc(D);
    if (rc != QLDPC_OK) return rc;
    QLDPC_REQUIRE(ptr && logmask && idx_needed && (idx || idx_cap == 0), "NULL output");
    QLDPC_USE_DEVICE(0);
    std::vector<int32_t> ops(D->base_ops, D->base_ops + D->base_len), q1(D->base_q1, D->base_q1 + D->base_len), q2(D->base_q2, D->base_q2 + D->base_len);
    ops.insert(ops.end(), D->suffix_ops, D->suffix_ops + D->suffix_len);
    q1.insert(q1.end(), D->suffix_q1, D->suffix_q1 + D->suffix_len);
    q2.insert(q2.end(), D->suffix_q2, D->suffix_q2 + D->suffix_len);
    std::vector<int32_t> loc_op(D->base_len);
    for (int64_t i = 0; i < D->base_len; i++) loc_op[i] = (int32_t)i;
    std::vector<int32_t> p;
    std::vector<uint16_t> ix;
    std::vector<uint64_t> lm;
    if ((rc = build_signatures(D, sector_is_x != 0, loc_op, ops, q1, q2, p, ix, lm)) != QLDPC_OK) return rc;
    *idx_needed = (int64_t)ix.size();
    QLDPC_REQUIRE((int64_t)ix.size() <= idx_cap, "idx buffer too small: need %lld entries", (long long)ix.size());
    std::memcpy(ptr, p.data(), p.size() * sizeof(int32_t));
    if (!ix.empty()) std::memcpy(idx, ix.data(), ix.size() * sizeof(uint16_t));
    std::memcpy(logmask, lm.data(), lm.size() * sizeof(uint64_t));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_circuit_plan_create(const qldpc_circuit_desc *D, const qldpc_graph *gz, const qldpc_graph *gx, const double *prior_z,
                                           const double *prior_x, const uint64_t *logmask_z, const uint64_t *logmask_x, double p, int max_iter,
                                           int alpha_mode, double alpha_val_z, double alpha_val_x, const double *alpha_seq_z, int alpha_len_z,
                                           const double *alpha_seq_x, int alpha_len_x, double damping, double clip_llr, int use_osd, int flags,
                                           int64_t batch, qldpc_circuit_plan **out) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int rc = validate_desc(D);
    if (rc != QLDPC_OK) return rc;
    QLDPC_REQUIRE(gz && gx && prior_z && prior_x && logmask_z && logmask_x, "NULL argument");
    QLDPC_REQUIRE(p > 0.0 && p < 1.0, "error rate must be in (0,1)");
    QLDPC_REQUIRE(batch > 0 && batch <= (1 << 24), "batch out of range");
    QLDPC_REQUIRE(gz->device == gx->device, "both sector graphs must live on the same device");
    const int nsx = D->x_syn_ptrs[D->num_x_checks], nsz = D->z_syn_ptrs[D->num_z_checks];
    QLDPC_REQUIRE(gz->m == nsx && gx->m == nsz, "decoding matrices have %d / %d rows but the circuit measures %d X / %d Z syndromes", gz->m, gx->m, nsx, nsz);
    QLDPC_REQUIRE(nsx < 65536 && nsz < 65536, "too many detectors for 16-bit signature indices");
    QLDPC_USE_DEVICE(gz->device);

    std::vector<int32_t> ops(D->base_ops, D->base_ops + D->base_len), q1(D->base_q1, D->base_q1 + D->base_len), q2(D->base_q2, D->base_q2 + D->base_len);
    ops.insert(ops.end(), D->suffix_ops, D->suffix_ops + D->suffix_len);
    q1.insert(q1.end(), D->suffix_q1, D->suffix_q1 + D->suffix_len);
    q2.insert(q2.end(), D->suffix_q2, D->suffix_q2 + D->suffix_len);
    std::vector<int32_t> loc_op;               // every base op is an error location (kernels.py:206-351 visits them in order)
    std::vector<uint8_t> loc_type;
    for (int64_t i = 0; i < D->base_len; i++) { loc_op.push_back((int32_t)i); loc_type.push_back((uint8_t)D->base_ops[i]); }

    qldpc_circuit_plan *P = new qldpc_circuit_plan();
    auto fail = [&](int code) { qldpc_circuit_plan_destroy(P); return code; };
    P->gz = gz; P->gx = gx; P->device = gz->device; P->k = D->k; P->n_locs = (int)loc_op.size(); P->nsx = nsx; P->nsz = nsz;
    P->max_iter = max_iter; P->use_osd = use_osd; P->flags = flags; P->p = p; P->damping = damping; P->clip = clip_llr; P->batch = batch;
    P->thr = bernoulli_threshold(p);
    std::vector<double> az, ax;
    if ((rc = build_alpha_table(max_iter, alpha_mode, alpha_val_z, alpha_seq_z, alpha_len_z, az)) != QLDPC_OK) return fail(rc);
    if ((rc = build_alpha_table(max_iter, alpha_mode, alpha_val_x, alpha_seq_x, alpha_len_x, ax)) != QLDPC_OK) return fail(rc);
    // "clean" inputs (finite, no -0.0 priors, positive finite clip / alphas) select the lean kernel; graphs with degree-1 checks
    // (+-inf messages) still keep the NaN test of kernels.py:328 inside it
    P->nanfree = inputs_clean(prior_z, gz->n, clip_llr, az.data(), max_iter) && inputs_clean(prior_x, gx->n, clip_llr, ax.data(), max_iter);
    std::vector<int32_t> zp, xp;
    std::vector<uint16_t> zi, xi;
    std::vector<uint64_t> zl, xl;
    if ((rc = build_signatures(D, false, loc_op, ops, q1, q2, zp, zi, zl)) != QLDPC_OK) return fail(rc);
    if ((rc = build_signatures(D, true, loc_op, ops, q1, q2, xp, xi, xl)) != QLDPC_OK) return fail(rc);
    if ((rc = up(P->d_loc_type, loc_type)) || (rc = up(P->d_zptr, zp)) || (rc = up(P->d_zidx, zi)) || (rc = up(P->d_zlog, zl)) || (rc = up(P->d_xptr, xp)) ||
        (rc = up(P->d_xidx, xi)) || (rc = up(P->d_xlog, xl)) || (rc = up(P->d_alpha_z, az)) || (rc = up(P->d_alpha_x, ax)))
        return fail(rc);
    std::vector<double> pz(prior_z, prior_z + gz->n), px(prior_x, prior_x + gx->n);
    P->h_prior_z = pz; P->h_prior_x = px;
    std::vector<uint64_t> lz(logmask_z, logmask_z + gz->n), lx(logmask_x, logmask_x + gx->n);
    if ((rc = up(P->d_prior_z, pz)) || (rc = up(P->d_prior_x, px)) || (rc = up(P->d_lm_z, lz)) || (rc = up(P->d_lm_x, lx))) return fail(rc);
    const size_t Bz = (size_t)batch;
    if ((rc = P->d_syn_z.ensure(Bz * nsx)) || (rc = P->d_syn_x.ensure(Bz * nsz)) || (rc = P->d_true_z.ensure(Bz * 8)) || (rc = P->d_true_x.ensure(Bz * 8)) ||
        (rc = P->d_det_z.ensure(Bz * gz->n)) || (rc = P->d_det_x.ensure(Bz * gx->n)) || (rc = P->d_llr_z.ensure(Bz * gz->n * 8)) ||
        (rc = P->d_llr_x.ensure(Bz * gx->n * 8)) || (rc = P->d_conv_z.ensure(Bz)) || (rc = P->d_conv_x.ensure(Bz)) || (rc = P->d_iter_z.ensure(Bz * 4)) ||
        (rc = P->d_iter_x.ensure(Bz * 4)) || (rc = P->d_list_z.ensure(Bz * 4)) || (rc = P->d_list_x.ensure(Bz * 4)) || (rc = P->d_count.ensure(64)) ||
        (rc = P->d_tally.ensure(QLDPC_TALLY_SLOTS * 8)) || (rc = P->d_clk.ensure(2 * kClkSlots * 16)))
        return fail(rc);
    if (zero_now(P->d_clk.p, 2 * kClkSlots * 16) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
    // (round 2 ran sector X on a second stream beside sector Z: two persistent kernels that each own every CU do not overlap -- 177.9 vs 178.7 ms
    // per step -- and the small launches queued behind them polluted the profile; everything runs on the caller's stream now)
    if (zero_now(P->d_tally.p, QLDPC_TALLY_SLOTS * 8) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
    *out = P;
    return QLDPC_OK;
}

static int launch_sampler(qldpc_circuit_plan *P, uint64_t seed, int64_t begin, int64_t B, hipStream_t s, bool zero_counts = false) {
    SigTab Z{P->d_zptr.as<int32_t>(), P->d_zidx.as<uint16_t>(), P->d_zlog.as<uint64_t>()};
    SigTab X{P->d_xptr.as<int32_t>(), P->d_xidx.as<uint16_t>(), P->d_xlog.as<uint64_t>()};
    const int wz = (P->nsx + 31) / 32, wx = (P->nsz + 31) / 32;
    const size_t lds = (size_t)((wz + wx + 1) & ~1) * 4 + 16;
    const unsigned grid = (unsigned)std::min<int64_t>(B, 256 * 16);
    hipLaunchKernelGGL(circuit_sample_kernel, dim3(grid), dim3(256), lds, s, B, begin, (uint32_t)seed, (uint32_t)(seed >> 32), P->thr, P->n_locs,
                       P->d_loc_type.as<uint8_t>(), Z, X, P->nsx, P->nsz, P->d_syn_z.as<int8_t>(), P->d_syn_x.as<int8_t>(),
                       P->d_true_z.as<unsigned long long>(), P->d_true_x.as<unsigned long long>(), zero_counts ? P->d_count.as<int32_t>() : (int32_t *)nullptr);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

static int decode_sector(qldpc_circuit_plan *P, const qldpc_graph *g, int64_t B, DevBuf &syn, DevBuf &prior, DevBuf &alpha, DevBuf &det, DevBuf &llr,
                         DevBuf &conv, DevBuf &iter, DevBuf &list, int sector, hipStream_t s) {
    const std::vector<double> &hp = sector ? P->h_prior_x : P->h_prior_z;
    int rc;
    int32_t *count = P->d_count.as<int32_t>() + 4 * sector;
    const int ph_bp = sector ? QLDPC_PHASE_BP_X : QLDPC_PHASE_BP_Z, ph_osd = sector ? QLDPC_PHASE_OSD_X : QLDPC_PHASE_OSD_Z;
    unsigned long long *clk = (P->flags & QLDPC_FLAG_CLOCK_PROBE) ? P->d_clk.as<unsigned long long>() : nullptr;
    if ((rc = phase_mark(P, ph_bp, s, true)) != QLDPC_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->clk_probe = (clk && sector == 0) ? clk : nullptr;                  // sector Z carries the probe (one writer per buffer)
        rc = minsum_decode_dispatch(g, B, syn.as<int8_t>(), prior.as<double>(), P->max_iter, alpha.as<double>(), P->damping, P->clip,
                                    (P->flags & QLDPC_FLAG_PUBLIC_MASK) | (P->nanfree ? QLDPC_FLAG_INTERNAL_PRIOR_FINITE : 0),
                                    P->nanfree, det.as<int8_t>(), llr.as<double>(), conv.as<uint8_t>(), iter.as<int32_t>(), s, hp.empty() ? nullptr : hp.data());
        g->clk_probe = nullptr;
    }
    if (rc != QLDPC_OK) return rc;
    if ((rc = phase_mark(P, ph_bp, s, false)) != QLDPC_OK || !P->use_osd) return rc;
    if ((rc = phase_mark(P, ph_osd, s, true)) != QLDPC_OK) return rc;
    hipLaunchKernelGGL(collect_failed2_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, conv.as<uint8_t>(), list.as<int32_t>(), count);
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->clk_probe = (clk && sector == 0) ? clk + 2 * kClkSlots : nullptr;
        rc = osd0_listed_launch(g, list.as<int32_t>(), count, B, syn.as<int8_t>(), llr.as<double>(), det.as<int8_t>(), nullptr,
                                det.as<int8_t>(), P->flags, s);
        g->clk_probe = nullptr;
    }
    if (rc != QLDPC_OK) return rc;
    QLDPC_HIP_TRY(hipGetLastError());
    return phase_mark(P, ph_osd, s, false);                      // (the judge kernel adds the two failure counts to the OSD tally slots)
}

// one pass over [trial_begin, trial_begin + count); `outcome` (host, may be NULL) receives bit0 = z_err, bit1 = x_err per trial
static int circuit_run(qldpc_circuit_plan *P, uint64_t seed, int64_t trial_begin, int64_t count, hipStream_t s, uint8_t *outcome) {
    QLDPC_USE_DEVICE(P->device);
    int rc = QLDPC_OK; (void)rc;
    if (outcome && (rc = P->d_outcome.ensure((size_t)P->batch)) != QLDPC_OK) return rc;
    for (int64_t off = 0; off < count; off += P->batch) {
        const int64_t B = std::min<int64_t>(P->batch, count - off);
        if (P->pending.size() > 256) drain_phases(P, false);         // a caller that never reads phase times: recycle finished brackets (bounded event count)
        if ((rc = phase_mark(P, QLDPC_PHASE_SAMPLE, s, true)) != QLDPC_OK) return rc;
        if ((rc = launch_sampler(P, seed, trial_begin + off, B, s, true)) != QLDPC_OK) return rc;
        if ((rc = phase_mark(P, QLDPC_PHASE_SAMPLE, s, false)) != QLDPC_OK) return rc;
        hipStream_t sx = s;
        if (P->side && P->gz != P->gx) {                // sector X on the plan's own stream, joined again before the judge
            sx = P->side;
            QLDPC_HIP_TRY(hipEventRecord(P->ev_sampled, s));
            QLDPC_HIP_TRY(hipStreamWaitEvent(sx, P->ev_sampled, 0));
        }
        if ((rc = decode_sector(P, P->gz, B, P->d_syn_z, P->d_prior_z, P->d_alpha_z, P->d_det_z, P->d_llr_z, P->d_conv_z, P->d_iter_z, P->d_list_z, 0, s)) != QLDPC_OK) return rc;
        if ((rc = decode_sector(P, P->gx, B, P->d_syn_x, P->d_prior_x, P->d_alpha_x, P->d_det_x, P->d_llr_x, P->d_conv_x, P->d_iter_x, P->d_list_x, 1, sx)) != QLDPC_OK) return rc;
        if (sx != s) {
            QLDPC_HIP_TRY(hipEventRecord(P->ev_x_done, sx));
            QLDPC_HIP_TRY(hipStreamWaitEvent(s, P->ev_x_done, 0));
        }
        if ((rc = phase_mark(P, QLDPC_PHASE_JUDGE, s, true)) != QLDPC_OK) return rc;
        JudgeSector Z{P->gz->m, P->gz->n, P->gz->d_indptr, P->gz->d_indices, P->gz->d_colptr, P->gz->d_rowidx, P->d_lm_z.as<uint64_t>(), P->d_syn_z.as<int8_t>(), P->d_det_z.as<int8_t>(),
                      P->d_conv_z.as<uint8_t>(), P->d_iter_z.as<int32_t>(), P->d_true_z.as<unsigned long long>()};
        JudgeSector X{P->gx->m, P->gx->n, P->gx->d_indptr, P->gx->d_indices, P->gx->d_colptr, P->gx->d_rowidx, P->d_lm_x.as<uint64_t>(), P->d_syn_x.as<int8_t>(), P->d_det_x.as<int8_t>(),
                      P->d_conv_x.as<uint8_t>(), P->d_iter_x.as<int32_t>(), P->d_true_x.as<unsigned long long>()};
        if (P->gz->m <= 4096 && P->gx->m <= 4096 && P->gz->d_colptr && P->gx->d_colptr)
            hipLaunchKernelGGL(circuit_judge_kernel<true>, dim3((unsigned)((B + 7) / 8)), dim3(256), 0, s, B, Z, X, P->d_tally.as<unsigned long long>(),
                               outcome ? P->d_outcome.as<uint8_t>() : (uint8_t *)nullptr, P->use_osd ? P->d_count.as<int32_t>() : (const int32_t *)nullptr);
        else
            hipLaunchKernelGGL(circuit_judge_kernel<false>, dim3((unsigned)((B + 7) / 8)), dim3(256), 0, s, B, Z, X, P->d_tally.as<unsigned long long>(),
                               outcome ? P->d_outcome.as<uint8_t>() : (uint8_t *)nullptr, P->use_osd ? P->d_count.as<int32_t>() : (const int32_t *)nullptr);
        QLDPC_HIP_TRY(hipGetLastError());
        if ((rc = phase_mark(P, QLDPC_PHASE_JUDGE, s, false)) != QLDPC_OK) return rc;
        P->batches++;           // (the next batch's sampler follows the judge on s, which already waited for sector X)
        if (outcome) {
            QLDPC_HIP_TRY(hipMemcpyAsync(outcome + off, P->d_outcome.p, (size_t)B, hipMemcpyDeviceToHost, s));
            QLDPC_HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_circuit_plan_run(qldpc_circuit_plan *P, uint64_t seed, int64_t trial_begin, int64_t count, void *stream) {
    QLDPC_REQUIRE(P != nullptr, "plan is NULL");
    QLDPC_REQUIRE(count >= 0 && trial_begin >= 0, "negative trial range");
    return circuit_run(P, seed, trial_begin, count, reinterpret_cast<hipStream_t>(stream), nullptr);
}

QLDPC_EXPORT int qldpc_circuit_plan_run_outcomes(qldpc_circuit_plan *P, uint64_t seed, int64_t trial_begin, int64_t count, void *stream,
                                                 uint8_t *outcome) {
    QLDPC_REQUIRE(P != nullptr, "plan is NULL");
    QLDPC_REQUIRE(count >= 0 && trial_begin >= 0, "negative trial range");
    QLDPC_REQUIRE(count == 0 || outcome != nullptr, "outcome is NULL");
    return circuit_run(P, seed, trial_begin, count, reinterpret_cast<hipStream_t>(stream), outcome);
}

QLDPC_EXPORT int qldpc_circuit_plan_read(qldpc_circuit_plan *P, void *stream, int clear, int64_t *tally) {
    QLDPC_REQUIRE(P != nullptr && tally != nullptr, "NULL argument");
    QLDPC_USE_DEVICE(P->device);
    int rc = QLDPC_OK; (void)rc;
    QLDPC_HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    if (P->side) QLDPC_HIP_TRY(hipStreamSynchronize(P->side));
    drain_phases(P, true);                                           // everything enqueued has finished: fold the brackets, recycle their events
    QLDPC_HIP_TRY(hipMemcpy(tally, P->d_tally.p, QLDPC_TALLY_SLOTS * 8, hipMemcpyDeviceToHost));
    if (clear) QLDPC_HIP_TRY(zero_now(P->d_tally.p, QLDPC_TALLY_SLOTS * 8));
    return QLDPC_OK;
}

// batched run_trial_fast: sparse_z int8[count][nsx], true_z int8[count][k], sparse_x int8[count][nsz], true_x int8[count][k] (host)
QLDPC_EXPORT int qldpc_circuit_plan_sample(qldpc_circuit_plan *P, uint64_t seed, int64_t trial_begin, int64_t count, int8_t *sparse_z,
                                           int8_t *true_z, int8_t *sparse_x, int8_t *true_x) {
    QLDPC_REQUIRE(P != nullptr, "plan is NULL");
    QLDPC_REQUIRE(count >= 0 && trial_begin >= 0, "negative trial range");
    QLDPC_REQUIRE(count == 0 || (sparse_z && true_z && sparse_x && true_x), "NULL output");
    QLDPC_USE_DEVICE(P->device);
    int rc = QLDPC_OK; (void)rc;
    std::vector<unsigned long long> tz, tx;
    for (int64_t off = 0; off < count; off += P->batch) {
        const int64_t B = std::min<int64_t>(P->batch, count - off);
        if ((rc = launch_sampler(P, seed, trial_begin + off, B, nullptr)) != QLDPC_OK) return rc;
        QLDPC_HIP_TRY(hipDeviceSynchronize());
        QLDPC_HIP_TRY(hipMemcpy(sparse_z + off * P->nsx, P->d_syn_z.p, (size_t)B * P->nsx, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(sparse_x + off * P->nsz, P->d_syn_x.p, (size_t)B * P->nsz, hipMemcpyDeviceToHost));
        tz.resize(B); tx.resize(B);
        QLDPC_HIP_TRY(hipMemcpy(tz.data(), P->d_true_z.p, (size_t)B * 8, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(tx.data(), P->d_true_x.p, (size_t)B * 8, hipMemcpyDeviceToHost));
        for (int64_t b = 0; b < B; b++)
            for (int r = 0; r < P->k; r++) {
                true_z[(off + b) * P->k + r] = (int8_t)((tz[b] >> r) & 1);
                true_x[(off + b) * P->k + r] = (int8_t)((tx[b] >> r) & 1);
            }
    }
    return QLDPC_OK;
}

// Sums (ms) of the hipEvent brackets of each phase over the batches enqueued since the last call, and the number of batches.  With the
// two sectors on two streams the brackets overlap in time: their sum exceeds the wall time, each is the span of that phase on its stream.
QLDPC_EXPORT int qldpc_circuit_plan_phase_times(qldpc_circuit_plan *P, double *ms, int64_t *batches) {
    QLDPC_REQUIRE(P != nullptr && ms != nullptr, "NULL argument");
    QLDPC_USE_DEVICE(P->device);
    drain_phases(P, true);
    for (int i = 0; i < QLDPC_CIRCUIT_PHASES; i++) { ms[i] = P->phase_ms[i]; P->phase_ms[i] = 0; }
    if (batches) *batches = P->batches;
    P->batches = 0;
    return QLDPC_OK;
}

// Shader clock (MHz) held while the decode kernel [0] and the OSD-0 kernel [1] of sector Z ran in the last batch: median over workgroups of
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz.  Needs QLDPC_FLAG_CLOCK_PROBE at plan creation; 0 where nothing was stamped.
QLDPC_EXPORT int qldpc_circuit_plan_clock(qldpc_circuit_plan *P, void *stream, double *mhz) {
    QLDPC_REQUIRE(P != nullptr && mhz != nullptr, "NULL argument");
    QLDPC_REQUIRE(P->flags & QLDPC_FLAG_CLOCK_PROBE, "the plan was created without QLDPC_FLAG_CLOCK_PROBE");
    QLDPC_USE_DEVICE(P->device);
    QLDPC_HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    if (P->side) QLDPC_HIP_TRY(hipStreamSynchronize(P->side));
    std::vector<unsigned long long> h(4 * kClkSlots);
    QLDPC_HIP_TRY(hipMemcpy(h.data(), P->d_clk.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 2; k++) mhz[k] = clock_probe_median(h.data() + 2 * kClkSlots * k, kClkSlots);
    return QLDPC_OK;
}

QLDPC_EXPORT void qldpc_circuit_plan_destroy(qldpc_circuit_plan *P) {
    if (!P) return;
    (void)hipSetDevice(P->device);
    if (P->side) { (void)hipStreamSynchronize(P->side); (void)hipStreamDestroy(P->side); }
    if (P->ev_sampled) (void)hipEventDestroy(P->ev_sampled);
    if (P->ev_x_done) (void)hipEventDestroy(P->ev_x_done);
    for (auto &br : P->pending) { (void)hipEventDestroy(br.a); if (br.b) (void)hipEventDestroy(br.b); }
    for (auto e : P->pool) (void)hipEventDestroy(e);
    for (DevBuf *b : P->all()) b->release();
    delete P;
}
