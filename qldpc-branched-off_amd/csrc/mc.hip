// Code-capacity Monte-Carlo pipeline around the decoder: Philox sampler, GF(2) syndrome SpMV (a6),
// failed-shot collection for OSD-0, logical-error judge and the int64 tally (replaces the Python tally loop
// src/simulation/engine.py:450-457).  Sampling law: src/decoding/alpha.py:127-128; failure rule: engine.py:99-100.
#include "common.h"
#include "mc_common.h"
#include "minsum_common.h"

#include <cmath>
#include <vector>

namespace qldpc {

// ---- sampler: one thread per (shot, 4-variable block); bit j is an error iff Philox word < thr ----
__global__ void cc_sample_kernel(int64_t count, int64_t shot_begin, uint32_t seed_lo, uint32_t seed_hi, int n, uint32_t thr,
                                 int8_t *__restrict__ err) {
    const int nq = (n + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * nq) return;
    const int64_t b = t / nq;
    const int q = (int)(t - b * nq);
    const uint64_t g = (uint64_t)(shot_begin + b);
    uint32_t o[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)q, 0u, seed_lo, seed_hi, o);
    int8_t *e = err + b * n + 4 * q;
#pragma unroll
    for (int w = 0; w < 4; w++)
        if (4 * q + w < n) e[w] = (o[w] < thr) ? 1 : 0;
}

// ---- a6: s_i = XOR_{j in row i} e_j ; one thread per (shot, row) (kernels.py:222-231) ----
__global__ void gf2_spmv_kernel(int64_t B, int m, int n, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                const int8_t *__restrict__ vec, int8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * m) return;
    const int64_t b = t / m;
    const int i = (int)(t - b * m);
    const int8_t *v = vec + b * n;
    int s = 0;
    for (int e = indptr[i]; e < indptr[i + 1]; e++) s ^= v[indices[e]];
    out[t] = (int8_t)(s & 1);
}

// ---- list the shots BP did not converge on (input of the OSD-0 stage) ----
__global__ void collect_failed_kernel(int64_t B, const uint8_t *__restrict__ conv, int32_t *__restrict__ list, int32_t *__restrict__ count) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (!conv[b]) list[atomicAdd(count, 1)] = (int32_t)b;
}

// ---- judge: 32 lanes per shot; logical failure iff L (e xor e_hat) != 0; unsat iff H e_hat != s ----
__global__ __launch_bounds__(256) void cc_judge_kernel(int64_t B, int m, int n, const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices, const uint64_t *__restrict__ Lmask,
                                                       const int8_t *__restrict__ err, const int8_t *__restrict__ synd,
                                                       const int8_t *__restrict__ dec, const uint8_t *__restrict__ conv,
                                                       const int32_t *__restrict__ iters, unsigned long long *__restrict__ tally) {
    __shared__ unsigned long long acc[6];
    if (threadIdx.x < 6) acc[threadIdx.x] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 31;
    const int64_t b = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (b < B) {
        const int8_t *e = err + b * n, *d = dec + b * n, *s = synd + b * m;
        uint64_t lm = 0;
        for (int j = lane; j < n; j += 32)
            if ((e[j] ^ d[j]) & 1) lm ^= Lmask[j];
        int bad = 0, nz = 0;
        for (int i = lane; i < m; i += 32) {
            int p = 0;
            for (int k = indptr[i]; k < indptr[i + 1]; k++) p ^= d[indices[k]];
            bad |= ((p ^ s[i]) & 1);
            nz |= (s[i] & 1);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            lm ^= __shfl_xor(lm, off, 32);
            bad |= __shfl_xor(bad, off, 32);
            nz |= __shfl_xor(nz, off, 32);
        }
        if (lane == 0) {
            if (lm != 0) atomicAdd(&acc[0], 1ull);
            if (conv[b]) atomicAdd(&acc[1], 1ull);
            atomicAdd(&acc[2], (unsigned long long)(iters[b] + 1));
            if (!nz) atomicAdd(&acc[3], 1ull);
            if (bad) atomicAdd(&acc[4], 1ull);
            atomicAdd(&acc[5], 1ull);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (acc[0]) { atomicAdd(&tally[QLDPC_TALLY_Z_ERR], acc[0]); atomicAdd(&tally[QLDPC_TALLY_TOTAL_ERR], acc[0]); }
        if (acc[1]) atomicAdd(&tally[QLDPC_TALLY_BP_CONV_Z], acc[1]);
        if (acc[2]) atomicAdd(&tally[QLDPC_TALLY_ITERS_Z], acc[2]);
        if (acc[3]) atomicAdd(&tally[QLDPC_TALLY_ZERO_SYND_Z], acc[3]);
        if (acc[4]) atomicAdd(&tally[QLDPC_TALLY_UNSAT_Z], acc[4]);
        if (acc[5]) atomicAdd(&tally[QLDPC_TALLY_TRIALS], acc[5]);
    }
}

__global__ void add_osd_count_kernel(const int32_t *count, unsigned long long *tally) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&tally[QLDPC_TALLY_OSD_Z], (unsigned long long)*count);
}

int gf2_spmv_launch(const qldpc_graph *g, int64_t B, const int8_t *d_vec, int8_t *d_out, hipStream_t stream) {
    const int64_t total = B * g->m;
    if (total == 0) return QLDPC_OK;
    hipLaunchKernelGGL(gf2_spmv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, B, g->m, g->n, g->d_indptr,
                       g->d_indices, d_vec, d_out);
    QLDPC_HIP_TRY(hipGetLastError());
    return QLDPC_OK;
}

}  // namespace qldpc

using namespace qldpc;

// ------------------------------------------------------------------------------------------ plan
static const int64_t kFanBatch = 32768;
static const int kFanLanes = 8;
static int g_big_lanes = 3;
namespace qldpc { void mc_set_big_lanes(int n) { g_big_lanes = n; } }

struct qldpc_cc_plan {
    const qldpc_graph *g = nullptr;
    int k = 0, max_iter = 0, use_osd = 0, flags = 0;
    double p = 0, damping = 1, clip = 20;
    uint32_t thr = 0;
    int64_t batch = 0;
    std::vector<double> alpha;
    DevBuf d_alpha, d_prior, d_Lmask, d_err, d_synd, d_dec, d_llr, d_conv, d_iter, d_tally, d_list, d_count, d_sol, d_cold, d_clk;
    DevBuf d_lptr, d_lidx, d_cont;        // logical rows in CSR form and the list of shots the bit-sliced first iteration hands on (mc_first.hip)
    // Fused pipeline, failure records per LANE (lane 0 = the buffers above).  Two ways of keeping the chip busy between batches:
    //   large batches (2 lanes, reference semantics only): the tail of batch k (OSD-0 on its BP failures + their judge: a latency-bound ~0.1 ms on a
    //     handful of CUs) runs on the plan's side stream beside the decode of batch k + 1 on the caller's stream; the lanes double-buffer the records;
    //   small batches (batch <= kFanBatch, kFanLanes lanes): one launch of a few thousand shots cannot fill 256 CUs (BASELINE config 2 is quoted at
    //     batch 4096: 585 workgroups), so batch k runs ENTIRELY on lane k % kFanLanes' own stream and up to kFanLanes batches are in flight at once.
    //     The lanes start behind whatever the caller had enqueued on its stream; qldpc_cc_plan_read joins them.
    // (a fan-out lane owns a private copy of the graph handle: the OSD-0 workspaces hang off the handle and are handed over between streams in
    // order, which would serialise the tails of concurrent batches)
    struct Lane { DevBuf err, synd, dec, llr, list, count, cold, cont; hipStream_t st = nullptr; hipEvent_t decoded = nullptr, tail = nullptr; bool tail_pending = false;
                  qldpc_graph *g = nullptr; };
    std::vector<Lane> lanes;              // lane 0 uses d_err, d_synd, ... above
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr;
    bool fan = false;
    int64_t batch_no = 0;
    bool clk_first = false;               // the last launch stamped the first-iteration kernel's probe buffer
    bool first_ok = false;                // the closed form of iteration 0 applies to this plan (uniform prior > 0, column degree <= 3, ...)
    unsigned negbits = 0;
    bool fused = false, generic = false, nanfree = false, clean = false;      // generic: the fused kernel is the irregular-degree one (minsum_resident.hip)
    // (clean below)      // clean: nanfree and |prior| <= clip (what the wave-private kernel needs)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // decode-kernel brackets not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_first;   // ... and those of the first-iteration kernel alone (a part of the bracket above)
    double ms_first = 0;
    std::vector<hipEvent_t> pool;
    double ms_total = 0;
    int64_t launches = 0;
};

static hipEvent_t get_event(qldpc_cc_plan *P) {
    if (!P->pool.empty()) { hipEvent_t e = P->pool.back(); P->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

QLDPC_EXPORT int qldpc_cc_plan_create(const qldpc_graph *g, int k, const uint8_t *L, double p, int max_iter, int alpha_mode,
                                      double alpha_val, const double *alpha_seq, int alpha_len, double damping, double clip_llr,
                                      int use_osd, int flags, int64_t batch, int64_t min_launch, qldpc_cc_plan **out) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(k >= 0 && k <= 64, "k=%d logical rows unsupported (0..64)", k);
    QLDPC_REQUIRE(k == 0 || L != nullptr, "L is NULL");
    QLDPC_REQUIRE(p > 0.0 && p < 1.0, "error rate must be in (0,1)");
    QLDPC_REQUIRE(batch > 0 && batch <= ((int64_t)1 << 30), "batch out of range");
    QLDPC_REQUIRE(max_iter >= 0, "negative max_iter");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    QLDPC_REQUIRE(min_launch >= 0 && min_launch <= ((int64_t)1 << 30), "min_launch out of range");
    // `batch` is taken literally: a piece is three enqueues (decode, OSD-0, judge).  A caller that prefers fewer, larger launches to the memory bound
    // passes a granule for THIS plan (include/qldpc_hip.h); the tallies do not depend on the cut.
    if (batch < min_launch) batch = min_launch;
    qldpc_cc_plan *P = new qldpc_cc_plan();
    P->g = g; P->k = k; P->max_iter = max_iter; P->use_osd = use_osd; P->flags = flags;
    P->p = p; P->damping = damping; P->clip = clip_llr; P->batch = batch;
    P->thr = bernoulli_threshold(p);
    if ((rc = build_alpha_table(max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, P->alpha)) != QLDPC_OK) { delete P; return rc; }
    const size_t n = g->n, m = g->m;
    std::vector<double> prior(n ? n : 1, std::log((1.0 - p) / p));     // uniform prior log((1-p)/p), alpha.py:119-120
    std::vector<uint64_t> Lmask(n ? n : 1, 0);
    for (int r = 0; r < k; r++)
        for (size_t j = 0; j < n; j++)
            if (L[(size_t)r * n + j] & 1) Lmask[j] |= (uint64_t)1 << r;
    auto fail = [&](int code) { qldpc_cc_plan_destroy(P); return code; };
    {
        const double pr = std::log((1.0 - p) / p);
        P->nanfree = inputs_clean(&pr, 1, clip_llr, P->alpha.data(), max_iter);
        P->clean = P->nanfree && std::fabs(pr) <= clip_llr;
    }
    P->fused = !(flags & (QLDPC_FLAG_MC_UNFUSED | QLDPC_FLAG_KERNEL_STREAM | QLDPC_FLAG_KERNEL_GENERIC)) && damping == 1.0 &&
               regular_supported(g, clip_llr, max_iter);
    if (!P->fused && !(flags & (QLDPC_FLAG_MC_UNFUSED | QLDPC_FLAG_KERNEL_STREAM)) && damping == 1.0 && resident_supported(g, damping)) {
        P->fused = true; P->generic = true;      // small irregular graphs (Steane, BASELINE config 1), or a regular one asked through QLDPC_FLAG_KERNEL_GENERIC
    }
    if ((rc = P->d_alpha.ensure(P->alpha.size() * 8)) || (rc = P->d_prior.ensure(prior.size() * 8)) ||
        (rc = P->d_Lmask.ensure(Lmask.size() * 8)) || (rc = P->d_err.ensure(batch * n)) || (rc = P->d_synd.ensure(batch * m)) ||
        (rc = P->d_dec.ensure(batch * n)) || (rc = P->d_llr.ensure(batch * n * 8)) || (rc = P->d_conv.ensure(batch)) ||
        (rc = P->d_iter.ensure(batch * 4)) || (rc = P->d_tally.ensure(QLDPC_TALLY_SLOTS * 8)) ||
        (rc = P->d_list.ensure(batch * 4)) || (rc = P->d_count.ensure(16)))
        return fail(rc);
    if (zero_now(P->d_count.p, 16) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
    if (hipMemcpy(P->d_alpha.p, P->alpha.data(), P->alpha.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->d_prior.p, prior.data(), prior.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->d_Lmask.p, Lmask.data(), Lmask.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        zero_now(P->d_tally.p, QLDPC_TALLY_SLOTS * 8) != hipSuccess) {
        set_error("plan upload failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(QLDPC_ERR_HIP);
    }
    if (P->fused && !P->generic && !(flags & QLDPC_FLAG_FIXED_ITERS) && max_iter >= 1 && P->clean) {
        P->first_ok = mc_first_table(g, prior[0], P->alpha[0], clip_llr, max_iter, P->negbits) && (size_t)(g->m + g->n) * 64 <= 60 * 1024;
        if (P->first_ok) {
            std::vector<int32_t> lptr(k + 1, 0), lidx;
            for (int r = 0; r < k; r++) {
                for (size_t j = 0; j < n; j++) if (L[(size_t)r * n + j] & 1) lidx.push_back((int32_t)j);
                lptr[r + 1] = (int32_t)lidx.size();
            }
            if ((rc = P->d_lptr.ensure(lptr.size() * 4)) || (rc = P->d_lidx.ensure(std::max<size_t>(lidx.size(), 1) * 4)) || (rc = P->d_cont.ensure(batch * 4)))
                return fail(rc);
            if (hipMemcpy(P->d_lptr.p, lptr.data(), lptr.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                (!lidx.empty() && hipMemcpy(P->d_lidx.p, lidx.data(), lidx.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
                set_error("plan upload failed: %s", hipGetErrorString(hipGetLastError()));
                return fail(QLDPC_ERR_HIP);
            }
        }
    }
    if (P->fused) {
        if ((rc = P->d_cold.ensure(mc_regular_cold_bytes())) != QLDPC_OK) return fail(rc);
        if (flags & QLDPC_FLAG_CLOCK_PROBE) {
            // two probe buffers: [0] the full decoder's workgroups, [1] the first-iteration kernel's
            if ((rc = P->d_clk.ensure(4 * kClkSlots * 8)) != QLDPC_OK) return fail(rc);
            if (zero_now(P->d_clk.p, 4 * kClkSlots * 8) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
        }
        if ((rc = mc_regular_fill_cold(P->d_cold.p, P->d_tally.as<unsigned long long>(), P->d_count.as<int32_t>(), P->d_list.as<int32_t>(),
                                       P->d_synd.as<int8_t>(), P->d_err.as<int8_t>(), P->d_dec.as<int8_t>(), P->d_llr.as<double>(),
                                       (flags & QLDPC_FLAG_CLOCK_PROBE) ? P->d_clk.as<unsigned long long>() : nullptr)) != QLDPC_OK)
            return fail(rc);
        // (fixed-work plans with large batches keep everything on the caller's stream: beside a 13 ms decode launch the 0.1 ms tail gains nothing, and
        // measured concurrently it slows the persistent decode kernel by 5 %, profiles/r03_experiments.txt)
        if (mc_tail_overlap_choice() >= 1 && (!(flags & QLDPC_FLAG_FIXED_ITERS) || batch <= kFanBatch)) {
            // reference semantics with large batches: a batch is the first-iteration kernel (every CU, 0.18 ms) followed by latency-bound pieces -- the full
            // decoder on the few listed shots (its time is the 50 iterations of the slowest shot: ~0.11 ms), OSD-0 and the judge.  Whole batches on
            // a few (option mc_big_lanes, default 3) streams of their own let those pieces of batch k run beside the first-iteration kernel of batch k + 1 (option mc_tail_overlap = 2
            // keeps the round-3a form: decode on the caller's stream, only OSD-0 + judge on a side stream)
            const bool big_fan = !(flags & QLDPC_FLAG_FIXED_ITERS) && batch > kFanBatch && mc_tail_overlap_choice() == 1;
            P->fan = batch <= kFanBatch || big_fan;
            const int nl = P->fan ? (big_fan ? g_big_lanes : kFanLanes) : (use_osd ? 2 : 1);
            P->lanes.resize(nl);
            bool ok = true;
            for (int i = 1; i < nl; i++) {              // lane 0 = the plan's own buffers
                qldpc_cc_plan::Lane &Ln = P->lanes[i];
                if ((rc = Ln.err.ensure(batch * n)) || (rc = Ln.synd.ensure(batch * m)) || (rc = Ln.dec.ensure(batch * n)) || (rc = Ln.llr.ensure(batch * n * 8)) ||
                    (rc = Ln.list.ensure(batch * 4)) || (rc = Ln.count.ensure(16)) || (rc = Ln.cold.ensure(mc_regular_cold_bytes())) ||
                    (P->first_ok && (rc = Ln.cont.ensure(batch * 4))))
                    return fail(rc);
                if (zero_now(Ln.count.p, 16) != hipSuccess) { set_error("memset failed"); return fail(QLDPC_ERR_HIP); }
                if ((rc = mc_regular_fill_cold(Ln.cold.p, P->d_tally.as<unsigned long long>(), Ln.count.as<int32_t>(), Ln.list.as<int32_t>(), Ln.synd.as<int8_t>(),
                                               Ln.err.as<int8_t>(), Ln.dec.as<int8_t>(), Ln.llr.as<double>(),
                                               (flags & QLDPC_FLAG_CLOCK_PROBE) ? P->d_clk.as<unsigned long long>() : nullptr)) != QLDPC_OK)
                    return fail(rc);
            }
            for (int i = 0; i < nl && ok; i++) {
                qldpc_cc_plan::Lane &Ln = P->lanes[i];
                ok = hipEventCreateWithFlags(&Ln.decoded, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&Ln.tail, hipEventDisableTiming) == hipSuccess;
                if (ok && P->fan) ok = hipStreamCreateWithFlags(&Ln.st, hipStreamNonBlocking) == hipSuccess;
                // a lane owns a private copy of the graph handle (its OSD-0 workspaces are then used from the lane's stream alone: no hand-over events)
                if (ok && P->fan && use_osd) {
                    if (qldpc_graph_create(g->m, g->n, g->indptr.data(), g->indices.data(), g->device, &Ln.g) != QLDPC_OK) return fail(QLDPC_ERR_HIP);
                    Ln.g->ws_private = true;
                }
            }
            if (ok && !P->fan && nl == 2) ok = hipStreamCreateWithFlags(&P->side, hipStreamNonBlocking) == hipSuccess;
            if (ok && P->fan) ok = hipEventCreateWithFlags(&P->ev_in, hipEventDisableTiming) == hipSuccess;
            if (!ok) { set_error("stream / event creation failed: %s", hipGetErrorString(hipGetLastError())); return fail(QLDPC_ERR_HIP); }
        }
    }
    *out = P;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_cc_plan_run(qldpc_cc_plan *P, uint64_t seed, int64_t shot_begin, int64_t count, void *stream) {
    QLDPC_REQUIRE(P != nullptr, "plan is NULL");
    QLDPC_REQUIRE(count >= 0 && shot_begin >= 0, "negative shot range");
    QLDPC_USE_DEVICE(P->g->device);
    int rc = QLDPC_OK; (void)rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const qldpc_graph *g = P->g;
    const int n = g->n, m = g->m;
    bool joined_lanes[kFanLanes] = {false};
    if (P->fused && P->fan && count > 0) QLDPC_HIP_TRY(hipEventRecord(P->ev_in, s));      // the lanes start behind the caller's earlier work
    for (int64_t off = 0; off < count; off += P->batch) {
        const int64_t B = (count - off < P->batch) ? (count - off) : P->batch;
        if (P->fused) {
            // one launch: sample -> syndrome -> decode -> logical compare -> tally; BP failures are exported for OSD-0.
            // The failure records reuse the per-shot buffers of the unfused path (synd/err/dec/llr), compacted.
            // the lane of this batch: its failure-record buffers and, for small batches, its own stream
            const int nl = (int)P->lanes.size();
            const int set = nl > 1 ? (int)(P->batch_no % nl) : 0;
            P->batch_no++;
            qldpc_cc_plan::Lane *Ln = nl ? &P->lanes[set] : nullptr;
            DevBuf &b_count = set ? Ln->count : P->d_count, &b_list = set ? Ln->list : P->d_list, &b_synd = set ? Ln->synd : P->d_synd, &b_err = set ? Ln->err : P->d_err,
                   &b_dec = set ? Ln->dec : P->d_dec, &b_llr = set ? Ln->llr : P->d_llr, &b_cold = set ? Ln->cold : P->d_cold, &b_cont = set ? Ln->cont : P->d_cont;
            hipStream_t caller = s;
            if (P->fan) {                                        // everything of this batch on the lane's stream, behind the caller's earlier work
                if (!joined_lanes[set]) { QLDPC_HIP_TRY(hipStreamWaitEvent(Ln->st, P->ev_in, 0)); joined_lanes[set] = true; }
                s = Ln->st;
            }
            if (P->side && Ln->tail_pending) QLDPC_HIP_TRY(hipStreamWaitEvent(s, Ln->tail, 0));
            // counters of the piece: [0] BP failures (OSD-0 list), [2] shots handed on by the first iteration.  Zero at plan creation; the judge kernel, their
            // last reader, zeroes them again (without OSD-0 there is no judge: a memset then)
            if (!P->use_osd) QLDPC_HIP_TRY(hipMemsetAsync(b_count.p, 0, 16, s));
            // timing brackets of the decode launch: pieces small enough to share the chip with their neighbours have no meaningful span of their own and
            // a piece costs the host ~10 us per enqueue, so only large pieces and instrumented plans (QLDPC_FLAG_CLOCK_PROBE) carry them
            const bool timed = (P->flags & QLDPC_FLAG_CLOCK_PROBE) || P->batch > kFanBatch;
            hipEvent_t e0 = timed ? get_event(P) : nullptr, e1 = timed ? get_event(P) : nullptr;
            if (e0 && e1) QLDPC_HIP_TRY(hipEventRecord(e0, s));
            if (P->first_ok && mc_first_choice() == 1 && wave_kernel_choice() != 2) {
                // reference semantics: every shot through the bit-sliced first iteration, the few that do not stop there through the full decoder
                unsigned long long *clk = (P->flags & QLDPC_FLAG_CLOCK_PROBE) ? P->d_clk.as<unsigned long long>() + 2 * kClkSlots : nullptr;
                P->clk_first = true;
                if ((rc = mc_first_launch(g, P->k, P->d_lptr.as<int32_t>(), P->d_lidx.as<int32_t>(), B, seed, shot_begin + off, P->thr, P->negbits,
                                          P->d_tally.as<unsigned long long>(), b_cont.as<int32_t>(), b_count.as<int32_t>() + 2, clk, s)) != QLDPC_OK)
                    return rc;
                if (e0) { hipEvent_t em = get_event(P); if (em) { QLDPC_HIP_TRY(hipEventRecord(em, s)); P->pending_first.emplace_back(e0, em); } }
                rc = mc_regular_launch(g, B, P->d_prior.as<double>(), P->max_iter, P->d_alpha.as<double>(), P->clip, P->flags & ~QLDPC_FLAG_CLOCK_PROBE, P->nanfree, seed,
                                       shot_begin + off, P->thr, P->use_osd, P->d_Lmask.as<uint64_t>(), b_cold.p, s, b_cont.as<int32_t>(),
                                       b_count.as<int32_t>() + 2);
            }
#ifdef QLDPC_EXPERIMENTS
            else if ((P->clk_first = false), wave_kernel_choice() == 2 && wave_supported(g, P->damping, P->clean))
                rc = mc_wave_launch(g, B, P->d_prior.as<double>(), P->max_iter, P->d_alpha.as<double>(), P->clip, P->flags, seed, shot_begin + off,
                                    P->thr, P->use_osd, P->d_Lmask.as<uint64_t>(), b_cold.p, s);
#endif
            else if ((P->clk_first = false), P->generic)
                rc = mc_resident_launch(g, B, P->d_prior.as<double>(), P->max_iter, P->d_alpha.as<double>(), P->clip, P->flags, seed, shot_begin + off, P->thr,
                                        P->use_osd, P->d_Lmask.as<uint64_t>(), b_cold.p, s);
            else
                rc = mc_regular_launch(g, B, P->d_prior.as<double>(), P->max_iter, P->d_alpha.as<double>(), P->clip, P->flags, P->nanfree, seed,
                                       shot_begin + off, P->thr, P->use_osd, P->d_Lmask.as<uint64_t>(), b_cold.p, s);
            if (rc != QLDPC_OK) return rc;
            if (e0 && e1) { QLDPC_HIP_TRY(hipEventRecord(e1, s)); P->pending.emplace_back(e0, e1); }
            if (P->use_osd) {
                hipStream_t ts = s;
                if (P->side) {                                           // the tail follows the decode on the side stream
                    ts = P->side;
                    QLDPC_HIP_TRY(hipEventRecord(Ln->decoded, s));
                    QLDPC_HIP_TRY(hipStreamWaitEvent(ts, Ln->decoded, 0));
                }
                const qldpc_graph *go = (Ln && Ln->g) ? Ln->g : g;
                std::lock_guard<std::mutex> lk(go->mu);
                // a lane's private handle: nobody else draws tickets from its OSD-0 queue, so the judge kernel (behind the OSD-0 launch on the same
                // stream) zeroes it for the next piece and the launch skips its memset
                int *osd_q = nullptr;
                if (go->ws_private && (rc = osd_small_queue(go, &osd_q)) != QLDPC_OK) return rc;
                // (on such a lane the one-wave OSD-0 kernels of small matrices also take the judge over: two enqueues per piece instead of three)
                OsdJudge J{b_err.as<int8_t>(), P->d_Lmask.as<uint64_t>(), P->d_tally.as<unsigned long long>(), b_count.as<int32_t>(), B, false};
                if ((rc = osd0_listed_launch(go, b_list.as<int32_t>(), b_count.as<int32_t>(), B, b_synd.as<int8_t>(), b_llr.as<double>(),
                                             b_dec.as<int8_t>(), nullptr, b_dec.as<int8_t>(), (P->flags & QLDPC_FLAG_PUBLIC_MASK) | (osd_q ? QLDPC_FLAG_INTERNAL_OSD_QUEUE_CLEAN : 0), ts,
                                             osd_q ? &J : nullptr)) != QLDPC_OK)
                    return rc;
                if (!J.fused && (rc = judge_failed_launch(g, b_count.as<int32_t>(), true, osd_q, P->d_Lmask.as<uint64_t>(), b_err.as<int8_t>(), b_synd.as<int8_t>(),
                                              b_dec.as<int8_t>(), P->d_tally.as<unsigned long long>(), ts)) != QLDPC_OK)
                    return rc;
                if (P->side) { QLDPC_HIP_TRY(hipEventRecord(Ln->tail, ts)); Ln->tail_pending = true; }
            }
            s = caller;
            continue;
        }
        const int64_t nq = (n + 3) / 4;
        if (B * nq > 0)
            hipLaunchKernelGGL(cc_sample_kernel, dim3((unsigned)((B * nq + 255) / 256)), dim3(256), 0, s, B, shot_begin + off,
                               (uint32_t)seed, (uint32_t)(seed >> 32), n, P->thr, P->d_err.as<int8_t>());
        QLDPC_HIP_TRY(hipGetLastError());
        if ((rc = gf2_spmv_launch(g, B, P->d_err.as<int8_t>(), P->d_synd.as<int8_t>(), s)) != QLDPC_OK) return rc;
        hipEvent_t e0 = get_event(P), e1 = get_event(P);
        if (e0 && e1) QLDPC_HIP_TRY(hipEventRecord(e0, s));
        {
            std::lock_guard<std::mutex> lk(g->mu);
            rc = minsum_decode_dispatch(g, B, P->d_synd.as<int8_t>(), P->d_prior.as<double>(), P->max_iter, P->d_alpha.as<double>(),
                                        P->damping, P->clip, (P->flags & QLDPC_FLAG_PUBLIC_MASK) | (P->nanfree ? QLDPC_FLAG_INTERNAL_PRIOR_FINITE : 0) | (P->clean ? QLDPC_FLAG_INTERNAL_PRIOR_LE_CLIP : 0), P->nanfree, P->d_dec.as<int8_t>(), P->d_llr.as<double>(),
                                        P->d_conv.as<uint8_t>(), P->d_iter.as<int32_t>(), s);
        }
        if (rc != QLDPC_OK) return rc;
        if (e0 && e1) { QLDPC_HIP_TRY(hipEventRecord(e1, s)); P->pending.emplace_back(e0, e1); }
        if (P->use_osd) {
            QLDPC_HIP_TRY(hipMemsetAsync(P->d_count.p, 0, 4, s));
            hipLaunchKernelGGL(collect_failed_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, P->d_conv.as<uint8_t>(),
                               P->d_list.as<int32_t>(), P->d_count.as<int32_t>());
            QLDPC_HIP_TRY(hipGetLastError());
            {
                std::lock_guard<std::mutex> lk(g->mu);
                rc = osd0_listed_launch(g, P->d_list.as<int32_t>(), P->d_count.as<int32_t>(), B, P->d_synd.as<int8_t>(), P->d_llr.as<double>(),
                                        P->d_dec.as<int8_t>(), nullptr, P->d_dec.as<int8_t>(), P->flags, s);
            }
            if (rc != QLDPC_OK) return rc;
            hipLaunchKernelGGL(add_osd_count_kernel, dim3(1), dim3(64), 0, s, P->d_count.as<int32_t>(),
                               P->d_tally.as<unsigned long long>());
        }
        hipLaunchKernelGGL(cc_judge_kernel, dim3((unsigned)((B + 7) / 8)), dim3(256), 0, s, B, m, n, g->d_indptr, g->d_indices,
                           P->d_Lmask.as<uint64_t>(), P->d_err.as<int8_t>(), P->d_synd.as<int8_t>(), P->d_dec.as<int8_t>(),
                           P->d_conv.as<uint8_t>(), P->d_iter.as<int32_t>(), P->d_tally.as<unsigned long long>());
        QLDPC_HIP_TRY(hipGetLastError());
    }
    return QLDPC_OK;
}

static int drain_events(qldpc_cc_plan *P) {
    for (auto &pr : P->pending_first) {          // (first, mid): `first` is shared with the whole-decode bracket and returns to the pool there
        QLDPC_HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0;
        QLDPC_HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        P->ms_first += ms;
        P->pool.push_back(pr.second);
    }
    P->pending_first.clear();
    for (auto &pr : P->pending) {
        QLDPC_HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0;
        QLDPC_HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        P->ms_total += ms; P->launches++;
        P->pool.push_back(pr.first); P->pool.push_back(pr.second);
    }
    P->pending.clear();
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_cc_plan_read(qldpc_cc_plan *P, void *stream, int clear, int64_t *tally) {
    QLDPC_REQUIRE(P != nullptr && tally != nullptr, "NULL argument");
    QLDPC_USE_DEVICE(P->g->device);
    int rc = QLDPC_OK; (void)rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    QLDPC_HIP_TRY(hipStreamSynchronize(s));
    if (P->side) QLDPC_HIP_TRY(hipStreamSynchronize(P->side));
    for (auto &Ln : P->lanes) { if (Ln.st) QLDPC_HIP_TRY(hipStreamSynchronize(Ln.st)); Ln.tail_pending = false; }
    QLDPC_HIP_TRY(hipMemcpy(tally, P->d_tally.p, QLDPC_TALLY_SLOTS * 8, hipMemcpyDeviceToHost));
    if (clear) QLDPC_HIP_TRY(zero_now(P->d_tally.p, QLDPC_TALLY_SLOTS * 8));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_cc_plan_kernel_time(qldpc_cc_plan *P, double *ms_total, int64_t *launches) {
    QLDPC_REQUIRE(P != nullptr, "plan is NULL");
    int rc = drain_events(P);
    if (rc != QLDPC_OK) return rc;
    if (ms_total) *ms_total = P->ms_total;
    if (launches) *launches = P->launches;
    P->ms_total = 0; P->launches = 0; P->ms_first = 0;
    return QLDPC_OK;
}

// the part of qldpc_cc_plan_kernel_time's total spent in the bit-sliced first-iteration kernel (0 when the plan does not use it); call BEFORE
// qldpc_cc_plan_kernel_time, which resets both sums
QLDPC_EXPORT int qldpc_cc_plan_first_iteration_time(qldpc_cc_plan *P, double *ms_first) {
    QLDPC_REQUIRE(P != nullptr && ms_first != nullptr, "NULL argument");
    int rc = drain_events(P);
    if (rc != QLDPC_OK) return rc;
    *ms_first = P->ms_first;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_cc_plan_clock(qldpc_cc_plan *P, void *stream, double *mhz) {
    QLDPC_REQUIRE(P != nullptr && mhz != nullptr, "NULL argument");
    QLDPC_REQUIRE((P->flags & QLDPC_FLAG_CLOCK_PROBE) && P->fused, "the plan was created without QLDPC_FLAG_CLOCK_PROBE (or does not use the fused kernel)");
    QLDPC_USE_DEVICE(P->g->device);
    QLDPC_HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    for (auto &Ln : P->lanes) if (Ln.st) QLDPC_HIP_TRY(hipStreamSynchronize(Ln.st));
    std::vector<unsigned long long> h(2 * kClkSlots);
    QLDPC_HIP_TRY(hipMemcpy(h.data(), P->d_clk.as<unsigned long long>() + (P->clk_first ? 2 * kClkSlots : 0), h.size() * 8, hipMemcpyDeviceToHost));
    *mhz = clock_probe_median(h.data(), kClkSlots);
    return QLDPC_OK;
}

QLDPC_EXPORT void qldpc_cc_plan_destroy(qldpc_cc_plan *P) {
    if (!P) return;
    (void)hipSetDevice(P->g->device);
    for (auto &pr : P->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : P->pending_first) (void)hipEventDestroy(pr.second);
    for (auto e : P->pool) (void)hipEventDestroy(e);
    if (P->side) { (void)hipStreamSynchronize(P->side); (void)hipStreamDestroy(P->side); }
    if (P->ev_in) (void)hipEventDestroy(P->ev_in);
    for (auto &Ln : P->lanes) {
        if (Ln.st) { (void)hipStreamSynchronize(Ln.st); (void)hipStreamDestroy(Ln.st); }
        if (Ln.decoded) (void)hipEventDestroy(Ln.decoded);
        if (Ln.tail) (void)hipEventDestroy(Ln.tail);
        if (Ln.g) qldpc_graph_destroy(Ln.g);
        for (DevBuf *b : {&Ln.err, &Ln.synd, &Ln.dec, &Ln.llr, &Ln.list, &Ln.count, &Ln.cold, &Ln.cont}) b->release();
    }
    for (DevBuf *b : {&P->d_alpha, &P->d_prior, &P->d_Lmask, &P->d_err, &P->d_synd, &P->d_dec, &P->d_llr, &P->d_conv, &P->d_iter,
                      &P->d_tally, &P->d_list, &P->d_count, &P->d_sol, &P->d_cold, &P->d_clk, &P->d_lptr, &P->d_lidx, &P->d_cont})
        b->release();
    delete P;
}

QLDPC_EXPORT int qldpc_cc_sample_decode_tally(const qldpc_graph *g, int k, const uint8_t *L, double p, uint64_t seed,
                                              int64_t shot_begin, int64_t count, int max_iter, int alpha_mode, double alpha_val,
                                              const double *alpha_seq, int alpha_len, double damping, double clip_llr, int use_osd,
                                              int flags, int64_t *tally) {
    QLDPC_REQUIRE(tally != nullptr, "tally is NULL");
    QLDPC_REQUIRE(count >= 0, "negative count");
    qldpc_cc_plan *P = nullptr;
    int64_t batch = count < 1 ? 1 : (count < (1 << 18) ? count : (1 << 18));
    int rc = qldpc_cc_plan_create(g, k, L, p, max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, damping, clip_llr, use_osd, flags,
                                  batch, 0, &P);
    if (rc != QLDPC_OK) return rc;
    rc = qldpc_cc_plan_run(P, seed, shot_begin, count, nullptr);
    if (rc == QLDPC_OK) rc = qldpc_cc_plan_read(P, nullptr, 0, tally);
    qldpc_cc_plan_destroy(P);
    return rc;
}

QLDPC_EXPORT int qldpc_gf2_spmv_batch(const qldpc_graph *g, int64_t B, const int8_t *vectors, int8_t *out) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0, "negative batch");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || g->m == 0) return QLDPC_OK;
    QLDPC_REQUIRE(vectors != nullptr && out != nullptr, "NULL buffer");
    DevTmp dv, dout;
    if ((rc = dv.alloc((size_t)B * g->n)) || (rc = dout.alloc((size_t)B * g->m))) return rc;
    if (g->n) QLDPC_HIP_TRY(hipMemcpy(dv.p, vectors, (size_t)B * g->n, hipMemcpyHostToDevice));
    if ((rc = gf2_spmv_launch(g, B, dv.as<int8_t>(), dout.as<int8_t>(), nullptr)) != QLDPC_OK) return rc;
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(out, dout.p, (size_t)B * g->m, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

// device-pointer form: only enqueues on `stream`
QLDPC_EXPORT int qldpc_gf2_spmv_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_vectors, int8_t *d_out, void *stream) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0, "negative batch");
    QLDPC_USE_DEVICE(g->device);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || g->m == 0) return QLDPC_OK;
    QLDPC_REQUIRE(d_vectors != nullptr && d_out != nullptr, "NULL buffer");
    return gf2_spmv_launch(g, B, d_vectors, d_out, reinterpret_cast<hipStream_t>(stream));
}

QLDPC_EXPORT void qldpc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
