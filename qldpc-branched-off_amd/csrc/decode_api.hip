// C-ABI entry points of the batched min-sum decoder (a1/a2) and kernel selection.
#include "common.h"
#include "minsum_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace qldpc {

// alpha_k for k < max_iter, evaluated exactly as the reference does (src/decoding/kernels.py:272-275, 402-405).
int build_alpha_table(int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                      std::vector<double> &tab) {
    QLDPC_REQUIRE(alpha_mode == QLDPC_ALPHA_CONST || alpha_mode == QLDPC_ALPHA_DYNAMIC || alpha_mode == QLDPC_ALPHA_SEQ,
                  "unknown alpha_mode %d", alpha_mode);
    if (alpha_mode == QLDPC_ALPHA_SEQ) QLDPC_REQUIRE(alpha_seq != nullptr && alpha_len > 0, "alpha_seq must be a non-empty sequence for QLDPC_ALPHA_SEQ");
    tab.resize(max_iter > 0 ? max_iter : 1);
    for (int k = 0; k < max_iter; k++) {
        if (alpha_mode == QLDPC_ALPHA_DYNAMIC) tab[k] = 1.0 - std::ldexp(1.0, -(k + 1));   // 1.0 - 2.0**(-(k+1)), exact
        else if (alpha_mode == QLDPC_ALPHA_SEQ) tab[k] = (k < alpha_len) ? alpha_seq[k] : alpha_seq[alpha_len - 1];
        else tab[k] = alpha_val;
    }
    return QLDPC_OK;
}

bool resident_supported(const qldpc_graph *g, double damping);

bool inputs_clean(const double *prior, int n, double clip, const double *alpha, int n_alpha) {
    if (!(std::isfinite(clip) && clip > 0.0)) return false;
    for (int k = 0; k < n_alpha; k++) if (!(std::isfinite(alpha[k]) && alpha[k] > 0.0)) return false;
    for (int j = 0; j < n; j++) if (!std::isfinite(prior[j]) || (prior[j] == 0.0 && std::signbit(prior[j]))) return false;
    return true;
}

static int dispatch_kernel(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                           const double *d_alpha, double damping, double clip, int flags, bool nanfree, int8_t *d_err, double *d_llr,
                           uint8_t *d_conv, int32_t *d_iter, hipStream_t stream, const double *h_prior) {
    const bool want_stream = flags & QLDPC_FLAG_KERNEL_STREAM;
    const bool want_res = flags & (QLDPC_FLAG_KERNEL_RESIDENT | QLDPC_FLAG_KERNEL_GENERIC);
    const bool can_res = resident_supported(g, damping);
    if (!want_stream && !(flags & QLDPC_FLAG_KERNEL_GENERIC) && regular_supported(g, clip, max_iter)) {
#ifdef QLDPC_EXPERIMENTS
        const bool clean = nanfree && (flags & QLDPC_FLAG_INTERNAL_PRIOR_LE_CLIP);
        if (wave_kernel_choice() == 2 && wave_supported(g, damping, clean))
            return minsum_wave_launch(g, B, d_synd, d_prior, max_iter, d_alpha, clip, flags, d_err, d_llr, d_conv, d_iter, stream);
#endif
        return minsum_regular_launch(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip, flags, nanfree, d_err, d_llr, d_conv, d_iter, stream);
    }
    if (want_res && !can_res) {
        set_error("resident kernel does not support this graph (m=%d n=%d max row degree %d, max column degree %d)", g->m, g->n,
                  g->max_row_deg, g->max_col_deg);
        return QLDPC_ERR_UNSUPPORTED;
    }
    if (!want_stream && can_res)
        return minsum_resident_launch(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip, flags, d_err, d_llr, d_conv, d_iter, stream);
    if (!want_stream && wg_supported(g, damping)) {
        // the prior is known on the host and the inputs are clean: the form with every table in LDS (minsum_wg2.hip), unless a flag asks for a form of the
        // table kernel (QLDPC_FLAG_WG_TABLES and the layout / experiment selectors)
        if (h_prior && nanfree && damping == 1.0 && !(flags & (QLDPC_FLAG_WG_TABLES | QLDPC_FLAG_WG_VGLOBAL | QLDPC_FLAG_WG_GENERIC | QLDPC_FLAG_WG_ROWMAJOR |
                                                                  QLDPC_FLAG_WG_EDGE_LANES | QLDPC_FLAG_WG_IDXLOAD))) {
            const Wg2Prep *prep = nullptr;
            const int rcp = wg2_prepare(g, h_prior, &prep);
            if (rcp != QLDPC_OK) return rcp;
            if (prep) return minsum_wg2_launch(g, prep, B, d_synd, max_iter, d_alpha, clip, flags, d_err, d_llr, d_conv, d_iter, stream);
        }
        return minsum_wg_launch(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip, flags, nanfree, d_err, d_llr, d_conv, d_iter, stream);
    }
    return minsum_stream_launch(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip, flags, d_err, d_llr, d_conv, d_iter, stream);
}

// callers hold g->mu; the graph's device workspaces are handed over in stream order (common.h)
int minsum_decode_dispatch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                           const double *d_alpha, double damping, double clip, int flags, bool nanfree, int8_t *d_err, double *d_llr,
                           uint8_t *d_conv, int32_t *d_iter, hipStream_t stream, const double *h_prior) {
    int rc = g->ws_acquire(stream);
    if (rc != QLDPC_OK) return rc;
    rc = dispatch_kernel(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip, flags, nanfree, d_err, d_llr, d_conv, d_iter, stream, h_prior);
    const int rel = g->ws_release(stream);          // always: a failing call may have enqueued launches the next stream has to wait for
    return rc != QLDPC_OK ? rc : rel;
}

}  // namespace qldpc

using namespace qldpc;

static int check_decode_args(const qldpc_graph *g, int64_t B, const void *synd, const void *prior, int max_iter, double clip,
                             const void *e, const void *l, const void *c, const void *it) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    QLDPC_REQUIRE(B >= 0, "negative batch size");
    QLDPC_REQUIRE(max_iter >= 0, "negative max_iter");
    QLDPC_REQUIRE(!(clip != clip), "clip_llr is NaN");
    if (B > 0) {
        QLDPC_REQUIRE(prior != nullptr || g->n == 0, "prior is NULL");
        QLDPC_REQUIRE(synd != nullptr || g->m == 0, "syndromes is NULL");
        QLDPC_REQUIRE(e && l && c && it, "an output pointer is NULL");
    }
    return QLDPC_OK;
}

static int decode_dev_impl(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior,
                                               int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq,
                                               int alpha_len, double damping, double clip_llr, int flags, bool prior_finite, bool prior_le_clip, int8_t *d_err,
                                               double *d_llr, uint8_t *d_conv, int32_t *d_iter, void *stream, const double *h_prior = nullptr) {
    int rc = check_decode_args(g, B, d_synd, d_prior, max_iter, clip_llr, d_err, d_llr, d_conv, d_iter);
    if (rc != QLDPC_OK) return rc;
    QLDPC_USE_DEVICE(g->device);
    if (B == 0) return QLDPC_OK;
    std::vector<double> tab;
    if ((rc = build_alpha_table(max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, tab)) != QLDPC_OK) return rc;
    std::lock_guard<std::mutex> lk(g->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const double *d_alpha = nullptr;          // per-graph cache of alpha tables: a new schedule is uploaded once, nothing synchronises
    if ((rc = g->alpha_table(tab, s, &d_alpha)) != QLDPC_OK) return rc;
    // prior_finite here means "host-verified clean prior" (finite, no -0.0); clip and alphas are checked the same way
    bool nanfree = prior_finite && std::isfinite(damping) && inputs_clean(nullptr, 0, clip_llr, tab.data(), max_iter);
    flags = (flags & QLDPC_FLAG_PUBLIC_MASK) | (prior_finite ? QLDPC_FLAG_INTERNAL_PRIOR_FINITE : 0) |
            ((prior_finite && prior_le_clip) ? QLDPC_FLAG_INTERNAL_PRIOR_LE_CLIP : 0);
    return minsum_decode_dispatch(g, B, d_synd, d_prior, max_iter, d_alpha, damping, clip_llr, flags, nanfree, d_err,
                                  d_llr, d_conv, d_iter, s, h_prior);
}

QLDPC_EXPORT int qldpc_minsum_decode_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior,
                                               int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq,
                                               int alpha_len, double damping, double clip_llr, int flags, int8_t *d_err,
                                               double *d_llr, uint8_t *d_conv, int32_t *d_iter, void *stream) {
    // the prior lives on the device: its finiteness is unknown here, so the NaN test of kernels.py:328 stays in
    return decode_dev_impl(g, B, d_synd, d_prior, max_iter, alpha_mode, alpha_val, alpha_seq, alpha_len, damping, clip_llr, flags, false, false,
                           d_err, d_llr, d_conv, d_iter, stream);
}

QLDPC_EXPORT int qldpc_minsum_decode_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *prior,
                                           int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq,
                                           int alpha_len, double damping, double clip_llr, int flags, int8_t *out_err,
                                           double *out_llr, uint8_t *out_conv, int32_t *out_iter) {
    // out_llr may be NULL: the posteriors (8n of the 9n + 5 result bytes per shot) then stay on the device
    int rc = check_decode_args(g, B, syndromes, prior, max_iter, clip_llr, out_err, out_llr ? (const void *)out_llr : (const void *)out_err, out_conv, out_iter);
    if (rc != QLDPC_OK) return rc;
    QLDPC_USE_DEVICE(g->device);
    if (B == 0) return QLDPC_OK;
    const size_t m = g->m, n = g->n;
    // One grow-only device slab per graph handle (no hipMalloc / hipFree per call: the reference-style single-shot call is
    // latency bound) laid out as  prior | syndromes | llr | iter | err | conv ; small results return in ONE copy through a
    // pinned staging buffer, large ones directly into the caller's arrays.
    const size_t o_prior = 0, o_synd = o_prior + round_up((int64_t)n * 8, 16), o_llr = o_synd + round_up((int64_t)B * m, 16),
                 o_iter = o_llr + round_up((int64_t)B * n * 8, 16), o_err = o_iter + round_up((int64_t)B * 4, 16),
                 o_conv = o_err + round_up((int64_t)B * n, 16), total = o_conv + round_up(B, 16);
    std::unique_lock<std::mutex> io(g->mu_io);
    if ((rc = g->ws_io.ensure(total)) != QLDPC_OK) return rc;
    unsigned char *base = g->ws_io.as<unsigned char>();
    if (m) QLDPC_HIP_TRY(hipMemcpyAsync(base + o_synd, syndromes, B * m, hipMemcpyHostToDevice, nullptr));
    if (n) QLDPC_HIP_TRY(hipMemcpyAsync(base + o_prior, prior, n * 8, hipMemcpyHostToDevice, nullptr));
    const double one = 1.0;
    const bool prior_finite = inputs_clean(prior, (int)n, 1.0, &one, 1);
    bool prior_le_clip = prior_finite;
    for (size_t j = 0; j < n && prior_le_clip; j++) prior_le_clip = std::fabs(prior[j]) <= clip_llr;
    rc = decode_dev_impl(g, B, reinterpret_cast<int8_t *>(base + o_synd), reinterpret_cast<double *>(base + o_prior), max_iter, alpha_mode,
                         alpha_val, alpha_seq, alpha_len, damping, clip_llr, flags, prior_finite, prior_le_clip, reinterpret_cast<int8_t *>(base + o_err),
                         reinterpret_cast<double *>(base + o_llr), reinterpret_cast<uint8_t *>(base + o_conv),
                         reinterpret_cast<int32_t *>(base + o_iter), nullptr, prior);
    if (rc != QLDPC_OK) return rc;
    const size_t out_bytes = total - o_llr;
    if (out_bytes <= ((size_t)1 << 20) && out_llr) {
        if (g->pin_cap < out_bytes) {
            if (g->pin) (void)hipHostFree(g->pin);
            g->pin = nullptr; g->pin_cap = 0;
            QLDPC_HIP_TRY(hipHostMalloc(&g->pin, (size_t)1 << 20, hipHostMallocDefault));
            g->pin_cap = (size_t)1 << 20;
        }
        QLDPC_HIP_TRY(hipMemcpyAsync(g->pin, base + o_llr, out_bytes, hipMemcpyDeviceToHost, nullptr));
        QLDPC_HIP_TRY(hipStreamSynchronize(nullptr));
        const unsigned char *h = static_cast<const unsigned char *>(g->pin);
        if (n) std::memcpy(out_llr, h, B * n * 8);
        std::memcpy(out_iter, h + (o_iter - o_llr), B * 4);
        if (n) std::memcpy(out_err, h + (o_err - o_llr), B * n);
        std::memcpy(out_conv, h + (o_conv - o_llr), B);
    } else {
        if (n && out_llr) QLDPC_HIP_TRY(hipMemcpy(out_llr, base + o_llr, B * n * 8, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(out_iter, base + o_iter, B * 4, hipMemcpyDeviceToHost));
        if (n) QLDPC_HIP_TRY(hipMemcpy(out_err, base + o_err, B * n, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(out_conv, base + o_conv, B, hipMemcpyDeviceToHost));
    }
    return QLDPC_OK;
}
