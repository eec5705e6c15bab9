// Device helpers shared by the min-sum kernels.  All arithmetic is IEEE f64 in the reference's operand order.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

struct qldpc_graph;

namespace qldpc {

// reference src/decoding/kernels.py:328-333: NaN -> 0, else clip to [-clip, clip]
__device__ __forceinline__ double clip_nan(double q, double clip) {
    if (q != q) return 0.0;
    if (q > clip) return clip;
    if (q < -clip) return -clip;
    return q;
}

// The reference evaluates  q_damped = damping * q_new + (1.0 - damping) * Q_old  even for damping == 1 (kernels.py:336).  Q_old of
// iteration 0 is the UNCLIPPED prior, so for a column whose prior is +-inf or NaN the product 0.0 * Q_old is NaN: every message into
// such a column's checks is NaN from iteration 1 on (NaN survives the second clip and 0.0 * NaN keeps it alive).  The kernels skip the
// damping arithmetic when damping == 1 (exact for finite Q_old) and reproduce this case explicitly: q = NaN where the prior is not finite.
__device__ __forceinline__ bool prior_not_finite(double p) { return !(fabs(p) < INFINITY); }

// internal flag (upper half of `flags`): the caller verified on the host that every prior is finite
#define QLDPC_FLAG_PUBLIC_MASK 0x0FFFFFFF          // flag bits callers may set (include/qldpc_hip.h)
#define QLDPC_FLAG_INTERNAL_PRIOR_FINITE 0x40000000
#define QLDPC_FLAG_INTERNAL_PRIOR_LE_CLIP 0x20000000   // ... and every |prior| <= clip (iteration 0 then needs no unclipped special case)
#define QLDPC_FLAG_INTERNAL_OSD_QUEUE_CLEAN 0x10000000 // OSD-0 launches: the handle's small-kernel ticket counter is zero and the caller zeroes it again afterwards

// reference src/decoding/kernels.py:339-342
__device__ __forceinline__ double clip_only(double q, double clip) {
    if (q > clip) return clip;
    if (q < -clip) return -clip;
    return q;
}

// Host-side launchers implemented by the kernel files.
int minsum_stream_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                         const double *d_alpha, double damping, double clip, int flags, int8_t *d_err, double *d_llr,
                         uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);

int minsum_resident_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                           const double *d_alpha, double damping, double clip, int flags, int8_t *d_err, double *d_llr,
                           uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);
bool resident_supported(const qldpc_graph *g, double damping);
int mc_resident_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                       uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask, void *d_cold, hipStream_t stream);
// regular-degree fast path (minsum_regular.hip).  nanfree: the caller proved prior / clip / alphas finite.
bool regular_supported(const qldpc_graph *g, double clip, int max_iter);
int minsum_regular_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                          const double *d_alpha, double damping, double clip, int flags, bool nanfree, int8_t *d_err, double *d_llr,
                          uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);
int mc_regular_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                      bool nanfree, uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask,
                      void *d_cold, hipStream_t stream, const int32_t *d_shot_list = nullptr, const int32_t *d_shot_count = nullptr);
// bit-sliced first iteration of a uniform-prior Monte-Carlo plan under reference semantics (mc_first.hip)
bool mc_first_table(const qldpc_graph *g, double p0, double alpha0, double clip, int max_iter, unsigned &negbits);
int mc_first_launch(const qldpc_graph *g, int k, const int32_t *d_lptr, const int32_t *d_lidx, int64_t B, uint64_t seed, int64_t shot_begin, uint32_t thr,
                    unsigned negbits, unsigned long long *d_tally, int32_t *d_cont_list, int32_t *d_cont_count, unsigned long long *d_clk, hipStream_t stream);
void mc_first_set_bits(int bits);
void regular_set_list_shots(int s);
void mc_set_big_lanes(int n);
int mc_tail_overlap_choice();  // qldpc_set_option("mc_tail_overlap"): 1 = OSD-0 + judge of a batch on a side stream beside the next batch's decode (default)
int mc_first_choice();       // qldpc_set_option("mc_first_iteration"): 1 = use it where it applies (default), 0 = full decoder for every shot
int mc_regular_fill_cold(void *d_cold, unsigned long long *d_tally, int32_t *d_fail_count, int32_t *d_fail_list, int8_t *f_synd,
                         int8_t *f_err, int8_t *f_hard, double *f_llr, unsigned long long *d_clk);
size_t mc_regular_cold_bytes();
int judge_failed_launch(const qldpc_graph *g, int32_t *d_count, bool reset_counters, int *d_osd_queue, const uint64_t *d_Lmask, const int8_t *f_err, const int8_t *f_synd,
                        const int8_t *f_dec, unsigned long long *d_tally, hipStream_t stream);
int osd_small_queue(const qldpc_graph *g, int **queue);      // osd_small.hip
// wave-private kernel for (6,3)-regular graphs and clean inputs (minsum_wave.hip); option "regular_kernel" selects between the two
bool wave_supported(const qldpc_graph *g, double damping, bool clean);
int wave_kernel_choice();     // 0 automatic, 1 team kernel, 2 wave kernel (qldpc_set_option)
int minsum_wave_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter, const double *d_alpha,
                       double clip, int flags, int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);
int mc_wave_launch(const qldpc_graph *g, int64_t B, const double *d_prior, int max_iter, const double *d_alpha, double clip, int flags,
                   uint64_t seed, int64_t shot_begin, uint32_t thr, int use_osd, const uint64_t *d_Lmask, void *d_cold, hipStream_t stream);
// workgroup-per-shot kernel for large graphs (minsum_wg.hip)
bool wg_supported(const qldpc_graph *g, double damping);
int minsum_wg_launch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter, const double *d_alpha,
                     double damping, double clip, int flags, bool clean, int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);
// LDS-resident form of that kernel for callers whose prior is known on the host (minsum_wg2.hip); *out = NULL when the input is not eligible
struct Wg2Prep;
int wg2_prepare(const qldpc_graph *g, const double *h_prior, const Wg2Prep **out);
int minsum_wg2_launch(const qldpc_graph *g, const Wg2Prep *P, int64_t B, const int8_t *d_synd, int max_iter, const double *d_alpha, double clip, int flags,
                      int8_t *d_err, double *d_llr, uint8_t *d_conv, int32_t *d_iter, hipStream_t stream);
void wg2_cache_free(void *cache);
// "clean" decoder inputs, verified on the host: every prior finite and not -0.0, clip finite > 0, every alpha finite > 0.
// Then no message or posterior can be -0.0 and no |q| NaN, which the regular and lean kernels exploit (see their headers).
bool inputs_clean(const double *prior, int n, double clip, const double *alpha, int n_alpha);
// h_prior: the same prior on the host when the caller has it (a circuit plan, the host-pointer entry point), else NULL
int minsum_decode_dispatch(const qldpc_graph *g, int64_t B, const int8_t *d_synd, const double *d_prior, int max_iter,
                           const double *d_alpha, double damping, double clip, int flags, bool nanfree, int8_t *d_err, double *d_llr,
                           uint8_t *d_conv, int32_t *d_iter, hipStream_t stream, const double *h_prior = nullptr);

}  // namespace qldpc
