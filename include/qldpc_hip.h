/*
 * qldpc_hip.h -- C ABI of the MI355X (gfx950) qLDPC decoding / Monte-Carlo library.
 *
 * This is the drop-in boundary for the hot path of michelebanfi/qLDPC-branched-off: the reference's
 * numba @njit kernels (src/decoding/kernels.py, src/noise/kernels.py) are what these entry points
 * replace; the reference's Python wrappers (src/decoding/{sparse,dense,osd}.py, src/noise/simulation.py,
 * src/simulation/engine.py) keep their names and call through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every function returns int: 0 = QLDPC_OK, negative = error; text via qldpc_last_error()
 *     (thread-local).  Kernels never "raise": non-convergence is an output value, as in the reference.
 *   - the caller owns every buffer; the library owns only opaque handles and its device workspaces.
 *   - *_dev variants take DEVICE pointers (hipMalloc'ed or torch CUDA tensors' data_ptr()) and a
 *     hipStream_t passed as void* (NULL = default stream); they enqueue work and do not synchronise
 *     (alpha tables are cached per graph handle: a schedule is uploaded the first time it is seen, asynchronously).
 *     The variants without _dev take HOST pointers, copy in/out and return when results are ready.
 *   - every entry point selects its handle's device for the duration of the call and restores the caller's
 *     current device before returning.
 *   - a graph handle owns device workspaces shared by every decode / OSD call on it; calls on different streams
 *     are ordered through an event (the later one waits for the earlier one's kernels), so concurrent streams are
 *     safe but serialise per graph handle.  Use one handle per stream for real concurrency.
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails with
 *     QLDPC_ERR_NO_DEVICE.
 *   - matrices over GF(2) are CSR with sorted column indices (int32 indptr[m+1], indices[nnz]).
 *   - batched arrays are shot-major: syndromes[B][m], errors[B][n], llr[B][n].
 */
#ifndef QLDPC_HIP_H
#define QLDPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLDPC_OK 0
#define QLDPC_ERR_INVALID (-1)    /* bad argument (NULL, negative size, unsorted CSR, ...) */
#define QLDPC_ERR_NO_DEVICE (-2)  /* no usable gfx950 device / HIP runtime error at init     */
#define QLDPC_ERR_HIP (-3)        /* a HIP runtime call failed                              */
#define QLDPC_ERR_UNSUPPORTED (-4)

/* alpha schedule of the normalised min-sum (selection rules live in the Python wrappers,
 * src/decoding/sparse.py:18-29):  CONST: alpha_k = alpha_val (kernels.py:275);
 * DYNAMIC: alpha_k = 1 - 2^-(k+1) (kernels.py:273);  SEQ: alpha_k = seq[min(k,len-1)] (kernels.py:402-405). */
#define QLDPC_ALPHA_CONST 0
#define QLDPC_ALPHA_DYNAMIC 1
#define QLDPC_ALPHA_SEQ 2

/* decode flags */
#define QLDPC_FLAG_FIXED_ITERS 0x1   /* execute all max_iter iterations for every shot; outputs are still
                                        frozen at each shot's first converged iteration (identical results) */
#define QLDPC_FLAG_KERNEL_STREAM 0x10   /* force the HBM-streaming kernel (any graph size)            */
#define QLDPC_FLAG_KERNEL_RESIDENT 0x20 /* force the LDS/register-resident kernels (small graphs only) */
#define QLDPC_FLAG_KERNEL_GENERIC 0x40  /* resident family: use the generic (irregular-degree) kernel even for regular graphs */
#define QLDPC_FLAG_MC_UNFUSED 0x80      /* Monte-Carlo plans: separate sample / decode / judge launches instead of the fused kernel */
/* size-class selectors of the product library (parity tests force every class on small inputs; results are identical whichever runs) */
#define QLDPC_FLAG_OSD_LDS 0x20000      /* OSD-0: the row-transform kernel even for small matrices (m <= 128, n <= 1024), which otherwise take the
                                           literal one-wave-per-shot elimination */
#define QLDPC_FLAG_WG_VGLOBAL 0x100     /* workgroup-per-shot decoder: posteriors in HBM/L2 even when they fit LDS (the large-graph form) */
#define QLDPC_FLAG_WG_GENERIC 0x200     /* workgroup-per-shot decoder: the any-input kernel even for host-verified clean inputs */
#define QLDPC_FLAG_OSD_REFORDER 0x80000 /* OSD-0, m <= 1024: every shot through the kernel that follows the reference's row choice at each pivot (by default
                                          only the shots whose right-hand side lies outside the column space take it; the others cannot tell) */
#define QLDPC_FLAG_OSD_UG 0x400         /* OSD-0: row transform in HBM/L2 even when it fits LDS (the m > 1024 form) */
#define QLDPC_FLAG_OSD_GLOBAL 0x800     /* OSD-0: the literal global-memory elimination (general fallback) */
#define QLDPC_FLAG_WG_TABLES 0x200000   /* workgroup-per-shot decoder: the form with its index tables in HBM/L2 (csrc/minsum_wg.hip; what the *_dev entry points always run)
                                          even when the prior is known on the host and the LDS-resident form (csrc/minsum_wg2.hip) applies */
#define QLDPC_FLAG_WG_ROWMAJOR 0x8000   /* workgroup-per-shot decoder: natural row / column order instead of the degree-sorted assignment */
#define QLDPC_FLAG_CLOCK_PROBE 0x4000   /* plans: workgroups stamp s_memtime / s_memrealtime around their work (see *_plan_clock) */
/* measured-and-rejected kernels: libqldpc_hip_experiments.so only (make -C csrc experiments; same ABI, loaded by the parity tests).  The product
 * library answers these with QLDPC_ERR_UNSUPPORTED.  Numbers: profiles/r02_osd_experiments.txt, r02_bp_lane_mapping.txt, r03_wave_kernel.txt */
#define QLDPC_FLAG_WG_EDGE_LANES 0x2     /* workgroup-per-shot decoder: check pass with 16 lanes per check and shuffle reductions (SURVEY 7-6 option B) */
#define QLDPC_FLAG_WG_IDXLOAD 0x40000    /* workgroup-per-shot decoder: reload the row's column indices every iteration (m <= 1024 keeps them in registers) */
#define QLDPC_FLAG_OSD_QUEUE 0x100000     /* OSD-0, 897 <= m <= 1024: the free-pivot kernel with a look-ahead queue of reduced columns (csrc/osd_gjq.hip) */

/* tally slots written by the *_sample_decode_tally entry points (int64[QLDPC_TALLY_SLOTS]);
 * replaces the Python tally loop of src/simulation/engine.py:450-457 */
#define QLDPC_TALLY_SLOTS 16
#define QLDPC_TALLY_TRIALS 0
#define QLDPC_TALLY_Z_ERR 1       /* code capacity: logical errors of the single decoded sector */
#define QLDPC_TALLY_X_ERR 2
#define QLDPC_TALLY_TOTAL_ERR 3
#define QLDPC_TALLY_BP_CONV_Z 4
#define QLDPC_TALLY_BP_CONV_X 5
#define QLDPC_TALLY_OSD_Z 6
#define QLDPC_TALLY_OSD_X 7
#define QLDPC_TALLY_ITERS_Z 8     /* sum over shots of (final_iter + 1) = iterations executed under reference semantics */
#define QLDPC_TALLY_ITERS_X 9
#define QLDPC_TALLY_ZERO_SYND_Z 10
#define QLDPC_TALLY_ZERO_SYND_X 11
#define QLDPC_TALLY_UNSAT_Z 12    /* decoder output (after OSD if enabled) does not reproduce the syndrome */
#define QLDPC_TALLY_UNSAT_X 13

typedef struct qldpc_graph qldpc_graph; /* Tanner graph: host CSR + CSC (ascending check order) + device copies */

const char *qldpc_last_error(void);
int qldpc_version(void);
/* number of usable HIP devices (0 when none; never fails) */
int qldpc_device_count(void);
/* Process-wide kernel-selection switches for tools/ and the parity tests.  Results never depend on them (every selectable path is checked
 * against the same fixtures); the defaults are what bench.py measures.  New here (the reference has no counterpart).
 *   "mc_first_iteration"  reference-semantics Monte-Carlo plans: 1 (default) = bit-sliced first iteration (csrc/mc_first.hip) + the full decoder
 *                         on the shots it lists, 0 = the full decoder for every shot
 *   "mc_first_bits"       shots per lane of that kernel: 8 (default), 16, 32
 *   "osd_presort"         read at an OSD-0 launch: how many columns of the |llr| order the free-pivot OSD-0 kernels sort up front (default -1 = automatic, about m; the rest is
 *                         ordered only when a sweep gets that far; 0 = everything up front)
 *   "mc_tail_overlap"     read at plan creation: 1 (default) = whole batches on the plan's own streams so that the latency-bound pieces of one batch run
 *                         beside the next batch's first kernel (3 streams for large batches under reference semantics, 8 for batches <= 32768; fixed-work
 *                         plans with large batches keep the caller's stream), 2 = only OSD-0 + judge on a side stream, 0 = everything on the caller's stream
 *   "mc_big_lanes"        streams whole batches rotate over under reference semantics with large batches: 2 .. 8 (default 3)
 *   "mc_list_shots"       shots per workgroup of the full decoder on listed shots: 0 (default: the kernel's own 7), 1 .. 16
 *   "regular_kernel", "wave_cpl", "wave_rst", "wave_grid"  experiments build only: the wave-private decoder (csrc/minsum_wave.hip); the
 *                         product library accepts 0 and answers anything else with QLDPC_ERR_UNSUPPORTED */
int qldpc_set_option(const char *name, int value);
/* A stream of `device` for the `stream` arguments below (a host without PyTorch that drives several plans -- one per device, or several on one
 * device -- gives each its own; NULL there means the device's default stream).  New here: the reference's counterpart is the worker process
 * of its pool (src/simulation/engine.py:433-435).  qldpc_stream_sync blocks until everything enqueued on it has finished. */
int qldpc_stream_create(int device, void **stream);
int qldpc_stream_sync(int device, void *stream);
int qldpc_stream_destroy(int device, void *stream);

/* Build a Tanner-graph handle on `device`.  Validates the CSR (monotone indptr, 0 <= col < n, strictly
 * increasing columns per row) and derives the CSC view with per-column ASCENDING check order, which is what
 * reproduces the reference's scatter-add order R_sum[col] += msg (kernels.py:316).  Immutable afterwards. */
int qldpc_graph_create(int m, int n, const int32_t *indptr, const int32_t *indices, int device, qldpc_graph **out);
void qldpc_graph_destroy(qldpc_graph *g);
int qldpc_graph_dims(const qldpc_graph *g, int *m, int *n, int *nnz);

/* a1 + a2: minsum_decoder_full (src/decoding/kernels.py:234-366) and minsum_decoder_full_autoregressive
 * (kernels.py:369-485), batched over B independent syndromes.  B = 1 backs performMinSum_Symmetric_Sparse
 * (src/decoding/sparse.py:5-54).  Outputs per shot: candidateError int8[n], converged, values f64[n],
 * final_iter (max_iter-1 when not converged).  alpha_seq may be NULL unless alpha_mode == SEQ. * The host-pointer form accepts out_llr == NULL: the posteriors (8n of the 9n + 5 result bytes per shot) are then not copied back.
 */
int qldpc_minsum_decode_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *prior,
                              int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                              double damping, double clip_llr, int flags,
                              int8_t *out_err, double *out_llr, uint8_t *out_conv, int32_t *out_iter);
int qldpc_minsum_decode_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_syndromes, const double *d_prior,
                                  int max_iter, int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len,
                                  double damping, double clip_llr, int flags,
                                  int8_t *d_out_err, double *d_out_llr, uint8_t *d_out_conv, int32_t *d_out_iter,
                                  void *stream);

/* a3: minsum_core_sparse (kernels.py:138-169): one check-node pass; Q[B][nnz], syndrome_sign[B][m] (+-1.0)
 * -> R[B][nnz], R_sum[B][n].  B = 1 is the reference call. */
int qldpc_minsum_check_pass(const qldpc_graph *g, int64_t B, const double *Q, const double *syndrome_sign, double alpha,
                            double *R, double *R_sum);
/* a5: bp_core (kernels.py:171-193) on the same CSR edge layout: tanh-product rule, clip_val = 0.9999999 */
int qldpc_bp_check_pass(const qldpc_graph *g, int64_t B, const double *Q, const double *syndrome_sign, double clip_val,
                        double *R, double *R_sum);
/* a5 driver: performBeliefPropagationFast (src/decoding/dense.py:75-96), batched */
int qldpc_bp_decode_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *prior, int max_iter,
                          int8_t *out_err, double *out_llr, uint8_t *out_conv, int32_t *out_iter);

/* a6: GF(2) syndrome SpMV  s = H e (kernels.py:222-231, 352-359; H_csr.dot(e)%2 in alpha.py:128): vectors[B][n] -> out[B][m] */
int qldpc_gf2_spmv_batch(const qldpc_graph *g, int64_t B, const int8_t *vectors, int8_t *out);
/* same on device pointers; only enqueues on `stream` (hipStream_t) */
int qldpc_gf2_spmv_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_vectors, int8_t *d_out, void *stream);

/* a7: gf2_elimination (kernels.py:5-34): Gauss-Jordan on B byte matrices A[B][m][n] (0/1), b[B][m], in place.
 * pivot_rows/pivot_cols: int64[B][min(m,n)], num_pivots int32[B]. */
int qldpc_gf2_eliminate(int64_t B, int m, int n, uint8_t *A, uint8_t *b, int64_t *pivot_rows, int64_t *pivot_cols,
                        int32_t *num_pivots);
/* a8: gf2_elimination_packed_core (kernels.py:48-96) on rows packed little-endian into uint64 words
 * (layout of _pack_rows_uint64, kernels.py:36-46): A[B][m][nwords], in place. */
int qldpc_gf2_eliminate_packed(int64_t B, int m, int n, int nwords, uint64_t *A, uint8_t *b, int64_t *pivot_rows,
                               int64_t *pivot_cols, int32_t *num_pivots);
/* a9: performOSD_enhanced with order = 0 (src/decoding/osd.py:5-29), batched over B shots of one graph.
 * ordering (int32[B][n], may be NULL): column order to eliminate in; NULL = ascending |llr| with ties broken by
 * ascending index (np.argsort's default kind leaves ties implementation-defined, osd.py:12). solution int8[B][n]. */
int qldpc_osd0_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *llr, const int8_t *hard,
                     const int32_t *ordering, int flags, int8_t *solution);
/* same on device pointers; only enqueues on `stream`.  d_select / d_select_count (both NULL = all B shots): device list of the shots to
 * solve and its device-resident length, e.g. the shots qldpc_minsum_decode_batch_dev left unconverged -- the decode -> OSD-0 hand-over of
 * src/simulation/engine.py:96-97 without a host round trip.  Shots not listed keep whatever d_solution holds. */
int qldpc_osd0_batch_dev(const qldpc_graph *g, int64_t B, const int8_t *d_syndromes, const double *d_llr, const int8_t *d_hard,
                         const int32_t *d_ordering, const int32_t *d_select, const int32_t *d_select_count, int flags, int8_t *d_solution,
                         void *stream);
/* Phase counters of the OSD-0 kernels and of the workgroup BP kernel on the current device (uint64[32]; layout in csrc/osd_common.h).  Only the diagnostic build
 * (make -C csrc timers) counts; the product build carries no clock reads and returns QLDPC_ERR_UNSUPPORTED. */
int qldpc_osd_timers_read(uint64_t *out32, int reset);
/* f1: performOSD_enhanced(H, syndrome, llr, hard, order, max_combinations) (src/decoding/osd.py:5-77), batched.  The OSD-0 solution is
 * returned whenever it reproduces the syndrome (osd.py:27-29); otherwise the <= C(order+10, <= order) flip sets over the least
 * reliable non-pivot positions are scored with recompute_solution / compute_metric (src/decoding/kernels.py:195-219) and the
 * reference's selection rule.  max_combinations <= 0: no limit (Python None).  order <= 10; at most 65536 flip sets per shot
 * (QLDPC_ERR_UNSUPPORTED beyond, use max_combinations).  Ties of |llr|: ascending index (see qldpc_osd0_batch). */
int qldpc_osdw_batch(const qldpc_graph *g, int64_t B, const int8_t *syndromes, const double *llr, const int8_t *hard,
                     const int32_t *ordering, int order, int64_t max_combinations, int8_t *solution);

/* a10: generate_noisy_circuit_jit (src/noise/kernels.py:175-353), batched over B draws of explicit random
 * arrays rv/rp/rt [B][n_locs]; out_* [B][cap]; out_len int64[B]. */
int qldpc_noisy_circuit_batch(int64_t B, int64_t len, const int32_t *ops, const int32_t *q1, const int32_t *q2, double p,
                              int64_t n_locs, const double *rv, const int32_t *rp, const int32_t *rt, int64_t cap,
                              int32_t *out_ops, int32_t *out_q1, int32_t *out_q2, int64_t *out_len);
/* a11: simulate_circuit_Z_jit / simulate_circuit_X_jit (noise/kernels.py:13-91 / 94-172), batched over B op
 * lists ops[B][cap] with lengths len[B]; hist int8[B][max_syn], state int8[B][total_qubits], counts int64[B][2]. */
int qldpc_frame_sim_batch(int sector_is_x, int64_t B, int64_t cap, const int64_t *len, const int32_t *ops,
                          const int32_t *q1, const int32_t *q2, int total_qubits, int max_syn, int8_t *hist, int8_t *state,
                          int64_t *counts);
/* a12: sparsify_syndrome_jit (noise/kernels.py:356-380), batched: hist[B][stride] -> out[B][stride] */
int qldpc_sparsify_batch(int64_t B, int64_t stride, const int8_t *hist, const int64_t *syn_count, const int32_t *positions,
                         const int32_t *ptrs, int num_checks, int8_t *out);

/* Code-capacity Monte-Carlo (BASELINE configs 1-4), fused on the device: for global shots
 * [shot_begin, shot_begin+count): sample e ~ Bernoulli(p)^n (law of src/decoding/alpha.py:127-128, Philox4x32-10
 * stream keyed (seed, shot)), s = H e, decode (a1), OSD-0 on non-converged shots if use_osd, logical failure iff
 * L (e xor e_hat) != 0 (rule of src/simulation/engine.py:99-100), tally.  L: dense k x n bytes (host).
 * Host-synchronous: tally (host int64[16]) is complete on return.  Results are independent of how the shot range
 * is split across calls / devices. */
int qldpc_cc_sample_decode_tally(const qldpc_graph *g, int k, const uint8_t *L, double p, uint64_t seed,
                                 int64_t shot_begin, int64_t count, int max_iter, int alpha_mode, double alpha_val,
                                 const double *alpha_seq, int alpha_len, double damping, double clip_llr, int use_osd,
                                 int flags, int64_t *tally);

/* MC plan handle: same pipeline, asynchronous, for benchmarking and multi-stream use.  qldpc_cc_plan_run cuts its shot range into pieces of
 * `batch` shots, each enqueued as one pass of the pipeline accumulating into the plan's device tally; the plan owns device buffers for `batch`
 * shots per piece in flight (up to 8 pieces for batch <= 32768, up to 3 under reference semantics above that, else 1); qldpc_cc_plan_read
 * synchronises and returns (and optionally clears) the tally.  `batch` is taken literally.  A piece costs the host three enqueues (decode,
 * OSD-0, judge; ~10 us each), so small pieces bound the rate from the host side: `min_launch` > batch (an argument of THIS plan, 0 = none) lets
 * the plan cut at that granule instead, with buffers sized for it -- the tallies do not depend on the cut (the random stream is keyed by the
 * global shot index), only the memory bound `batch` expressed is given up. */
typedef struct qldpc_cc_plan qldpc_cc_plan;
int qldpc_cc_plan_create(const qldpc_graph *g, int k, const uint8_t *L, double p, int max_iter, int alpha_mode,
                         double alpha_val, const double *alpha_seq, int alpha_len, double damping, double clip_llr,
                         int use_osd, int flags, int64_t batch, int64_t min_launch, qldpc_cc_plan **out);
/* _run only enqueues: on `stream`, or (option mc_tail_overlap >= 1: reference-semantics plans and plans with batch <= 32768) on streams the plan owns,
 * which start behind everything `stream` held when _run was called; several batches are then in flight at once.  _read waits for `stream` and for
 * the plan's own streams, then copies the tally: it is the only way results leave the plan. */
int qldpc_cc_plan_run(qldpc_cc_plan *plan, uint64_t seed, int64_t shot_begin, int64_t count, void *stream);
int qldpc_cc_plan_read(qldpc_cc_plan *plan, void *stream, int clear, int64_t *tally);
/* time of the decode kernel launches enqueued since the last call, measured with hipEvents on the launch
 * stream (ms, summed) and their count; used by bench.py for the roofline line.  With batches in flight on several streams the spans overlap (their
 * sum exceeds the wall time and each contains its neighbours' work): bench.py takes kernel times from a plan created with mc_tail_overlap = 2. */
int qldpc_cc_plan_kernel_time(qldpc_cc_plan *plan, double *ms_total, int64_t *launches);
/* the part of that total spent in the bit-sliced first-iteration kernel of the reference-semantics pipeline (csrc/mc_first.hip; 0 for plans
 * that do not use it).  Call before qldpc_cc_plan_kernel_time, which resets both sums. */
int qldpc_cc_plan_first_iteration_time(qldpc_cc_plan *plan, double *ms_first);
/* shader clock (MHz) held under the last decode launch (plans created with QLDPC_FLAG_CLOCK_PROBE; fused regular kernel only):
 * median over workgroups of delta(s_memtime) / delta(s_memrealtime) x 100 MHz.  Synchronises `stream`. */
int qldpc_cc_plan_clock(qldpc_cc_plan *plan, void *stream, double *mhz);
void qldpc_cc_plan_destroy(qldpc_cc_plan *plan);

/* ---- circuit-level Monte-Carlo (BASELINE config 5) ------------------------------------------------------------
 * Compiled syndrome-extraction circuit in the wire format of the reference's CompiledCircuit (src/noise/compiled.py:116-173;
 * op codes src/noise/constants.py:8-14).  All pointers are host pointers, copied at plan creation. */
typedef struct {
    int64_t base_len, suffix_len;                       /* noisy part (cycle * num_cycles) and noiseless suffix (cycle * 2) */
    const int32_t *base_ops, *base_q1, *base_q2, *suffix_ops, *suffix_q1, *suffix_q2;
    int32_t total_qubits, num_x_checks, num_z_checks, n_data, k, reserved;
    const int32_t *x_syn_positions, *x_syn_ptrs;        /* CSR: X check -> indices of its MeasX outcomes (compiled.py:72-103) */
    const int32_t *z_syn_positions, *z_syn_ptrs;
    const int32_t *data_qubit_indices;                  /* [n_data] */
    const uint8_t *Lx, *Lz;                             /* logical operators, dense k x n_data */
} qldpc_circuit_desc;

/* Single-fault signatures of one sector: the batched form of the per-fault simulations in src/noise/builder.py:37-66
 * (_simulate_Z_from_spec / _simulate_X_from_spec).  Entry e = 2 * base_op_index + slot (slot 1 = CNOT target). */
int qldpc_circuit_fault_signatures(const qldpc_circuit_desc *circuit, int sector_is_x, int32_t *ptr, uint16_t *idx, int64_t idx_cap,
                                   uint64_t *logmask, int64_t *idx_needed);

/* ---- f4: the trial loops of the alpha / beta estimators, batched ------------------------------------------------------------
 * Replaces the per-trial Python loops of estimate_alpha_alvarado (src/decoding/alpha.py:119-137),
 * estimate_alpha_alvarado_autoregressive (alpha.py:206-255, one call per iteration index) and estimate_scopt_beta
 * (src/decoding/scopt.py:80-134).  errors: int8[B][n], drawn by the caller (the reference draws them from the caller's numpy
 * Generator); syndromes, decoder state, samples and histograms live on the device.
 *   QLDPC_STATS_CHECK_MESSAGES  samples = R_flat of one check pass with alpha = 1 taken after `iters` decoder iterations that
 *                               use alpha_k of the given alpha mode (iters = 0 is alpha.py:119-137); class = error[col[edge]].
 *   QLDPC_STATS_POSTERIOR       samples = the posterior `values` the decoder stops with (early exit, at most `iters`
 *                               iterations; scopt.py:88-131); class = error[j].
 * range[0..1] = min / max over the finite samples of both classes (alpha.py:29-31), finite[c] = number of finite samples of
 * class c (alpha.py:23-27).  qldpc_msgstats_histogram then bins them with np.histogram's rule for the given strictly increasing
 * edges (bins + 1 values): edges[i] <= x < edges[i+1], last bin closed; hist0 / hist1: int64[bins]. */
#define QLDPC_STATS_CHECK_MESSAGES 0
#define QLDPC_STATS_POSTERIOR 1
typedef struct qldpc_msgstats qldpc_msgstats;
int qldpc_msgstats_create(const qldpc_graph *g, int64_t B, const int8_t *errors, const double *prior, int kind, int iters,
                          int alpha_mode, double alpha_val, const double *alpha_seq, int alpha_len, double damping, double clip_llr,
                          double *range, int64_t *finite, qldpc_msgstats **out);
int qldpc_msgstats_histogram(qldpc_msgstats *stats, const double *edges, int bins, int64_t *hist0, int64_t *hist1);
void qldpc_msgstats_destroy(qldpc_msgstats *stats);

typedef struct qldpc_circuit_plan qldpc_circuit_plan;
/* a13 + a14: run_trial_fast (src/noise/simulation.py:21-107) + _run_single_trial_fast and the tally
 * (src/simulation/engine.py:68-122, 450-457), batched.  gz / gx: Tanner graphs of HdecZ / HdecX; prior_*: LLRs of
 * engine.py:210-212; logmask_*[j]: bit r set iff logical row r of H*_full has a one in column j (engine.py:99,119).
 * Trials are addressed by a global index (Philox streams), so any split over calls / devices gives the same tally. */
int qldpc_circuit_plan_create(const qldpc_circuit_desc *circuit, const qldpc_graph *gz, const qldpc_graph *gx, const double *prior_z,
                              const double *prior_x, const uint64_t *logmask_z, const uint64_t *logmask_x, double p, int max_iter,
                              int alpha_mode, double alpha_val_z, double alpha_val_x, const double *alpha_seq_z, int alpha_len_z,
                              const double *alpha_seq_x, int alpha_len_x, double damping, double clip_llr, int use_osd, int flags,
                              int64_t batch, qldpc_circuit_plan **out);
int qldpc_circuit_plan_run(qldpc_circuit_plan *plan, uint64_t seed, int64_t trial_begin, int64_t count, void *stream);
/* Same pass, and additionally the per-trial verdicts in trial order: outcome[i] bit0 = z_err, bit1 = x_err of trial
 * trial_begin + i (host buffer, `count` bytes; the call synchronises `stream`).  This is what the reference's in-order
 * early stop needs (src/simulation/engine.py:441-464: stop at the trial where the target-th logical error occurs). */
int qldpc_circuit_plan_run_outcomes(qldpc_circuit_plan *plan, uint64_t seed, int64_t trial_begin, int64_t count, void *stream,
                                    uint8_t *outcome);
int qldpc_circuit_plan_read(qldpc_circuit_plan *plan, void *stream, int clear, int64_t *tally);
/* hipEvent time (ms, summed over the batches enqueued since the last call) of each phase of the per-trial pipeline of
 * src/simulation/engine.py:68-122, and the number of batches.  Sector X runs on the plan's own stream beside sector Z (unless the plan
 * was created with QLDPC_FLAG_MC_UNFUSED), so the phase spans overlap and their sum exceeds the wall time. */
#define QLDPC_CIRCUIT_PHASES 6
#define QLDPC_PHASE_SAMPLE 0   /* run_trial_fast, engine.py:75 */
#define QLDPC_PHASE_BP_Z 1     /* engine.py:84-94 */
#define QLDPC_PHASE_OSD_Z 2    /* engine.py:96-97 */
#define QLDPC_PHASE_BP_X 3     /* engine.py:103-113 */
#define QLDPC_PHASE_OSD_X 4    /* engine.py:115-116 */
#define QLDPC_PHASE_JUDGE 5    /* engine.py:99-100,119-122 + tally */
int qldpc_circuit_plan_phase_times(qldpc_circuit_plan *plan, double *ms /* [QLDPC_CIRCUIT_PHASES] */, int64_t *batches);
/* shader clock (MHz) held under the sector-Z decode kernel [0] and OSD-0 kernel [1] of the last batch (plans created with
 * QLDPC_FLAG_CLOCK_PROBE): median over workgroups of delta(s_memtime) / delta(s_memrealtime) x 100 MHz.  Synchronises `stream`. */
int qldpc_circuit_plan_clock(qldpc_circuit_plan *plan, void *stream, double *mhz /* [2] */);
/* the sampler alone = batched run_trial_fast: sparse_z int8[count][#MeasX], true_z int8[count][k], sparse_x, true_x (host) */
int qldpc_circuit_plan_sample(qldpc_circuit_plan *plan, uint64_t seed, int64_t trial_begin, int64_t count, int8_t *sparse_z,
                              int8_t *true_z, int8_t *sparse_x, int8_t *true_x);
void qldpc_circuit_plan_destroy(qldpc_circuit_plan *plan);

/* ---- (e) multi-GPU: the one collective of the path, natively on RCCL --------------------------------------------------------
 * Sum of the int64[QLDPC_TALLY_SLOTS] tally over the GPUs of a node; replaces the Python loop that sums the workers' results in
 * src/simulation/engine.py:450-457.  librccl is loaded on the first qldpc_comm_* call.  Two ways to form the communicator:
 *   qldpc_comm_init_all(ndev, devices, &comm)            one process drives `ndev` GPUs (ncclCommInitAll); devices NULL = 0..ndev-1
 *   qldpc_comm_unique_id(id) on rank 0, id handed to the other ranks by the launcher, then
 *   qldpc_comm_init_rank(nranks, rank, id, device, &comm)  one process per GPU (ncclCommInitRank)
 * qldpc_tally_allreduce(comm, tallies): host int64[nlocal][QLDPC_TALLY_SLOTS] (nlocal = ndev after init_all, 1 after init_rank);
 * on return every row holds the sum over all ranks.  The _dev form reduces a device-resident tally in place on `stream` without
 * synchronising; with several local ranks, issue the calls of all local ranks between qldpc_comm_group_begin / _end. */
#define QLDPC_COMM_ID_BYTES 128
typedef struct qldpc_comm qldpc_comm;
int qldpc_comm_init_all(int ndev, const int *devices, qldpc_comm **out);
int qldpc_comm_unique_id(uint8_t *id /* [QLDPC_COMM_ID_BYTES] */);
int qldpc_comm_init_rank(int nranks, int rank, const uint8_t *id, int device, qldpc_comm **out);
int qldpc_comm_size(const qldpc_comm *comm, int *nranks, int *nlocal);
int qldpc_tally_allreduce(qldpc_comm *comm, int64_t *tallies);
int qldpc_tally_allreduce_dev(qldpc_comm *comm, int local_rank, int64_t *d_tally, void *stream);
int qldpc_comm_group_begin(void);
int qldpc_comm_group_end(void);
void qldpc_comm_destroy(qldpc_comm *comm);

/* Philox4x32-10 reference vector helper (host; lets tests pin the generator against the oracle) */
void qldpc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* QLDPC_HIP_H */
