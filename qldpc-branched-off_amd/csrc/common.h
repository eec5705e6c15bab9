// Shared host-side infrastructure of libqldpc_hip (error reporting, device buffers, graph handle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/qldpc_hip.h"

#define QLDPC_EXPORT extern "C" __attribute__((visibility("default")))

namespace qldpc {

void set_error(const char *fmt, ...);

#define QLDPC_HIP_TRY(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            qldpc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return QLDPC_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

#define QLDPC_REQUIRE(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            qldpc::set_error(__VA_ARGS__);        \
            return QLDPC_ERR_INVALID;             \
        }                                         \
    } while (0)

// Zeroes device memory and WAITS for it.  hipMemset returns once the fill is queued on the null stream, and the null stream is not ordered against
// hipStreamNonBlocking streams: a kernel enqueued on such a stream right after a bare hipMemset can start before the fill has run.  Round 4 found
// a lane's OSD-0 ticket counter zeroed in the middle of the lane's first launch that way (profiles/r04_experiments.txt, item 5; the hand-out in
// isolation: tools/microbench/ticket_race.hip).  Every "zero this before anyone uses it" in the library goes through here.
inline hipError_t zero_now(void *p, size_t bytes) {
    const hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// Checks that a gfx9 device is present and selects it. Returns QLDPC_OK or QLDPC_ERR_NO_DEVICE.
int use_device(int device);

// Entry points select their handle's device for the calling thread and put the caller's current device back on return
// (a process that also drives the GPU through another runtime, e.g. torch, keeps its own current device).
struct DeviceScope {
    int prev = -1;
    int enter(int device);
    ~DeviceScope();
};
#define QLDPC_USE_DEVICE(dev)                                   \
    qldpc::DeviceScope _dev_scope;                              \
    {                                                           \
        const int _rc_dev = _dev_scope.enter(dev);              \
        if (_rc_dev != QLDPC_OK) return _rc_dev;                \
    }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set it once per (device, kernel), thread-safe.
int ensure_max_lds(int device, const void *func, int bytes);

// Grow-only device buffer (never shrinks; freed with the owner).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// RAII temporary device allocation for the host-pointer entry points.
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); p = nullptr; return QLDPC_ERR_HIP; }
        return QLDPC_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// Shader-clock probe (QLDPC_FLAG_CLOCK_PROBE): thread 0 of a workgroup stamps the shader-clock counter (s_memtime) and the constant
// 100 MHz counter (s_memrealtime) when it starts and when it ends; clock held under this kernel's load = delta ratio x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  Buffer: kClkSlots pairs (delta memtime, delta memrealtime) indexed by blockIdx.x.
constexpr int kClkSlots = 512;
struct ClkStamp { unsigned long long t = 0, r = 0; };
__device__ __forceinline__ ClkStamp clk_begin(const unsigned long long *clk) {
    ClkStamp s;
    if (clk) { s.t = __builtin_amdgcn_s_memtime(); s.r = __builtin_amdgcn_s_memrealtime(); }
    return s;
}
__device__ __forceinline__ void clk_end(unsigned long long *clk, const ClkStamp &s) {
    if (clk && threadIdx.x == 0 && blockIdx.x < kClkSlots) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - s.t;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - s.r;
    }
}
double clock_probe_median(const unsigned long long *pairs, int slots);   // MHz; 0 when nothing was stamped

}  // namespace qldpc

// Tanner graph handle.  Host CSR/CSC plus device copies; immutable after create except the workspace cache.
struct qldpc_graph {
    int m = 0, n = 0, nnz = 0, device = 0;
    int max_row_deg = 0, max_col_deg = 0;
    std::vector<int32_t> indptr, indices;        // CSR (rows = checks), sorted columns
    std::vector<int32_t> colptr, rowidx, csc2csr; // CSC view: per column ascending check index; csc2csr[k] = CSR edge id
    std::vector<int32_t> csr2csc;                 // inverse permutation
    // device copies
    int32_t *d_indptr = nullptr, *d_indices = nullptr, *d_colptr = nullptr, *d_rowidx = nullptr, *d_csc2csr = nullptr,
            *d_csr2csc = nullptr;
    // ELL (slot-major) views for the workgroup-per-shot kernel: coalesced index loads across rows / columns
    uint16_t *d_ell_col = nullptr;   // [round_up(max_row_deg, 8)][m]  column of the k-th edge of row i; unused slots hold column 0
    uint32_t *d_ell_var = nullptr;   // [max_col_deg][n]  (row << 8) | position-in-row of the d-th edge of column j (ascending rows)
    // the same views with rows / columns handed to threads in DEGREE order (stable, descending): a wave's rows (columns) then have one
    // degree and its edge loop has no idle lanes.  Check state and index entries are in row-SLOT space; posteriors stay in column space.
    int32_t *d_row_of_slot = nullptr, *d_col_of_slot = nullptr;   // [m], [n]
    uint8_t *d_deg_of_rslot = nullptr, *d_deg_of_row = nullptr;      // [m]
    uint16_t *d_deg_of_cslot = nullptr, *d_deg_of_col = nullptr;     // [n]
    uint16_t *d_ell_col_s = nullptr; // [round_up(max_row_deg, 8)][m] indexed by row slot
    uint32_t *d_ell_var_s = nullptr; // [max_col_deg][n] indexed by column slot: (row slot << 8) | position-in-row, ascending ROW order
    int32_t *d_identity = nullptr;   // [max(m, n)] 0, 1, 2, ... (the natural-order "permutation")
    // Device workspaces of the decode / OSD kernels.  `mu` guards the bookkeeping while launches are enqueued; the buffers themselves
    // are protected in STREAM order: every user calls ws_acquire(stream) before its launches and ws_release(stream) after them, so a
    // launch on another stream first waits (hipStreamWaitEvent) for the previous user of the workspaces to finish.
    mutable std::mutex mu;
    mutable qldpc::DevBuf ws_msg, ws_qold, ws_vals, ws_misc, ws_queue, ws_list, ws_prior, ws_redo;
    mutable qldpc::DevBuf ws_squeue;  // work queue of the one-wave OSD-0 kernels (osd_small.hip): zeroed once, the kernels reset it themselves
    mutable bool ws_private = false;  // the handle is used from ONE stream only (a private copy owned by a plan lane): no hand-over events
    mutable hipEvent_t ws_event = nullptr;
    mutable hipStream_t ws_stream = nullptr;
    mutable bool ws_used = false;
    int ws_acquire(hipStream_t stream) const;     // callers hold mu
    int ws_release(hipStream_t stream) const;
    mutable std::mutex mu_io;        // host-pointer entry points: serialises use of ws_io (taken before mu)
    mutable qldpc::DevBuf ws_io;
    // alpha tables already on the device (guarded by mu).  A table is uploaded once from a pinned host copy and never overwritten, so
    // the *_dev entry points stay pure enqueues (no synchronisation) however the alpha schedule changes between calls.
    struct AlphaEntry { std::vector<double> host; double *pinned = nullptr; double *dev = nullptr; hipEvent_t ready = nullptr; hipStream_t stream = nullptr; };
    mutable std::vector<AlphaEntry> alpha_cache;
    int alpha_table(const std::vector<double> &tab, hipStream_t stream, const double **d_out) const;   // callers hold mu
    mutable void *pin = nullptr;     // pinned host staging for small results
    mutable size_t pin_cap = 0;
    mutable int gf2_rank = -1;       // rank of H over GF(2), computed on first OSD use
    mutable uint16_t *d_col_rows = nullptr;   // [n][max_col_deg] rows of every column (ascending, padded with m), built on first OSD use
    mutable void *wg2_cache = nullptr;   // tables of the LDS-resident workgroup decoder per prior (minsum_wg2.hip), built on first use (guarded by mu)
    mutable unsigned long long *clk_probe = nullptr;   // set (under mu) by a plan created with QLDPC_FLAG_CLOCK_PROBE around one launch
};
