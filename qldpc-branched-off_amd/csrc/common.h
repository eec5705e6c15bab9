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

// Checks that a gfx9 device is present and selects it. Returns QLDPC_OK or QLDPC_ERR_NO_DEVICE.
int use_device(int device);

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
    // workspace cache for the decode kernels (guarded by mu; one decode at a time per graph handle)
    mutable std::mutex mu;
    mutable qldpc::DevBuf ws_msg, ws_qold, ws_vals, ws_alpha, ws_misc, ws_queue, ws_list;
    mutable std::mutex mu_io;        // host-pointer entry points: serialises use of ws_io (taken before mu)
    mutable qldpc::DevBuf ws_io;
    mutable std::vector<double> alpha_host;   // alpha table currently in ws_alpha (guarded by mu)
    mutable void *pin = nullptr;     // pinned host staging for small results
    mutable size_t pin_cap = 0;
    mutable int gf2_rank = -1;       // rank of H over GF(2), computed on first OSD use
};
