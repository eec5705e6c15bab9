// Error state, device selection and the Tanner-graph handle of libqldpc_hip.
#include "common.h"
#include "minsum_common.h"

#include <algorithm>
#include <cstring>

namespace qldpc {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int use_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libqldpc_hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        (void)hipGetLastError();
        return QLDPC_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (have %d)", device, count);
        return QLDPC_ERR_INVALID;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        return QLDPC_ERR_NO_DEVICE;
    }
    return QLDPC_OK;
}

int DeviceScope::enter(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    const int rc = use_device(device);
    if (rc == QLDPC_OK && cur >= 0) prev = cur;        // recorded unconditionally: an entry point may walk through several devices (comm.hip)
    return rc;
}
DeviceScope::~DeviceScope() {
    int cur = -1;
    if (prev >= 0 && (hipGetDevice(&cur) != hipSuccess || cur != prev)) (void)hipSetDevice(prev);
}

int ensure_max_lds(int device, const void *func, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> done;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &d : done) if (d.first == device && d.second == func) return QLDPC_OK;
    QLDPC_HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.emplace_back(device, func);
    return QLDPC_OK;
}

double clock_probe_median(const unsigned long long *pairs, int slots) {
    std::vector<double> v;
    for (int i = 0; i < slots; i++)
        if (pairs[2 * i + 1] > 0) v.push_back(100.0 * (double)pairs[2 * i] / (double)pairs[2 * i + 1]);
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return QLDPC_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        p = nullptr;
        return QLDPC_ERR_HIP;
    }
    cap = bytes;
    return QLDPC_OK;
}

void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

}  // namespace qldpc

using namespace qldpc;

int qldpc_graph::ws_acquire(hipStream_t stream) const {
    if (ws_private) return QLDPC_OK;
    if (ws_used && stream != ws_stream && ws_event) QLDPC_HIP_TRY(hipStreamWaitEvent(stream, ws_event, 0));
    return QLDPC_OK;
}
int qldpc_graph::ws_release(hipStream_t stream) const {
    if (ws_private) return QLDPC_OK;
    if (!ws_event) QLDPC_HIP_TRY(hipEventCreateWithFlags(&ws_event, hipEventDisableTiming));
    QLDPC_HIP_TRY(hipEventRecord(ws_event, stream));
    ws_stream = stream; ws_used = true;
    return QLDPC_OK;
}

int qldpc_graph::alpha_table(const std::vector<double> &tab, hipStream_t stream, const double **d_out) const {
    for (const AlphaEntry &e : alpha_cache)
        if (e.host == tab) {
            if (stream != e.stream) QLDPC_HIP_TRY(hipStreamWaitEvent(stream, e.ready, 0));   // uploaded on another stream: order behind it
            *d_out = e.dev;
            return QLDPC_OK;
        }
    if (alpha_cache.size() >= 64) {            // a caller cycling through > 64 schedules: start over once everything in flight is done
        QLDPC_HIP_TRY(hipDeviceSynchronize());
        for (AlphaEntry &e : alpha_cache) { if (e.dev) (void)hipFree(e.dev); if (e.pinned) (void)hipHostFree(e.pinned); if (e.ready) (void)hipEventDestroy(e.ready); }
        alpha_cache.clear();
    }
    AlphaEntry e;
    e.host = tab;
    const size_t bytes = std::max<size_t>(tab.size(), 1) * sizeof(double);
    QLDPC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e.pinned), bytes, hipHostMallocDefault));
    std::memcpy(e.pinned, tab.data(), tab.size() * sizeof(double));
    if (hipMalloc(reinterpret_cast<void **>(&e.dev), bytes) != hipSuccess) { (void)hipHostFree(e.pinned); set_error("hipMalloc failed for an alpha table"); return QLDPC_ERR_HIP; }
    QLDPC_HIP_TRY(hipMemcpyAsync(e.dev, e.pinned, tab.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    QLDPC_HIP_TRY(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
    QLDPC_HIP_TRY(hipEventRecord(e.ready, stream));
    e.stream = stream;
    alpha_cache.push_back(e);
    *d_out = e.dev;
    return QLDPC_OK;
}

QLDPC_EXPORT const char *qldpc_last_error(void) { return g_last_error.c_str(); }
QLDPC_EXPORT int qldpc_version(void) { return 100; }

QLDPC_EXPORT int qldpc_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

template <class T>
static int upload(T **dst, const std::vector<T> &src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    QLDPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (!src.empty()) QLDPC_HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_graph_create(int m, int n, const int32_t *indptr, const int32_t *indices, int device, qldpc_graph **out) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    QLDPC_REQUIRE(m >= 0 && n >= 0, "negative dimensions m=%d n=%d", m, n);
    QLDPC_REQUIRE(indptr != nullptr, "indptr is NULL");
    QLDPC_REQUIRE(indptr[0] == 0, "indptr[0] must be 0");
    for (int i = 0; i < m; i++) QLDPC_REQUIRE(indptr[i + 1] >= indptr[i], "indptr not monotone at row %d", i);
    const int nnz = indptr[m];
    QLDPC_REQUIRE(nnz == 0 || indices != nullptr, "indices is NULL");
    for (int i = 0; i < m; i++)
        for (int e = indptr[i]; e < indptr[i + 1]; e++) {
            QLDPC_REQUIRE(indices[e] >= 0 && indices[e] < n, "column index %d out of range in row %d", indices[e], i);
            QLDPC_REQUIRE(e == indptr[i] || indices[e] > indices[e - 1], "row %d: column indices must be strictly increasing (canonical CSR)", i);
        }
    int rc = use_device(device);
    if (rc != QLDPC_OK) return rc;

    qldpc_graph *g = new qldpc_graph();
    g->m = m; g->n = n; g->nnz = nnz; g->device = device;
    g->indptr.assign(indptr, indptr + m + 1);
    g->indices.assign(indices, indices + nnz);
    // CSC view, per column in ascending check order (rows are visited in ascending i)
    g->colptr.assign(n + 1, 0);
    for (int e = 0; e < nnz; e++) g->colptr[indices[e] + 1]++;
    for (int j = 0; j < n; j++) g->colptr[j + 1] += g->colptr[j];
    g->rowidx.resize(nnz); g->csc2csr.resize(nnz); g->csr2csc.resize(nnz);
    std::vector<int32_t> fill(g->colptr.begin(), g->colptr.end() - 1);
    for (int i = 0; i < m; i++) {
        g->max_row_deg = std::max(g->max_row_deg, indptr[i + 1] - indptr[i]);
        for (int e = indptr[i]; e < indptr[i + 1]; e++) {
            const int k = fill[indices[e]]++;
            g->rowidx[k] = i; g->csc2csr[k] = e; g->csr2csc[e] = k;
        }
    }
    for (int j = 0; j < n; j++) g->max_col_deg = std::max(g->max_col_deg, g->colptr[j + 1] - g->colptr[j]);
    rc = upload(&g->d_indptr, g->indptr);
    if (rc == QLDPC_OK) rc = upload(&g->d_indices, g->indices);
    if (rc == QLDPC_OK) rc = upload(&g->d_colptr, g->colptr);
    if (rc == QLDPC_OK) rc = upload(&g->d_rowidx, g->rowidx);
    if (rc == QLDPC_OK) rc = upload(&g->d_csc2csr, g->csc2csr);
    if (rc == QLDPC_OK) rc = upload(&g->d_csr2csc, g->csr2csc);
    if (rc == QLDPC_OK && n < 65535 && m < (1 << 24) && g->max_row_deg < 256) {
        std::vector<uint16_t> ec((size_t)round_up(std::max(g->max_row_deg, 1), 8) * std::max(m, 1), 0);   // slots padded to 8 rows, column 0
        std::vector<uint32_t> ev((size_t)std::max(g->max_col_deg, 1) * std::max(n, 1), 0xFFFFFFFFu);
        for (int i = 0; i < m; i++)
            for (int e = indptr[i]; e < indptr[i + 1]; e++) ec[(size_t)(e - indptr[i]) * m + i] = (uint16_t)indices[e];
        for (int j = 0; j < n; j++)
            for (int k = g->colptr[j]; k < g->colptr[j + 1]; k++) {
                const int row = g->rowidx[k], pos = g->csc2csr[k] - indptr[row];
                ev[(size_t)(k - g->colptr[j]) * n + j] = ((uint32_t)row << 8) | (uint32_t)pos;
            }
        rc = upload(&g->d_ell_col, ec);
        if (rc == QLDPC_OK) rc = upload(&g->d_ell_var, ev);
        // degree-ordered views (stable sort, descending degree)
        std::vector<int32_t> ros(std::max(m, 1)), cos(std::max(n, 1)), slot_of_row(std::max(m, 1));
        for (int i = 0; i < m; i++) ros[i] = i;
        for (int j = 0; j < n; j++) cos[j] = j;
        std::stable_sort(ros.begin(), ros.begin() + m, [&](int a, int b) { return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b]; });
        std::stable_sort(cos.begin(), cos.begin() + n, [&](int a, int b) { return g->colptr[a + 1] - g->colptr[a] > g->colptr[b + 1] - g->colptr[b]; });
        for (int s = 0; s < m; s++) slot_of_row[ros[s]] = s;
        std::vector<uint8_t> drs(std::max(m, 1), 0), dr(std::max(m, 1), 0);
        std::vector<uint16_t> dcs(std::max(n, 1), 0), dc(std::max(n, 1), 0);
        for (int s = 0; s < m; s++) { drs[s] = (uint8_t)(indptr[ros[s] + 1] - indptr[ros[s]]); dr[s] = (uint8_t)(indptr[s + 1] - indptr[s]); }
        for (int s = 0; s < n; s++) { dcs[s] = (uint16_t)std::min(g->colptr[cos[s] + 1] - g->colptr[cos[s]], 65535); dc[s] = (uint16_t)std::min(g->colptr[s + 1] - g->colptr[s], 65535); }
        std::vector<uint16_t> ecs(ec.size(), 0);
        std::vector<uint32_t> evs(ev.size(), 0xFFFFFFFFu);
        for (int s = 0; s < m; s++) {
            const int i = ros[s];
            for (int e = indptr[i]; e < indptr[i + 1]; e++) ecs[(size_t)(e - indptr[i]) * m + s] = (uint16_t)indices[e];
        }
        for (int s = 0; s < n; s++) {
            const int j = cos[s];
            for (int k = g->colptr[j]; k < g->colptr[j + 1]; k++) {
                const int row = g->rowidx[k], pos = g->csc2csr[k] - indptr[row];
                evs[(size_t)(k - g->colptr[j]) * n + s] = ((uint32_t)slot_of_row[row] << 8) | (uint32_t)pos;
            }
        }
        std::vector<int32_t> ident(std::max(std::max(m, n), 1));
        for (size_t i = 0; i < ident.size(); i++) ident[i] = (int32_t)i;
        if (rc == QLDPC_OK) rc = upload(&g->d_row_of_slot, ros);
        if (rc == QLDPC_OK) rc = upload(&g->d_col_of_slot, cos);
        if (rc == QLDPC_OK) rc = upload(&g->d_deg_of_rslot, drs);
        if (rc == QLDPC_OK) rc = upload(&g->d_deg_of_cslot, dcs);
        if (rc == QLDPC_OK) rc = upload(&g->d_deg_of_row, dr);
        if (rc == QLDPC_OK) rc = upload(&g->d_deg_of_col, dc);
        if (rc == QLDPC_OK) rc = upload(&g->d_ell_col_s, ecs);
        if (rc == QLDPC_OK) rc = upload(&g->d_ell_var_s, evs);
        if (rc == QLDPC_OK) rc = upload(&g->d_identity, ident);
    }
    if (rc != QLDPC_OK) { qldpc_graph_destroy(g); return rc; }
    *out = g;
    return QLDPC_OK;
}

QLDPC_EXPORT void qldpc_graph_destroy(qldpc_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (int32_t *p : {g->d_indptr, g->d_indices, g->d_colptr, g->d_rowidx, g->d_csc2csr, g->d_csr2csc})
        if (p) (void)hipFree(p);
    if (g->d_ell_col) (void)hipFree(g->d_ell_col);
    if (g->d_ell_var) (void)hipFree(g->d_ell_var);
    if (g->d_col_rows) (void)hipFree(g->d_col_rows);
    for (void *p : {(void *)g->d_row_of_slot, (void *)g->d_col_of_slot, (void *)g->d_deg_of_rslot, (void *)g->d_deg_of_cslot, (void *)g->d_deg_of_row,
                    (void *)g->d_deg_of_col, (void *)g->d_ell_col_s, (void *)g->d_ell_var_s, (void *)g->d_identity})
        if (p) (void)hipFree(p);
    g->ws_prior.release();
    g->ws_msg.release(); g->ws_qold.release(); g->ws_vals.release(); g->ws_misc.release(); g->ws_io.release(); g->ws_queue.release(); g->ws_squeue.release(); g->ws_list.release(); g->ws_redo.release();
    for (auto &e : g->alpha_cache) { if (e.dev) (void)hipFree(e.dev); if (e.pinned) (void)hipHostFree(e.pinned); if (e.ready) (void)hipEventDestroy(e.ready); }
    if (g->wg2_cache) qldpc::wg2_cache_free(g->wg2_cache);
    if (g->ws_event) (void)hipEventDestroy(g->ws_event);
    if (g->pin) (void)hipHostFree(g->pin);
    delete g;
}

QLDPC_EXPORT int qldpc_graph_dims(const qldpc_graph *g, int *m, int *n, int *nnz) {
    QLDPC_REQUIRE(g != nullptr, "graph is NULL");
    if (m) *m = g->m;
    if (n) *n = g->n;
    if (nnz) *nnz = g->nnz;
    return QLDPC_OK;
}
