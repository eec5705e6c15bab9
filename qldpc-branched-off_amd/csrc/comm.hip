// The one collective of the path, natively on RCCL: sum of the int64[QLDPC_TALLY_SLOTS] tally over the GPUs of a node
// (replaces the Python tally loop of the reference's process pool, src/simulation/engine.py:450-457).
//
// librccl is resolved with dlopen at the first qldpc_comm_* call (the decoder itself never needs it), preferring a copy the
// process has already loaded (e.g. the one bundled with torch) over a second one.  128 bytes per rank: the all-reduce is
// latency bound, so nothing here is tuned for bandwidth; it only has to be correct and not require Python.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace qldpc {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi *rccl() {
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;      // already in the process?
    if (!h) for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!h) { set_error("librccl could not be loaded: %s", dlerror()); return nullptr; }
    auto sym = [&](const char *n) { return dlsym(h, n); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommInitAll || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
        set_error("librccl lacks an expected symbol");
        return nullptr;
    }
    api.handle = h;
    return &api;
}

#define QLDPC_RCCL_TRY(api, expr)                                                                  \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            qldpc::set_error("%s failed: %s", #expr, (api)->GetErrorString(_r));                   \
            return QLDPC_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

}  // namespace qldpc

using namespace qldpc;

// One communicator object holds the LOCAL ranks of this process: all `ndev` devices after qldpc_comm_init_all (single process driving
// every GPU), exactly one after qldpc_comm_init_rank (one process per GPU).
struct qldpc_comm {
    int nranks = 0;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<int64_t *> bufs;          // device int64[QLDPC_TALLY_SLOTS] per local rank
};

static int comm_alloc_local(qldpc_comm *c) {
    for (size_t i = 0; i < c->devices.size(); i++) {
        QLDPC_HIP_TRY(hipSetDevice(c->devices[i]));
        hipStream_t s = nullptr;
        int64_t *b = nullptr;
        QLDPC_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->streams.push_back(s);
        QLDPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&b), QLDPC_TALLY_SLOTS * sizeof(int64_t)));
        c->bufs.push_back(b);
    }
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_comm_init_all(int ndev, const int *devices, qldpc_comm **out) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    QLDPC_REQUIRE(ndev >= 1, "ndev must be >= 1");
    QLDPC_USE_DEVICE(devices ? devices[0] : 0);
    int count = 0;
    QLDPC_HIP_TRY(hipGetDeviceCount(&count));
    QLDPC_REQUIRE(ndev <= count, "%d devices requested, %d visible", ndev, count);
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    qldpc_comm *c = new qldpc_comm();
    c->nranks = ndev;
    for (int i = 0; i < ndev; i++) c->devices.push_back(devices ? devices[i] : i);
    c->comms.assign(ndev, nullptr);
    ncclResult_t r = api->CommInitAll(c->comms.data(), ndev, c->devices.data());
    if (r != ncclSuccess) { set_error("ncclCommInitAll failed: %s", api->GetErrorString(r)); delete c; return QLDPC_ERR_HIP; }
    int rc = comm_alloc_local(c);
    if (rc != QLDPC_OK) { qldpc_comm_destroy(c); return rc; }
    *out = c;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_comm_unique_id(uint8_t *id128) {
    QLDPC_REQUIRE(id128 != nullptr, "id buffer is NULL");
    static_assert(sizeof(ncclUniqueId) == QLDPC_COMM_ID_BYTES, "ncclUniqueId size");
    QLDPC_USE_DEVICE(0);
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    ncclUniqueId id;
    QLDPC_RCCL_TRY(api, api->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_comm_init_rank(int nranks, int rank, const uint8_t *id128, int device, qldpc_comm **out) {
    QLDPC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    QLDPC_REQUIRE(id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    QLDPC_USE_DEVICE(device);
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    qldpc_comm *c = new qldpc_comm();
    c->nranks = nranks;
    c->devices.push_back(device);
    c->comms.assign(1, nullptr);
    ncclResult_t r = api->CommInitRank(&c->comms[0], nranks, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", api->GetErrorString(r)); delete c; return QLDPC_ERR_HIP; }
    int rc = comm_alloc_local(c);
    if (rc != QLDPC_OK) { qldpc_comm_destroy(c); return rc; }
    *out = c;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_comm_size(const qldpc_comm *c, int *nranks, int *nlocal) {
    QLDPC_REQUIRE(c != nullptr, "comm is NULL");
    if (nranks) *nranks = c->nranks;
    if (nlocal) *nlocal = (int)c->devices.size();
    return QLDPC_OK;
}

// tallies: host int64[nlocal][QLDPC_TALLY_SLOTS]; on return every row holds the sum over ALL ranks of the communicator.
QLDPC_EXPORT int qldpc_tally_allreduce(qldpc_comm *c, int64_t *tallies) {
    QLDPC_REQUIRE(c != nullptr && tallies != nullptr, "NULL argument");
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    const int nl = (int)c->devices.size();
    QLDPC_USE_DEVICE(c->devices[0]);
    const size_t bytes = QLDPC_TALLY_SLOTS * sizeof(int64_t);
    for (int i = 0; i < nl; i++) {
        QLDPC_HIP_TRY(hipSetDevice(c->devices[i]));
        QLDPC_HIP_TRY(hipMemcpyAsync(c->bufs[i], tallies + (size_t)i * QLDPC_TALLY_SLOTS, bytes, hipMemcpyHostToDevice, c->streams[i]));
    }
    QLDPC_RCCL_TRY(api, api->GroupStart());
    for (int i = 0; i < nl; i++) {
        ncclResult_t r = api->AllReduce(c->bufs[i], c->bufs[i], QLDPC_TALLY_SLOTS, ncclInt64, ncclSum, c->comms[i], c->streams[i]);
        if (r != ncclSuccess) { (void)api->GroupEnd(); set_error("ncclAllReduce failed: %s", api->GetErrorString(r)); return QLDPC_ERR_HIP; }
    }
    QLDPC_RCCL_TRY(api, api->GroupEnd());
    for (int i = 0; i < nl; i++) {
        QLDPC_HIP_TRY(hipSetDevice(c->devices[i]));
        QLDPC_HIP_TRY(hipMemcpyAsync(tallies + (size_t)i * QLDPC_TALLY_SLOTS, c->bufs[i], bytes, hipMemcpyDeviceToHost, c->streams[i]));
        QLDPC_HIP_TRY(hipStreamSynchronize(c->streams[i]));
    }
    return QLDPC_OK;
}

// Device form for one local rank: d_tally (device int64[QLDPC_TALLY_SLOTS] on that rank's GPU) is summed in place over all ranks,
// enqueued on `stream`; does not synchronise.  Every rank of the communicator must call it (for a communicator with several local
// ranks, from one thread between qldpc_comm_group_begin / _end).
QLDPC_EXPORT int qldpc_tally_allreduce_dev(qldpc_comm *c, int local_rank, int64_t *d_tally, void *stream) {
    QLDPC_REQUIRE(c != nullptr && d_tally != nullptr, "NULL argument");
    QLDPC_REQUIRE(local_rank >= 0 && local_rank < (int)c->devices.size(), "local rank %d out of range", local_rank);
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    QLDPC_USE_DEVICE(c->devices[local_rank]);
    QLDPC_RCCL_TRY(api, api->AllReduce(d_tally, d_tally, QLDPC_TALLY_SLOTS, ncclInt64, ncclSum, c->comms[local_rank], reinterpret_cast<hipStream_t>(stream)));
    return QLDPC_OK;
}
QLDPC_EXPORT int qldpc_comm_group_begin(void) {
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    QLDPC_RCCL_TRY(api, api->GroupStart());
    return QLDPC_OK;
}
QLDPC_EXPORT int qldpc_comm_group_end(void) {
    RcclApi *api = rccl();
    if (!api) return QLDPC_ERR_UNSUPPORTED;
    QLDPC_RCCL_TRY(api, api->GroupEnd());
    return QLDPC_OK;
}

QLDPC_EXPORT void qldpc_comm_destroy(qldpc_comm *c) {
    if (!c) return;
    RcclApi *api = rccl();
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < c->devices.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) { (void)hipStreamSynchronize(c->streams[i]); (void)hipStreamDestroy(c->streams[i]); }
        if (i < c->bufs.size() && c->bufs[i]) (void)hipFree(c->bufs[i]);
        if (api && i < c->comms.size() && c->comms[i]) (void)api->CommDestroy(c->comms[i]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
}
