// Process-wide kernel-selection switches (qldpc_set_option): state and the entry point.  Results never depend on them.
#include "common.h"
#include "minsum_common.h"

#include <atomic>
#include <cstring>

namespace qldpc {

static std::atomic<int> g_opt_kernel{0}, g_opt_first{1}, g_opt_tail{1}, g_opt_presort{-1};
int osd_presort_choice() { return g_opt_presort.load(); }
int wave_kernel_choice() { return g_opt_kernel.load(); }
int mc_first_choice() { return g_opt_first.load(); }
int mc_tail_overlap_choice() { return g_opt_tail.load(); }
#ifdef QLDPC_EXPERIMENTS
extern std::atomic<int> g_opt_wave_cpl, g_opt_wave_rst, g_opt_wave_grid;      // minsum_wave.hip
#endif

}  // namespace qldpc

// Process-wide switches for tools/ and the parity tests (results never depend on them; the defaults are what bench.py measures):
//   "mc_first_iteration"  reference-semantics Monte-Carlo plans: 1 = bit-sliced first iteration + full decoder on the shots it lists (default),
//                         0 = full decoder for every shot
//   "mc_first_bits"       shots per lane of that kernel: 8 (default), 16, 32
//   "mc_tail_overlap"     read at plan creation.  1 (default) = whole batches on the plan's own streams (3 for large batches under reference semantics, 8 for
//                         batches <= 32768), 2 = decode on the caller's stream and only OSD-0 + judge of a batch on a side stream, 0 = one stream
//   "osd_presort"         read at an OSD-0 launch: columns of the order that the free-pivot kernels sort up front (default -1 = automatic: m rounded up to whole
//                         chunks of 1024; the columns behind are ordered only if a sweep gets there; 0 = the whole order up front); tests/ use 0 and short heads on purpose
// experiments build only (libqldpc_hip_experiments.so):
//   "regular_kernel"      0 / 1 = the 72-thread-team kernel (minsum_regular.hip), 2 = the wave-private kernel (minsum_wave.hip) where eligible
//   "wave_cpl" / "wave_rst" / "wave_grid"  its checks per lane (0 = automatic, 4, 5, 6, 9), message-row stride in doubles (0, 6, 7), waves per CU (0 .. 32)
QLDPC_EXPORT int qldpc_set_option(const char *name, int value) {
    QLDPC_REQUIRE(name != nullptr, "name is NULL");
    if (!std::strcmp(name, "mc_first_iteration")) { QLDPC_REQUIRE(value == 0 || value == 1, "mc_first_iteration: 0 or 1"); qldpc::g_opt_first = value; return QLDPC_OK; }
    if (!std::strcmp(name, "mc_tail_overlap")) { QLDPC_REQUIRE(value >= 0 && value <= 2, "mc_tail_overlap: 0, 1 or 2"); qldpc::g_opt_tail = value; return QLDPC_OK; }
    if (!std::strcmp(name, "mc_first_bits")) { QLDPC_REQUIRE(value == 8 || value == 16 || value == 32, "mc_first_bits: 8, 16 or 32"); qldpc::mc_first_set_bits(value); return QLDPC_OK; }
    if (!std::strcmp(name, "osd_presort")) { QLDPC_REQUIRE(value >= -1 && value <= 65535, "osd_presort: -1 .. 65535"); qldpc::g_opt_presort = value; return QLDPC_OK; }
    if (!std::strcmp(name, "mc_big_lanes")) { QLDPC_REQUIRE(value >= 2 && value <= 8, "mc_big_lanes: 2 .. 8"); qldpc::mc_set_big_lanes(value); return QLDPC_OK; }
    if (!std::strcmp(name, "mc_list_shots")) { QLDPC_REQUIRE(value >= 0 && value <= 16, "mc_list_shots: 0 .. 16"); qldpc::regular_set_list_shots(value); return QLDPC_OK; }
    const bool wave_opt = !std::strcmp(name, "regular_kernel") || !std::strcmp(name, "wave_cpl") || !std::strcmp(name, "wave_rst") || !std::strcmp(name, "wave_grid");
#ifdef QLDPC_EXPERIMENTS
    if (!std::strcmp(name, "regular_kernel")) { QLDPC_REQUIRE(value >= 0 && value <= 2, "regular_kernel: 0, 1 or 2"); qldpc::g_opt_kernel = value; return QLDPC_OK; }
    if (!std::strcmp(name, "wave_cpl")) { QLDPC_REQUIRE(value == 0 || value == 4 || value == 5 || value == 6 || value == 9, "wave_cpl: 0, 4, 5, 6 or 9"); qldpc::g_opt_wave_cpl = value; return QLDPC_OK; }
    if (!std::strcmp(name, "wave_rst")) { QLDPC_REQUIRE(value == 0 || value == 6 || value == 7, "wave_rst: 0, 6 or 7"); qldpc::g_opt_wave_rst = value; return QLDPC_OK; }
    if (!std::strcmp(name, "wave_grid")) { QLDPC_REQUIRE(value >= 0 && value <= 32, "wave_grid: 0 .. 32"); qldpc::g_opt_wave_grid = value; return QLDPC_OK; }
#else
    if (wave_opt) {
        if (value == 0) return QLDPC_OK;             // the default is the only setting of the product library
        qldpc::set_error("option '%s': the wave-private kernel is a measured-and-rejected experiment (libqldpc_hip_experiments.so, make experiments)", name);
        return QLDPC_ERR_UNSUPPORTED;
    }
#endif
    (void)wave_opt;
    qldpc::set_error("unknown option '%s'", name);
    return QLDPC_ERR_INVALID;
}

// Streams for hosts that do not bring their own runtime (run_simulation(num_workers = N): one plan and one stream per worker).
QLDPC_EXPORT int qldpc_stream_create(int device, void **stream) {
    QLDPC_REQUIRE(stream != nullptr, "stream is NULL");
    QLDPC_USE_DEVICE(device);
    hipStream_t s = nullptr;
    QLDPC_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_stream_sync(int device, void *stream) {
    QLDPC_USE_DEVICE(device);
    QLDPC_HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_stream_destroy(int device, void *stream) {
    if (!stream) return QLDPC_OK;
    QLDPC_USE_DEVICE(device);
    QLDPC_HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return QLDPC_OK;
}
