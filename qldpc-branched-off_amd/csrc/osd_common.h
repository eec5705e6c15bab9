// Shared pieces of the OSD-0 kernels (gf2.hip, osd_fwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Phase timers of the OSD kernels exist only in the diagnostic build (`make timers` -> libqldpc_hip_timers.so, -DQLDPC_OSD_TIMERS);
// the default build carries no clock reads.  Counters (uint64[16]): [0] shots, [1] chunks, [2] columns taken into blocks, [3] pivots,
// [4] cycles, [5] kill passes, [6] blocks, [8] sort, [9] phase 1 (reduce columns), [10] phase 2 (block pivots), [11] phase 3 (row updates),
// [12] dependent-column tests, [13] back-substitution.
#ifdef QLDPC_OSD_TIMERS
#define OSD_CLOCK() clock64()
#else
#define OSD_CLOCK() 0ll
#endif

namespace qldpc {
unsigned long long *osd_timer_buffer();      // device buffer of the current device, NULL in the default build
}
