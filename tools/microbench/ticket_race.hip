// The tail of a Monte-Carlo piece in isolation (round 4: duplicate work hand-out under GPU_MAX_HW_QUEUES=16): per stream, a producer kernel lists T records
// through an atomic counter, a consumer kernel hands them out (workgroup b takes entry b, then tickets) and its last workgroup resets the counters.
//   hipcc --offload-arch=gfx950 -O2 ticket_race.hip -o build/ticket_race ;  GPU_MAX_HW_QUEUES=16 build/ticket_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void produce(int *count, int *list, int *rec, int T, int r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) { const int f = atomicAdd(&count[0], 1); list[f] = f; rec[f] = r; }
}
__global__ __launch_bounds__(64) void consume(int *count, int *queue, const int *list, const int *rec, int *mark, int r, int spin, unsigned long long *stats) {
    const int lane = threadIdx.x;
    const int total = count[0];
    for (int item = blockIdx.x; item < total;) {
        const int f = list[item];
        long long t0 = clock64();
        while (clock64() - t0 < spin) {}
        if (lane == 0) {
            atomicAdd(&stats[0], 1ull);
            if (rec[f] != r) atomicAdd(&stats[1], 1ull);
            if (atomicExch(&mark[item], r) == r) atomicAdd(&stats[2], 1ull);
        }
        int t = 0;
        if (lane == 0) t = atomicAdd(queue, 1);
        item = (int)gridDim.x + __builtin_amdgcn_readfirstlane(t);
    }
    if (lane != 0) return;
    if (blockIdx.x == 0) atomicAdd(&stats[3], (unsigned long long)total);
    __threadfence();
    if (atomicAdd(&count[3], 1) == (int)gridDim.x - 1) { count[0] = 0; count[2] = 0; count[3] = 0; *queue = 0; __threadfence(); }
}
int main(int argc, char **argv) {
    const int NS = 8, NMAX = 4096, ROUNDS = argc > 1 ? atoi(argv[1]) : 300, SPIN = argc > 2 ? atoi(argv[2]) : 20000;
    std::vector<hipStream_t> st(NS);
    std::vector<int *> count(NS), queue(NS), list(NS), rec(NS), mark(NS);
    unsigned long long *stats;
    hipMalloc(&stats, 64); hipMemset(stats, 0, 64);
    for (int s = 0; s < NS; s++) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipMalloc(&count[s], 16); hipMemset(count[s], 0, 16);
        hipMalloc(&queue[s], 16); hipMemset(queue[s], 0, 16);
        hipMalloc(&list[s], NMAX * 4); hipMalloc(&rec[s], NMAX * 4); hipMalloc(&mark[s], NMAX * 4);
        hipMemset(mark[s], 0, NMAX * 4);
    }
    hipDeviceSynchronize();
    unsigned long long want = 0;
    srand(7);
    for (int r = 1; r <= ROUNDS; r++)
        for (int s = 0; s < NS; s++) {
            const int T = 1500 + rand() % 1500;
            want += T;
            hipLaunchKernelGGL(produce, dim3(16), dim3(256), 0, st[s], count[s], list[s], rec[s], T, r);
            hipLaunchKernelGGL(consume, dim3(128), dim3(64), 0, st[s], count[s], queue[s], list[s], rec[s], mark[s], r, SPIN, stats);
        }
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, stats, 64, hipMemcpyDeviceToHost);
    printf("expected %llu  visited %llu  listed %llu  stale records %llu  duplicates %llu\n", want, h[0], h[3], h[1], h[2]);
    return (h[0] != want || h[1] || h[2]) ? 1 : 0;
}
