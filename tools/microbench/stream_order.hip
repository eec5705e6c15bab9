// Do two kernels enqueued back to back on ONE stream ever overlap when many streams are busy?  (round 4: a plan's eight lane streams on
// GPU_MAX_HW_QUEUES = 16 hardware queues gave racy tallies; 4 queues did not.)  Kernel A of a pair writes a buffer slowly, kernel B checks it.
//   hipcc --offload-arch=gfx950 -O2 stream_order.hip -o build/stream_order ;  GPU_MAX_HW_QUEUES=16 build/stream_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void writer(int *buf, int n, int tag, int spin) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        long long t0 = clock64();
        while (clock64() - t0 < spin) {}
        buf[i] = tag;
    }
}
__global__ void checker(const int *buf, int n, int tag, unsigned long long *bad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (buf[i] != tag) atomicAdd(bad, 1ull);
}
int main() {
    const int NS = 8, N = 1 << 16, ROUNDS = 400;
    std::vector<hipStream_t> st(NS);
    std::vector<int *> buf(NS);
    unsigned long long *bad;
    hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    for (int s = 0; s < NS; s++) { hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking); hipMalloc(&buf[s], N * 4); hipMemset(buf[s], 0, N * 4); }
    hipDeviceSynchronize();
    for (int r = 1; r <= ROUNDS; r++)
        for (int s = 0; s < NS; s++) {
            hipLaunchKernelGGL(writer, dim3(64), dim3(256), 0, st[s], buf[s], N, r, 2000);
            hipLaunchKernelGGL(checker, dim3(128), dim3(64), 0, st[s], buf[s], N, r, bad);
        }
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("stale reads: %llu of %d checks\n", h, NS * ROUNDS * N);
    return h ? 1 : 0;
}
