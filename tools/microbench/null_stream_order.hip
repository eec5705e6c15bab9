// Which "synchronous" HIP calls on device memory have completed when they return, as seen by a kernel enqueued right afterwards on a
// hipStreamNonBlocking stream while other streams keep every CU busy?  (round 4: a bare hipMemset had NOT -- profiles/r04_experiments.txt item 5.)
//   hipcc --offload-arch=gfx950 -O2 null_stream_order.hip -o build/null_stream_order ;  GPU_MAX_HW_QUEUES=16 build/null_stream_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void hog(unsigned long long *sink, long long spin) {
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1ull);
}
__global__ void check(const int *word, int want, unsigned long long *bad) {
    if (*word != want) atomicAdd(bad, 1ull);
}
__global__ void put(int *word, int v) { *word = v; }
int main() {
    const int NS = 8, ROUNDS = 200;
    std::vector<hipStream_t> st(NS);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStream_t probe;
    (void)hipStreamCreateWithFlags(&probe, hipStreamNonBlocking);
    unsigned long long *bad, *sink;
    int *word;
    (void)hipMalloc(&bad, 32); (void)hipMalloc(&sink, 8); (void)hipMalloc(&word, 4);
    (void)hipMemset(bad, 0, 32); (void)hipMemset(sink, 0, 8);
    (void)hipDeviceSynchronize();
    const char *names[3] = {"hipMemset (value 0 over a non-zero word)", "hipMemcpy host -> device (pageable, 4 bytes)", "hipMemsetAsync(null) + hipStreamSynchronize(null)"};
    for (int mode = 0; mode < 3; mode++) {
        for (int r = 1; r <= ROUNDS; r++) {
            for (int s = 0; s < NS; s++) hipLaunchKernelGGL(hog, dim3(512), dim3(256), 0, st[s], sink, 200000);      // the chip stays full
            int want = r;
            if (mode == 1) { (void)hipMemcpy(word, &want, 4, hipMemcpyHostToDevice); }
            else {
                hipLaunchKernelGGL(put, dim3(1), dim3(1), 0, probe, word, 12345);
                (void)hipStreamSynchronize(probe);
                want = 0;
                if (mode == 0) (void)hipMemset(word, 0, 4);
                else { (void)hipMemsetAsync(word, 0, 4, nullptr); (void)hipStreamSynchronize(nullptr); }
            }
            hipLaunchKernelGGL(check, dim3(1), dim3(1), 0, probe, word, want, bad + mode);
            (void)hipStreamSynchronize(probe);
        }
        (void)hipDeviceSynchronize();
        unsigned long long h[4];
        (void)hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
        printf("%-52s stale in %llu of %d rounds\n", names[mode], h[mode], ROUNDS);
    }
    return 0;
}
