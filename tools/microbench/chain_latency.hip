// Latency of DEPENDENT instruction chains on one wave of gfx950 (MI355X): what a pivot step of the OSD-0 chain (csrc/osd_gj.hip) is made of.
// One wave per workgroup, one workgroup per CU (nothing else competes for the SIMD); each kernel repeats a small dependent pattern REPS x 8 times
// between two s_memtime stamps; cycles per pattern = delta / (REPS x 8).  The patterns:
//   valu      v_xor -> v_xor                                     (vector -> vector)
//   valu64    v_and x2 -> v_cmp_ne_u64 -> v_cndmask              (vector, through vcc)
//   v2s       v_cmp_ne_u32 (sgpr pair) -> s_and_b64 -> v_cndmask (vector -> scalar -> vector)
//   ballot    v_cmp -> s_ff1_i32_b64 -> v_readlane -> s_add -> v_xor (the pivot search: vector -> scalar -> lane read -> scalar -> vector)
//   bitget    v_and -> v_cmp_ne_u32 -> s_lshr_b64 -> v_bfe_i32 -> v_and -> v_xor   (the per-column update of a pivot step)
//   branch    s_cmp -> s_cbranch (taken every time)
//   hipcc -O2 --offload-arch=gfx950 chain_latency.hip -o build/chain_latency && ./build/chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int REPS = 4096;
#define R8(x) x x x x x x x x

#define KERNEL(name, asmtext, ...)                                                                                     \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        unsigned a = seed + threadIdx.x, b = seed | 1u, c = (threadIdx.x & 3u);                                        \
        unsigned long long s = 0x1111111111111111ull;                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R8(asmtext) : "+v"(a), "+s"(s) : "v"(b), "v"(c) : __VA_ARGS__);                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                               \
        if (a == 0x12345u && s == 7) out[0] = 0;                                                                       \
    }

// %0 = a (vgpr, carried), %1 = s (sgpr pair, carried), %2 = b, %3 = c (vgpr constants)
KERNEL(valu, "v_xor_b32 %0, %0, %2\n v_xor_b32 %0, %0, %3\n", "memory")
KERNEL(valu64, "v_and_b32 v20, %0, %2\n v_and_b32 v21, %0, %3\n v_cmp_ne_u64 vcc, 0, v[20:21]\n v_cndmask_b32 %0, %0, %2, vcc\n", "v20", "v21", "vcc")
KERNEL(v2s, "v_cmp_ne_u32 s[20:21], 0, %0\n s_and_b64 s[20:21], s[20:21], %1\n v_cndmask_b32 %0, %2, %0, s[20:21]\n", "s20", "s21")
KERNEL(ballot, "v_cmp_ne_u32 s[20:21], 0, %0\n s_or_b64 s[20:21], s[20:21], %1\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n s_add_u32 s23, s23, 1\n v_xor_b32 %0, s23, %0\n",
       "s20", "s21", "s22", "s23", "scc")
KERNEL(bitget, "v_and_b32 v20, %0, %2\n v_cmp_ne_u32 s[20:21], 0, v20\n s_lshr_b64 s[20:21], s[20:21], 4\n v_bfe_i32 v20, s20, %3, 1\n v_and_b32 v20, v20, %2\n v_xor_b32 %0, %0, v20\n",
       "v20", "s20", "s21")
KERNEL(branch, "s_cmp_lg_u64 %1, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n", "scc")
// a whole pivot step as the compiler emits it (osd_gj.hip), one register of columns: search + mask + update
KERNEL(step, "v_and_b32 v20, %0, %2\n v_and_b32 v21, %0, %3\n v_cmp_ne_u64 vcc, 0, v[20:21]\n s_and_b64 s[20:21], vcc, %1\n s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc0 1f\n"
            "s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s24, v20, s22\n v_readlane_b32 s25, v21, s22\n s_ff1_i32_b64 s23, s[24:25]\n s_lshr_b32 s26, s22, 2\n s_lshl_b64 s[24:25], 1, s23\n"
            "v_mov_b32 v22, s25\n v_cmp_eq_u32 vcc, s26, %3\n v_cndmask_b32 v23, 0, v22, vcc\n v_mov_b32 v22, s24\n v_cndmask_b32 v24, 0, v22, vcc\n"
            "v_and_b32 v20, v23, %0\n v_and_b32 v21, v24, %2\n v_cmp_ne_u64 vcc, 0, v[20:21]\n s_and_b32 s22, s22, 60\n s_lshr_b64 s[24:25], vcc, s22\n v_bfe_i32 v20, s24, %3, 1\n"
            "v_and_b32 v20, v20, %2\n v_xor_b32 %0, %0, v20\n1:\n",
       "v20", "v21", "v22", "v23", "v24", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "vcc", "scc")

// ---- the pivot step of csrc/osd_gj.hip as the compiler builds it (gj_pivot_step_sl, copied), sixteen steps on synthetic columns, one wave per CU ----
#include "../../qldpc-branched-off_amd/csrc/osd_common.h"
using namespace qldpc;
struct GjBlock { unsigned long long X[4]; unsigned long long live; int nops, maxops; uint32_t depmask, pivmask; int oppv; };
template <int T>
__device__ __forceinline__ void step_sl(GjBlock &S, int lane) {
    constexpr int IT = T >> 2, GT = T & 3;
    const int g = lane & 3, w = lane >> 2;
    const unsigned long long owners = 0x1111111111111111ull << GT;
    const unsigned long long mword = S.X[IT] & S.live;
    const unsigned long long bal = __ballot(mword != 0ull) & owners;
    const bool piv = bal != 0ull;
    const int src = __builtin_ctzll(bal | (1ull << 63));
    const unsigned long long pword = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), src) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, src);
    const int pb = __builtin_ctzll(pword | (1ull << 63)), wq = src >> 2, wp = piv ? wq : 99, pp = wq * 64 + pb;
    const unsigned long long pl = (w == wp) ? (1ull << pb) : 0ull;
    const unsigned long long rm = S.X[IT] & ~pl;
    const unsigned long long rmq = quad_bcast<GT>(rm);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned long long x = S.X[i];
        const uint32_t np = (uint32_t)(__ballot((x & pl) != 0ull) >> (4 * wq));
        const int fp = __builtin_amdgcn_sbfe((int)np, g, 1);
        const unsigned long long add = sext64(fp) & rmq;
        S.X[i] = x ^ ((i == IT && g == GT) ? pl : add);
    }
    S.live &= ~pl;
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(S.oppv) : "s"(pp), "n"(T));
    S.pivmask |= piv ? (1u << T) : 0u;
    S.depmask |= piv ? 0u : (1u << T);
    S.nops += piv ? 1 : 0;
}
__global__ void k_chain(unsigned long long *out, unsigned seed) {
    const int lane = threadIdx.x;
    unsigned long long acc = 0, tsum = 0;
    unsigned long long r = 0x9E3779B97F4A7C15ull * (seed + lane + 1);
    for (int it = 0; it < 256; it++) {
        GjBlock S;
#pragma unroll
        for (int i = 0; i < 4; i++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; S.X[i] = r & (r >> 3) & (r << 5); }     // sparse-ish columns
        S.live = ~0ull; S.nops = 0; S.maxops = 16; S.depmask = 0; S.pivmask = 0; S.oppv = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define ST(TT) step_sl<TT>(S, lane);
        ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15)
#undef ST
        tsum += __builtin_amdgcn_s_memtime() - t0;
        acc ^= S.X[0] ^ S.X[1] ^ S.X[2] ^ S.X[3] ^ S.live ^ (unsigned)S.oppv ^ S.pivmask;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = tsum / (256 * 16);
    if (acc == 0x1234567ull) out[0] = 1;
}

template <class K>
static void run(const char *name, K kernel, int instrs, double mhz) {
    const int grid = 256;
    unsigned long long *d = nullptr;
    CHECK(hipMalloc(&d, grid * 8));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, d, 12345u);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[grid / 2] / ((double)REPS * 8.0);
    printf("%-8s %2d instructions per pattern: %7.1f cycles per pattern, %5.1f per instruction\n", name, instrs, cyc, cyc / instrs);
    (void)mhz;
    CHECK(hipFree(d));
}

int main() {
    printf("dependent-chain latencies, one wave alone on its CU (s_memtime ticks = shader cycles)\n");
    run("valu", k_valu, 2, 0);
    run("valu64", k_valu64, 4, 0);
    run("v2s", k_v2s, 3, 0);
    run("ballot", k_ballot, 6, 0);
    run("bitget", k_bitget, 6, 0);
    run("branch", k_branch, 3, 0);
    run("step", k_step, 25, 0);
    {
        const int grid = 256;
        unsigned long long *d = nullptr;
        CHECK(hipMalloc(&d, grid * 8));
        hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, d, 777u);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid);
        CHECK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("the compiled pivot step of csrc/osd_gj.hip (sixteen columns in four registers, straight-line form), one wave alone: %llu cycles per step\n", h[grid / 2]);
        CHECK(hipFree(d));
    }
    return 0;
}
