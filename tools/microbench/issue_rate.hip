// VALU / LDS issue-rate microbenchmark for gfx950 (MI355X): cycles per wave-instruction and per SIMD for the instruction classes the
// qLDPC kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD.  bench.py's roofline weights a kernel's instruction mix with these
// (profiles/r03_issue_rate.txt is the output on the GPU box; MI355X_MICROARCH.md states 2 cycles for 32-bit wave64 VALU with more
// than one wave per SIMD, 4 for one wave alone, f64 at half rate).
//   hipcc -O2 --offload-arch=gfx950 issue_rate.hip -o build/issue_rate && ./build/issue_rate
// Each kernel runs REPS x 32 independent instructions of one class per wave (8 accumulators, 4 rounds), stamps s_memtime around the
// loop; cycles per SIMD-instruction = max-over-waves(delta) / (waves_per_simd x REPS x 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int REPS = 8192;

#define R4(x) x x x x
// 8 accumulators a0..a7 (32-bit) / d0..d7 (64-bit); operand b / e is loop-invariant
#define OP32(ins) \
    ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" \
    ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7\n"
#define BODY32(name, ins)                                                                                              \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned b = seed | 1;                                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R4(OP32(ins)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                      \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;                                           \
    }
#define BODY64(name, ins)                                                                                              \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        double a0 = 1.0 + seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        double b = 1.0000001 + seed * 1e-9;                                                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R4(OP32(ins)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345) out[0] = 0;                                              \
    }
// compares write an SGPR pair (vcc would serialise: use explicit s[..] destinations through 8 different pairs is not expressible with
// constraints, so the compare result goes to vcc and the stream is 32 independent compares)
#define CMP8(ins) ins " vcc, %0, %8\n" ins " vcc, %1, %8\n" ins " vcc, %2, %8\n" ins " vcc, %3, %8\n" ins " vcc, %4, %8\n" ins " vcc, %5, %8\n" ins " vcc, %6, %8\n" ins " vcc, %7, %8\n"
#define BODYCMP(name, ins, T, init)                                                                                    \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        T a0 = init + seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        T b = a3;                                                                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R4(CMP8(ins)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                      \
    }
// v_cndmask_b32 reads vcc
#define CND8 "v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n" \
             "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
__global__ void k_cndmask_b32(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    const unsigned long long cond = __ballot((threadIdx.x + seed) & 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4(CND8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(cond));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
// mixed stream in the proportion of the regular kernel's loop: 3 f64 : 2 b32
#define MIX8 "v_add_f64 %0, %8, %0\n v_xor_b32 %4, %9, %4\n v_min_f64 %1, %8, %1\n v_add_f64 %2, %8, %2\n v_xor_b32 %5, %9, %5\n" \
             "v_max_f64 %3, %8, %3\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %9, %7\n"
__global__ void k_mix_3f64_2b32(unsigned long long *out, unsigned seed) {
    double d0 = 1.0 + seed + threadIdx.x, d1 = d0 * 3, d2 = d0 * 5, d3 = d0 * 7, e = 1.0000001;
    unsigned a4 = seed + threadIdx.x, a5 = a4 * 3, a6 = a4 * 5, a7 = a4 * 7, b = seed | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4(MIX8) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e), "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (d0 + d1 + d2 + d3 == 0.12345 || (a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
// LDS: 32 ds_read_b64 / ds_write_b64 per round on conflict-free addresses (lane * 8), results consumed once per round
__global__ void k_ds_read_b64(unsigned long long *out, unsigned seed) {
    __shared__ double buf[1024 * 2];
    buf[threadIdx.x] = seed; buf[threadIdx.x + 1024] = seed;
    __syncthreads();
    const unsigned addr = threadIdx.x * 8;
    double s = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++) {
        double x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8 offset:8192\n"
                     "ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:8192\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8 offset:8192\n"
                     "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8 offset:8192\n"
                     "ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:8192\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8 offset:8192\n"
                     "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8 offset:8192\n"
                     "ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:8192\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8 offset:8192\n"
                     "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8 offset:8192\n"
                     "ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:8192\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8 offset:8192\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(addr) : "memory");
        s += x0 + x7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (s == 0.12345) out[0] = 0;
}
__global__ void k_ds_write_b64(unsigned long long *out, unsigned seed) {
    __shared__ double buf[1024 * 2];
    const unsigned addr = threadIdx.x * 8;
    double v = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++) {
        asm volatile(R4(R4("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:8192\n")) "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (buf[(threadIdx.x * 7) & 1023] == 0.12345) out[0] = 0;
}

#define OP33(ins) \
    ins " %0, %8, %0, %1\n" ins " %1, %8, %1, %2\n" ins " %2, %8, %2, %3\n" ins " %3, %8, %3, %4\n" \
    ins " %4, %8, %4, %5\n" ins " %5, %8, %5, %6\n" ins " %6, %8, %6, %7\n" ins " %7, %8, %7, %0\n"
#define BODY32_3(name, ins)                                                                                            \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned b = (seed | 1) & 15;                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R4(OP33(ins)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                      \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;                                           \
    }
// one-source ops: dst = op(src)
#define OP31(ins) \
    ins " %0, %1\n" ins " %1, %2\n" ins " %2, %3\n" ins " %3, %4\n" ins " %4, %5\n" ins " %5, %6\n" ins " %6, %7\n" ins " %7, %8\n"
#define BODY1(name, ins, T)                                                                                            \
    __global__ void k_##name(unsigned long long *out, unsigned seed) {                                                 \
        T a0 = (T)(seed + threadIdx.x), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        T b = (T)(seed | 1);                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < REPS; i++)                                                                                 \
            asm volatile(R4(OP31(ins)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                      \
        if (a0 == (T)0x12345 && a7 == (T)1) out[0] = 0;                                                                \
    }
// v_cndmask_b32 in its VOP2 form (condition in vcc)
#define CNDV8 "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n" \
              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
__global__ void k_cndmask_vcc(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a3) : "vcc");
    for (int i = 0; i < REPS; i++)
        asm volatile(R4(CNDV8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
// lane-crossing moves: DPP quad permutation and ds_bpermute_b32
__global__ void k_mov_dpp(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
// 64-bit multiply-add (the Philox rounds compile to it) and the compare -> select pair the decoders are full of
__global__ void k_mad_u64_u32(unsigned long long *out, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = seed | 1, c = 0xD2511F53u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                        "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345ull) out[0] = 0;
}
// pairs: v_cmp_lt_u32 vcc + v_cndmask_b32 (VOP2, vcc): cycles per PAIR
__global__ void k_cmp_cndmask_pair(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
// one f64 compare feeding two selects (the min-sum message select): cycles per TRIPLE
__global__ void k_cmpf64_2cndmask(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, b = seed | 1;
    double d0 = 1.0 + seed, d1 = 2.0 + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("v_cmp_eq_f64 vcc, %6, %7\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
                        "v_cmp_eq_f64 vcc, %7, %6\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cmp_eq_f64 vcc, %6, %6\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(d0), "v"(d1), "v"(b) : "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5) == 0x12345u) out[0] = 0;
}
__global__ void k_ds_bpermute(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned addr = ((threadIdx.x * 13 + 5) & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("ds_bpermute_b32 %0, %8, %1\n ds_bpermute_b32 %1, %8, %2\n ds_bpermute_b32 %2, %8, %3\n ds_bpermute_b32 %3, %8, %4\n"
                        "ds_bpermute_b32 %4, %8, %5\n ds_bpermute_b32 %5, %8, %6\n ds_bpermute_b32 %6, %8, %7\n ds_bpermute_b32 %7, %8, %0\n s_waitcnt lgkmcnt(0)\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
__global__ void k_ds_read_b32(unsigned long long *out, unsigned seed) {
    __shared__ unsigned buf[1024 * 2];
    buf[threadIdx.x] = seed; buf[threadIdx.x + 1024] = seed;
    __syncthreads();
    const unsigned addr = threadIdx.x * 4;
    unsigned s = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++) {
        unsigned x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile(R4("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8 offset:4096\n"
                        "ds_read_b32 %4, %8\n ds_read_b32 %5, %8 offset:4096\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8 offset:4096\n")
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7) : "v"(addr) : "memory");
        s += x0 + x7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (s == 0x12345u) out[0] = 0;
}
__global__ void k_ds_write2_b64(unsigned long long *out, unsigned seed) {
    __shared__ double buf[1024 * 3];
    const unsigned addr = threadIdx.x * 16;
    double v = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4(R4("ds_write2_b64 %0, %1, %1 offset1:1\n ds_write2_b64 %0, %1, %1 offset0:2 offset1:3\n")) "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (buf[(threadIdx.x * 7) & 1023] == 0.12345) out[0] = 0;
}
BODY32(xor_b32, "v_xor_b32")
BODY32(or_b32, "v_or_b32")
BODY32(sub_u32, "v_sub_u32")
BODY32(lshrrev_b32, "v_lshrrev_b32")
BODY32(ashrrev_i32, "v_ashrrev_i32")
BODY32(min_u32, "v_min_u32")
BODY32(mul_f32, "v_mul_f32")
BODY32(mul_u32_u24, "v_mul_u32_u24")
BODY32_3(and_or_b32, "v_and_or_b32")
BODY32_3(or3_b32, "v_or3_b32")
BODY32_3(add3_u32, "v_add3_u32")
BODY32_3(lshl_add_u32, "v_lshl_add_u32")
BODY32_3(bfe_u32, "v_bfe_u32")
BODY32_3(alignbit_b32, "v_alignbit_b32")
BODY32_3(perm_b32, "v_perm_b32")
BODY32_3(fma_f32, "v_fma_f32")
BODY32_3(xad_u32, "v_xad_u32")
BODY1(mov_b32, "v_mov_b32", unsigned)
BODY1(mov_b64, "v_mov_b64", double)
BODY1(not_b32, "v_not_b32", unsigned)
BODY1(cvt_f32_u32, "v_cvt_f32_u32", unsigned)
BODY1(rcp_f32, "v_rcp_f32", float)
BODY32(add_u32, "v_add_u32")
BODY32(and_b32, "v_and_b32")
BODY32(lshlrev_b32, "v_lshlrev_b32")
BODY32(mul_lo_u32, "v_mul_lo_u32")
BODY32(mul_hi_u32, "v_mul_hi_u32")
BODY32(add_f32, "v_add_f32")
BODY32(min_f32, "v_min_f32")
BODY64(add_f64, "v_add_f64")
BODY64(min_f64, "v_min_f64")
BODY64(max_f64, "v_max_f64")
BODY64(mul_f64, "v_mul_f64")
BODYCMP(cmp_lt_f64, "v_cmp_lt_f64", double, 1.0)
BODYCMP(cmp_eq_f64, "v_cmp_eq_f64", double, 1.0)
BODYCMP(cmp_lt_u32, "v_cmp_lt_u32", unsigned, 1u)
BODYCMP(cmp_eq_u64, "v_cmp_eq_u64", unsigned long long, 1ull)
__global__ void k_bfi_b32(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++)
        asm volatile(R4("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                        "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 0;
}
__global__ void k_readlane(unsigned long long *out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x;
    unsigned s0, s1, s2, s3, s4, s5, s6, s7, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++) {
        asm volatile(R4("v_readlane_b32 %0, %8, 1\n v_readlane_b32 %1, %8, 2\n v_readlane_b32 %2, %8, 3\n v_readlane_b32 %3, %8, 4\n"
                        "v_readlane_b32 %4, %8, 5\n v_readlane_b32 %5, %8, 6\n v_readlane_b32 %6, %8, 7\n v_readlane_b32 %7, %8, 8\n")
                     : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(a0));
        acc += s0 ^ s7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (acc == 0x12345u) out[0] = 0;
}


__global__ void k_clock(unsigned long long *out) {           // shader clock: s_memtime ticks per 100 MHz s_memrealtime tick, under a VALU load
    unsigned a = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 200000; i++) asm volatile("v_xor_b32 %0, 1, %0\n v_add_u32 %0, 3, %0" : "+v"(a));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a == 0x12345u) out[2] = 0;
}

struct K { const char *name; void (*fn)(unsigned long long *, unsigned); int per_rep; };      // per_rep: instructions per loop trip
#define E(n) {#n, k_##n, 32}
#define EN(n, c) {#n, k_##n, c}
static const K kernels[] = {E(xor_b32), E(or_b32), E(sub_u32), E(mov_b32), E(not_b32), E(lshrrev_b32), E(ashrrev_i32), E(min_u32), E(mul_f32), E(mul_u32_u24),
                            E(and_or_b32), E(or3_b32), E(add3_u32), E(lshl_add_u32), E(bfe_u32), E(alignbit_b32), E(perm_b32), E(fma_f32), E(xad_u32),
                            E(cndmask_vcc), E(cmp_cndmask_pair), EN(cmpf64_2cndmask, 36), E(mad_u64_u32), E(mov_b64), E(cvt_f32_u32), E(rcp_f32), E(mov_dpp), E(ds_bpermute), E(ds_read_b32), E(ds_write2_b64), E(add_u32), E(and_b32), E(lshlrev_b32), E(cndmask_b32), E(bfi_b32), E(mul_lo_u32), E(mul_hi_u32), E(add_f32), E(min_f32),
                            E(add_f64), E(min_f64), E(max_f64), E(mul_f64), E(cmp_lt_f64), E(cmp_eq_f64), E(cmp_lt_u32), E(cmp_eq_u64),
                            E(mix_3f64_2b32), E(readlane), E(ds_read_b64), E(ds_write_b64)};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * cus * 32 * 2));
    std::vector<unsigned long long> h(cus * 32 * 2);
    hipLaunchKernelGGL(k_clock, dim3(cus), dim3(1024), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost));
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("device %s, %d CUs, nominal clock %d kHz, measured shader clock %.1f MHz (s_memtime / s_memrealtime under a VALU load)\n", prop.name, cus, prop.clockRate, mhz);
    printf("columns: waves per SIMD = 1 / 2 / 4 / 8;  first number = cycles per wave-instruction PER SIMD from the WALL time of a full-chip launch\n"
           "(hipEvents, best of 3; clock as measured above), in brackets = the same from the slowest wave's own s_memtime span\n");
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (const K &k : kernels) {
        printf("%-16s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            // wps <= 4: one block per CU (a block's waves go round-robin over the 4 SIMDs); 8: two 1024-thread blocks per CU
            const int threads = 64 * 4 * (wps > 4 ? 4 : wps);
            const int blocks = cus * (wps > 4 ? 2 : 1);
            // dynamic LDS the kernels never touch pins the placement: 100 KB -> one block per CU, 64 KB -> exactly two
            const size_t dyn_lds = (wps > 4 ? 64 : 100) * 1024;
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
            double best_ms = 1e30, best_span = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), dyn_lds, 0, d, (unsigned)rep);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const int nw = blocks * threads / 64;
                CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
                best_ms = std::min(best_ms, (double)ms);
                best_span = std::min(best_span, (double)*std::max_element(h.begin(), h.begin() + nw));
            }
            const double n_inst = (double)REPS * k.per_rep * wps;
            printf("  %6.2f [%6.2f]", best_ms * 1e-3 * mhz * 1e6 / n_inst, best_span / n_inst);
        }
        printf("\n");
    }
    CHECK(hipFree(d));
    return 0;
}
