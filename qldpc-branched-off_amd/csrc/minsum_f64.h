// f64 building blocks shared by the register-resident min-sum kernels (minsum_regular.hip, minsum_wave.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace qldpc {

// rarely used pointers of a fused Monte-Carlo launch live in device memory to keep scalar registers free
struct RegCold {
    unsigned long long *tally;
    int32_t *fail_count, *fail_list; int8_t *f_synd, *f_err, *f_hard; double *f_llr;
    unsigned long long *clk;        // QLDPC_FLAG_CLOCK_PROBE: per-workgroup (delta s_memtime, delta s_memrealtime), else NULL
};

// v_min_f64 / v_max_f64 without the canonicalising v_max the compiler adds around fmin()/fmax() (operands here are
// results of arithmetic or LDS loads of such results; for NaN operands the instructions return the other operand).
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin_abs(double a, double b) { double r; asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax_abs(double a, double b) { double r; asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The two smallest magnitudes (with multiplicity) of x[0..D): pairs -> (min,max), then merge (lo,hi) sets:
// lo = min(l1,l2), hi = min(max(l1,l2), min(h1,h2)).  min1/min2 of kernels.py:301-306 are exactly these values.
template <int D>
__device__ __forceinline__ void two_smallest_abs(const double *x, double &lo, double &hi) {
    static_assert(D % 2 == 0 && D >= 2, "even degree");
    lo = vmin_abs(x[0], x[1]);
    hi = vmax_abs(x[0], x[1]);
#pragma unroll
    for (int k = 2; k < D; k += 2) {
        const double l2 = vmin_abs(x[k], x[k + 1]), h2 = vmax_abs(x[k], x[k + 1]);
        const double nlo = vmin(lo, l2);
        hi = vmin(vmax(lo, l2), vmin(hi, h2));
        lo = nlo;
    }
}

__device__ __forceinline__ double min_tree6(const double *a) { return fmin(fmin(fmin(a[0], a[1]), fmin(a[2], a[3])), fmin(a[4], a[5])); }
template <int D> __device__ __forceinline__ double min_tree(const double *a) {
    double r = a[0];
#pragma unroll
    for (int k = 1; k < D; k++) r = fmin(r, a[k]);
    return r;
}
template <> __device__ __forceinline__ double min_tree<6>(const double *a) { return min_tree6(a); }


}  // namespace qldpc
