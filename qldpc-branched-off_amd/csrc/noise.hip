// Circuit-level noise kernels with explicit random inputs (bit-exact twins of src/noise/kernels.py):
// a10 generate_noisy_circuit_jit, a11 simulate_circuit_{Z,X}_jit, a12 sparsify_syndrome_jit.
// One lane per draw/shot; the op stream is shared by all lanes (wave-uniform control flow over the base circuit),
// per-shot Pauli frames live in global memory laid out [shot][qubit].
#include "common.h"

namespace qldpc {

enum { OP_CNOT = 1, OP_PREP_X = 2, OP_PREP_Z = 3, OP_MEAS_X = 4, OP_MEAS_Z = 5, OP_IDLE = 6, OP_X = 10, OP_Y = 11, OP_Z = 12,
       OP_XX = 20, OP_XY = 21, OP_XZ = 22, OP_YX = 23, OP_YY = 24, OP_YZ = 25, OP_ZX = 26, OP_ZY = 27, OP_ZZ = 28 };   // noise/constants.py:8-29

__constant__ int c_two_qubit_op[15] = {OP_X, OP_Y, OP_Z, OP_X, OP_Y, OP_Z, OP_XX, OP_YY, OP_ZZ, OP_XY, OP_YX, OP_YZ, OP_ZY, OP_XZ, OP_ZX};

// a10 (noise/kernels.py:175-353)
__global__ void noisy_circuit_kernel(int64_t B, int64_t len, const int32_t *__restrict__ ops, const int32_t *__restrict__ q1,
                                     const int32_t *__restrict__ q2, double p, int64_t n_locs, const double *__restrict__ rv,
                                     const int32_t *__restrict__ rp, const int32_t *__restrict__ rt, int64_t cap, int32_t *__restrict__ oo,
                                     int32_t *__restrict__ o1, int32_t *__restrict__ o2, int64_t *__restrict__ out_len) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *v = rv + b * n_locs;
    const int32_t *pp = rp + b * n_locs, *tt = rt + b * n_locs;
    int32_t *O = oo + b * cap, *A = o1 + b * cap, *Bq = o2 + b * cap;
    int64_t out = 0, ri = 0;
#define EMIT(x, y, z) do { if (out < cap) { O[out] = (x); A[out] = (y); Bq[out] = (z); } out++; } while (0)
    for (int64_t i = 0; i < len; i++) {
        const int op = ops[i], a = q1[i], c = q2[i];
        if (op == OP_MEAS_X) { if (v[ri] < p) EMIT(OP_Z, a, -1); ri++; EMIT(op, a, c); }          // :210-221 (flip BEFORE measurement)
        else if (op == OP_MEAS_Z) { if (v[ri] < p) EMIT(OP_X, a, -1); ri++; EMIT(op, a, c); }     // :223-234
        else if (op == OP_PREP_X) { EMIT(op, a, c); if (v[ri] < p) EMIT(OP_Z, a, -1); ri++; }     // :236-246 (flip AFTER preparation)
        else if (op == OP_PREP_Z) { EMIT(op, a, c); if (v[ri] < p) EMIT(OP_X, a, -1); ri++; }     // :248-258
        else if (op == OP_IDLE) {                                                                   // :260-272 (IDLE itself dropped)
            if (v[ri] < p) { const int ch = pp[ri]; EMIT(ch == 0 ? OP_X : (ch == 1 ? OP_Y : OP_Z), a, -1); }
            ri++;
        } else if (op == OP_CNOT) {                                                                 // :274-344
            EMIT(op, a, c);
            if (v[ri] < p) {
                int t = tt[ri];
                if (t < 0 || t > 14) t = 14;
                if (t < 3) EMIT(c_two_qubit_op[t], a, -1);
                else if (t < 6) EMIT(c_two_qubit_op[t], c, -1);
                else EMIT(c_two_qubit_op[t], a, c);
            }
            ri++;
        } else EMIT(op, a, c);                                                                      // :346-351
    }
#undef EMIT
    out_len[b] = out;
}

// a11 (noise/kernels.py:13-91 for Z-type errors / 94-172 for X-type errors)
template <bool XSECTOR>
__global__ void frame_sim_kernel(int64_t B, int64_t cap, const int64_t *__restrict__ len, const int32_t *__restrict__ ops,
                                 const int32_t *__restrict__ q1, const int32_t *__restrict__ q2, int total_qubits, int max_syn,
                                 int8_t *__restrict__ hist, int8_t *__restrict__ state, int64_t *__restrict__ counts) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int8_t *st = state + b * total_qubits, *h = hist + b * max_syn;
    for (int q = 0; q < total_qubits; q++) st[q] = 0;
    for (int s = 0; s < max_syn; s++) h[s] = 0;
    const int32_t *O = ops + b * cap, *A = q1 + b * cap, *C = q2 + b * cap;
    const int64_t L = len[b];
    int64_t sc = 0, ec = 0;
    for (int64_t i = 0; i < L; i++) {
        const int op = O[i], a = A[i], c = C[i];
        if (!XSECTOR) {
            if (op == OP_CNOT) st[a] ^= st[c];                                   // Z propagates target -> control (:55-57)
            else if (op == OP_PREP_X) st[a] = 0;
            else if (op == OP_MEAS_X) { if (sc < max_syn) h[sc] = st[a]; sc++; }
            else if (op == OP_Z || op == OP_Y || op == OP_ZX || op == OP_YX) { ec++; st[a] ^= 1; }
            else if (op == OP_XZ || op == OP_XY) { ec++; st[c] ^= 1; }
            else if (op == OP_ZZ || op == OP_YY || op == OP_YZ || op == OP_ZY) { ec++; st[a] ^= 1; st[c] ^= 1; }
        } else {
            if (op == OP_CNOT) st[c] ^= st[a];                                   // X propagates control -> target (:136-138)
            else if (op == OP_PREP_Z) st[a] = 0;
            else if (op == OP_MEAS_Z) { if (sc < max_syn) h[sc] = st[a]; sc++; }
            else if (op == OP_X || op == OP_Y || op == OP_XZ || op == OP_YZ) { ec++; st[a] ^= 1; }
            else if (op == OP_ZX || op == OP_ZY) { ec++; st[c] ^= 1; }
            else if (op == OP_XX || op == OP_YY || op == OP_XY || op == OP_YX) { ec++; st[a] ^= 1; st[c] ^= 1; }
        }
    }
    counts[2 * b] = sc; counts[2 * b + 1] = ec;
}

// a12 (noise/kernels.py:356-380): detector = XOR with the RAW previous measurement of the same check.
// One thread per (shot, check).
__global__ void sparsify_kernel(int64_t B, int64_t stride, const int8_t *__restrict__ hist, const int64_t *__restrict__ syn_count,
                                const int32_t *__restrict__ pos, const int32_t *__restrict__ ptrs, int num_checks, int8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * num_checks) return;
    const int64_t b = t / num_checks;
    const int c = (int)(t - b * num_checks);
    const int8_t *h = hist + b * stride;
    int8_t *o = out + b * stride;
    const int64_t sc = syn_count[b];
    for (int i = ptrs[c] + 1; i < ptrs[c + 1]; i++) {
        const int cur = pos[i], prev = pos[i - 1];
        if (cur < sc && prev < sc) o[cur] = (int8_t)(h[cur] ^ h[prev]);
    }
}

}  // namespace qldpc

using namespace qldpc;

QLDPC_EXPORT int qldpc_noisy_circuit_batch(int64_t B, int64_t len, const int32_t *ops, const int32_t *q1, const int32_t *q2, double p,
                                           int64_t n_locs, const double *rv, const int32_t *rp, const int32_t *rt, int64_t cap,
                                           int32_t *out_ops, int32_t *out_q1, int32_t *out_q2, int64_t *out_len) {
    QLDPC_REQUIRE(B >= 0 && len >= 0 && n_locs >= 0 && cap >= 0, "negative size");
    QLDPC_USE_DEVICE(0);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE((len == 0 || (ops && q1 && q2)) && (n_locs == 0 || (rv && rp && rt)) && out_len && (cap == 0 || (out_ops && out_q1 && out_q2)),
                  "NULL buffer");
    int64_t locs = 0;       // every error location consumes one random triple; never read past n_locs on the device
    for (int64_t i = 0; i < len; i++) locs += (ops[i] >= OP_CNOT && ops[i] <= OP_IDLE) ? 1 : 0;
    QLDPC_REQUIRE(locs <= n_locs, "circuit has %lld error locations but only %lld random values per draw", (long long)locs, (long long)n_locs);
    DevTmp dops, dq1, dq2, drv, drp, drt, doo, do1, do2, dlen;
    if ((rc = dops.alloc(len * 4)) || (rc = dq1.alloc(len * 4)) || (rc = dq2.alloc(len * 4)) || (rc = drv.alloc(B * n_locs * 8)) ||
        (rc = drp.alloc(B * n_locs * 4)) || (rc = drt.alloc(B * n_locs * 4)) || (rc = doo.alloc(B * cap * 4)) ||
        (rc = do1.alloc(B * cap * 4)) || (rc = do2.alloc(B * cap * 4)) || (rc = dlen.alloc(B * 8)))
        return rc;
    if (len) {
        QLDPC_HIP_TRY(hipMemcpy(dops.p, ops, len * 4, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(dq1.p, q1, len * 4, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(dq2.p, q2, len * 4, hipMemcpyHostToDevice));
    }
    if (n_locs) {
        QLDPC_HIP_TRY(hipMemcpy(drv.p, rv, B * n_locs * 8, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(drp.p, rp, B * n_locs * 4, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(drt.p, rt, B * n_locs * 4, hipMemcpyHostToDevice));
    }
    if (cap) {
        QLDPC_HIP_TRY(zero_now(doo.p, B * cap * 4)); QLDPC_HIP_TRY(zero_now(do1.p, B * cap * 4)); QLDPC_HIP_TRY(zero_now(do2.p, B * cap * 4));
    }
    hipLaunchKernelGGL(noisy_circuit_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, nullptr, B, len, dops.as<int32_t>(), dq1.as<int32_t>(),
                       dq2.as<int32_t>(), p, n_locs, drv.as<double>(), drp.as<int32_t>(), drt.as<int32_t>(), cap, doo.as<int32_t>(),
                       do1.as<int32_t>(), do2.as<int32_t>(), dlen.as<int64_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(out_len, dlen.p, B * 8, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; b++)
        QLDPC_REQUIRE(out_len[b] <= cap, "output circuit of draw %lld needs %lld slots, capacity %lld", (long long)b, (long long)out_len[b], (long long)cap);
    if (cap) {
        QLDPC_HIP_TRY(hipMemcpy(out_ops, doo.p, B * cap * 4, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(out_q1, do1.p, B * cap * 4, hipMemcpyDeviceToHost));
        QLDPC_HIP_TRY(hipMemcpy(out_q2, do2.p, B * cap * 4, hipMemcpyDeviceToHost));
    }
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_frame_sim_batch(int sector_is_x, int64_t B, int64_t cap, const int64_t *len, const int32_t *ops, const int32_t *q1,
                                       const int32_t *q2, int total_qubits, int max_syn, int8_t *hist, int8_t *state, int64_t *counts) {
    QLDPC_REQUIRE(B >= 0 && cap >= 0 && total_qubits >= 0 && max_syn >= 0, "negative size");
    QLDPC_USE_DEVICE(0);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0) return QLDPC_OK;
    QLDPC_REQUIRE(len && counts && (cap == 0 || (ops && q1 && q2)) && (max_syn == 0 || hist) && (total_qubits == 0 || state), "NULL buffer");
    for (int64_t b = 0; b < B; b++) QLDPC_REQUIRE(len[b] >= 0 && len[b] <= cap, "len[%lld] out of range", (long long)b);
    // every qubit index must be addressable: a faulting kernel can take the whole node down
    for (int64_t b = 0; b < B; b++)
        for (int64_t i = 0; i < len[b]; i++) {
            const int op = ops[b * cap + i], a = q1[b * cap + i], c = q2[b * cap + i];
            QLDPC_REQUIRE(a >= 0 && a < total_qubits, "q1 out of range at op %lld", (long long)i);
            const bool two = (op == 1) || (op >= 20 && op <= 28);
            QLDPC_REQUIRE(!two || (c >= 0 && c < total_qubits), "q2 out of range at op %lld", (long long)i);
        }
    DevTmp dlen, dops, dq1, dq2, dh, ds, dc;
    if ((rc = dlen.alloc(B * 8)) || (rc = dops.alloc(B * cap * 4)) || (rc = dq1.alloc(B * cap * 4)) || (rc = dq2.alloc(B * cap * 4)) ||
        (rc = dh.alloc((size_t)B * max_syn)) || (rc = ds.alloc((size_t)B * total_qubits)) || (rc = dc.alloc(B * 16)))
        return rc;
    QLDPC_HIP_TRY(hipMemcpy(dlen.p, len, B * 8, hipMemcpyHostToDevice));
    if (cap) {
        QLDPC_HIP_TRY(hipMemcpy(dops.p, ops, B * cap * 4, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(dq1.p, q1, B * cap * 4, hipMemcpyHostToDevice));
        QLDPC_HIP_TRY(hipMemcpy(dq2.p, q2, B * cap * 4, hipMemcpyHostToDevice));
    }
    if (sector_is_x)
        hipLaunchKernelGGL(frame_sim_kernel<true>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, nullptr, B, cap, dlen.as<int64_t>(),
                           dops.as<int32_t>(), dq1.as<int32_t>(), dq2.as<int32_t>(), total_qubits, max_syn, dh.as<int8_t>(), ds.as<int8_t>(),
                           dc.as<int64_t>());
    else
        hipLaunchKernelGGL(frame_sim_kernel<false>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, nullptr, B, cap, dlen.as<int64_t>(),
                           dops.as<int32_t>(), dq1.as<int32_t>(), dq2.as<int32_t>(), total_qubits, max_syn, dh.as<int8_t>(), ds.as<int8_t>(),
                           dc.as<int64_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    if (max_syn) QLDPC_HIP_TRY(hipMemcpy(hist, dh.p, (size_t)B * max_syn, hipMemcpyDeviceToHost));
    if (total_qubits) QLDPC_HIP_TRY(hipMemcpy(state, ds.p, (size_t)B * total_qubits, hipMemcpyDeviceToHost));
    QLDPC_HIP_TRY(hipMemcpy(counts, dc.p, B * 16, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}

QLDPC_EXPORT int qldpc_sparsify_batch(int64_t B, int64_t stride, const int8_t *hist, const int64_t *syn_count, const int32_t *positions,
                                      const int32_t *ptrs, int num_checks, int8_t *out) {
    QLDPC_REQUIRE(B >= 0 && stride >= 0 && num_checks >= 0, "negative size");
    QLDPC_USE_DEVICE(0);
    int rc = QLDPC_OK; (void)rc;
    if (B == 0 || stride == 0) return QLDPC_OK;
    QLDPC_REQUIRE(hist && syn_count && out && (num_checks == 0 || (positions && ptrs)), "NULL buffer");
    const int64_t npos = num_checks ? ptrs[num_checks] : 0;
    for (int64_t b = 0; b < B; b++) QLDPC_REQUIRE(syn_count[b] >= 0 && syn_count[b] <= stride, "syn_count[%lld] out of range", (long long)b);
    for (int64_t i = 0; i < npos; i++) QLDPC_REQUIRE(positions[i] >= 0, "negative measurement position");
    DevTmp dh, dsc, dpos, dptr, dout;
    if ((rc = dh.alloc(B * stride)) || (rc = dsc.alloc(B * 8)) || (rc = dpos.alloc(npos * 4)) || (rc = dptr.alloc((num_checks + 1) * 4)) ||
        (rc = dout.alloc(B * stride)))
        return rc;
    QLDPC_HIP_TRY(hipMemcpy(dh.p, hist, B * stride, hipMemcpyHostToDevice));
    QLDPC_HIP_TRY(hipMemcpy(dout.p, hist, B * stride, hipMemcpyHostToDevice));        // result = history.copy() (:367)
    QLDPC_HIP_TRY(hipMemcpy(dsc.p, syn_count, B * 8, hipMemcpyHostToDevice));
    if (npos) QLDPC_HIP_TRY(hipMemcpy(dpos.p, positions, npos * 4, hipMemcpyHostToDevice));
    if (num_checks) QLDPC_HIP_TRY(hipMemcpy(dptr.p, ptrs, (num_checks + 1) * 4, hipMemcpyHostToDevice));
    if (num_checks)
        hipLaunchKernelGGL(sparsify_kernel, dim3((unsigned)((B * num_checks + 255) / 256)), dim3(256), 0, nullptr, B, stride, dh.as<int8_t>(),
                           dsc.as<int64_t>(), dpos.as<int32_t>(), dptr.as<int32_t>(), num_checks, dout.as<int8_t>());
    QLDPC_HIP_TRY(hipGetLastError());
    QLDPC_HIP_TRY(hipDeviceSynchronize());
    QLDPC_HIP_TRY(hipMemcpy(out, dout.p, B * stride, hipMemcpyDeviceToHost));
    return QLDPC_OK;
}
