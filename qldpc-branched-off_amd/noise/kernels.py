"""Kernel-level circuit-noise entry points with the reference's names (src/noise/kernels.py), computed by HIP kernels."""
import ctypes as C

import numpy as np

from .._lib import check, f64, i8, i32, lib, ptr


def _frame_sim(is_x, circuit_ops, circuit_q1, circuit_q2, total_qubits, max_syndromes):
    ops, q1, q2 = i32(circuit_ops), i32(circuit_q1), i32(circuit_q2)
    n = np.array([ops.size], np.int64)
    hist = np.zeros(int(max_syndromes), np.int8)
    state = np.zeros(int(total_qubits), np.int8)
    counts = np.zeros(2, np.int64)
    check(lib().qldpc_frame_sim_batch(C.c_int(is_x), C.c_int64(1), C.c_int64(ops.size), ptr(n, C.c_int64), ptr(ops, C.c_int32),
                                      ptr(q1, C.c_int32), ptr(q2, C.c_int32), C.c_int(int(total_qubits)), C.c_int(int(max_syndromes)),
                                      ptr(hist, C.c_int8), ptr(state, C.c_int8), ptr(counts, C.c_int64)))
    return hist, state, int(counts[0]), int(counts[1])


def simulate_circuit_Z_jit(circuit_ops, circuit_q1, circuit_q2, total_qubits, x_check_indices, x_check_ptrs, max_syndromes):
    """Z-error Pauli-frame propagation (reference noise/kernels.py:13-91) -> (syndrome_history, state, syn_count, err_count)."""
    return _frame_sim(0, circuit_ops, circuit_q1, circuit_q2, total_qubits, max_syndromes)


def simulate_circuit_X_jit(circuit_ops, circuit_q1, circuit_q2, total_qubits, z_check_indices, z_check_ptrs, max_syndromes):
    """X-error Pauli-frame propagation (reference noise/kernels.py:94-172)."""
    return _frame_sim(1, circuit_ops, circuit_q1, circuit_q2, total_qubits, max_syndromes)


def generate_noisy_circuit_jit(base_ops, base_q1, base_q2, error_rate, random_vals, random_paulis, random_two_qubit, out_ops, out_q1, out_q2):
    """Insert Pauli faults into the base circuit from pre-drawn randoms (reference noise/kernels.py:175-353).
    Fills out_ops/out_q1/out_q2 in place and returns the output length."""
    ops, q1, q2 = i32(base_ops), i32(base_q1), i32(base_q2)
    rv, rp, rt = f64(random_vals), i32(random_paulis), i32(random_two_qubit)
    cap = int(out_ops.size)
    oo, o1, o2 = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    n = np.zeros(1, np.int64)
    check(lib().qldpc_noisy_circuit_batch(C.c_int64(1), C.c_int64(ops.size), ptr(ops, C.c_int32), ptr(q1, C.c_int32), ptr(q2, C.c_int32),
                                          C.c_double(float(error_rate)), C.c_int64(rv.size), ptr(rv, C.c_double), ptr(rp, C.c_int32),
                                          ptr(rt, C.c_int32), C.c_int64(cap), ptr(oo, C.c_int32), ptr(o1, C.c_int32), ptr(o2, C.c_int32),
                                          ptr(n, C.c_int64)))
    L = int(n[0])
    out_ops[:L], out_q1[:L], out_q2[:L] = oo[:L], o1[:L], o2[:L]
    return L


def sparsify_syndrome_jit(syndrome_history, syn_count, check_positions, check_ptrs, num_checks):
    """Detectors = XOR of consecutive raw measurements of the same check (reference noise/kernels.py:356-380)."""
    syn_count = int(syn_count)
    h = i8(syndrome_history)[:syn_count].copy()
    if syn_count == 0:
        return h
    pos, ptrs = i32(check_positions), i32(check_ptrs)
    out = np.zeros(syn_count, np.int8)
    sc = np.array([syn_count], np.int64)
    check(lib().qldpc_sparsify_batch(C.c_int64(1), C.c_int64(syn_count), ptr(h, C.c_int8), ptr(sc, C.c_int64), ptr(pos, C.c_int32),
                                     ptr(ptrs, C.c_int32), C.c_int(int(num_checks)), ptr(out, C.c_int8)))
    return out


def extract_data_state_jit(state, data_qubit_indices):
    """Gather of the data-qubit frame (reference noise/kernels.py:383-393); a 1-D index, no arithmetic."""
    return np.asarray(state, dtype=np.int8)[np.asarray(data_qubit_indices, dtype=np.int64)]
