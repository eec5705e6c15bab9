"""``run_trial_fast`` with the reference's signature and RNG draw order (src/noise/simulation.py:21-107)."""
import numpy as np

from .compiled import CompiledCircuit  # noqa: F401
from .kernels import (extract_data_state_jit, generate_noisy_circuit_jit, simulate_circuit_X_jit, simulate_circuit_Z_jit,
                      sparsify_syndrome_jit)


def run_trial_with_randoms(compiled, error_rate, Lx, Lz, random_vals, random_paulis, random_two_qubit):
    """One circuit-level trial from explicit random arrays -> (sparse_z, true_z, sparse_x, true_x)."""
    noisy_len = generate_noisy_circuit_jit(compiled.base_ops, compiled.base_q1, compiled.base_q2, error_rate, random_vals,
                                           random_paulis, random_two_qubit, compiled.out_ops, compiled.out_q1, compiled.out_q2)
    full_ops = np.concatenate([compiled.out_ops[:noisy_len], compiled.suffix_ops]).astype(np.int32)
    full_q1 = np.concatenate([compiled.out_q1[:noisy_len], compiled.suffix_q1]).astype(np.int32)
    full_q2 = np.concatenate([compiled.out_q2[:noisy_len], compiled.suffix_q2]).astype(np.int32)
    syn_z, state_z, cnt_z, _ = simulate_circuit_Z_jit(full_ops, full_q1, full_q2, compiled.total_qubits, compiled.x_check_indices,
                                                      compiled.x_check_ptrs, compiled.max_syndromes_x)
    true_z = (np.asarray(Lx) @ extract_data_state_jit(state_z, compiled.data_qubit_indices)) % 2
    sparse_z = sparsify_syndrome_jit(syn_z, cnt_z, compiled.x_syn_positions, compiled.x_syn_ptrs, compiled.num_x_checks)
    syn_x, state_x, cnt_x, _ = simulate_circuit_X_jit(full_ops, full_q1, full_q2, compiled.total_qubits, compiled.z_check_indices,
                                                      compiled.z_check_ptrs, compiled.max_syndromes_z)
    true_x = (np.asarray(Lz) @ extract_data_state_jit(state_x, compiled.data_qubit_indices)) % 2
    sparse_x = sparsify_syndrome_jit(syn_x, cnt_x, compiled.z_syn_positions, compiled.z_syn_ptrs, compiled.num_z_checks)
    return sparse_z, true_z.astype(np.int8), sparse_x, true_x.astype(np.int8)


def run_trial_fast(compiled, error_rate, Lx, Lz):
    """Draws the three random arrays from the legacy global ``np.random`` state in the reference's order
    (simulation.py:43-45), so a caller that seeds np.random gets the reference's trial bit for bit."""
    n_locs = compiled.num_error_locs
    random_vals = np.random.random(n_locs)
    random_paulis = np.random.randint(0, 3, n_locs, dtype=np.int32)
    random_two_qubit = np.random.randint(0, 15, n_locs, dtype=np.int32)
    return run_trial_with_randoms(compiled, error_rate, Lx, Lz, random_vals, random_paulis, random_two_qubit)
