"""Tuple circuit -> flat int32 arrays for the GPU kernels (interface of the reference's src/noise/compiled.py)."""
import numpy as np

from .constants import ERROR_LOCATION_GATES, GATE_TO_OPCODE


def circuit_to_arrays(circuit, lin_order):
    """[(gate, q[, q2]), ...] -> (ops, q1, q2) int32; q2 = -1 for one-qubit entries, unknown gates get op 0."""
    count = len(circuit)
    ops = np.zeros(count, dtype=np.int32)
    q1 = np.full(count, -1, dtype=np.int32)
    q2 = np.full(count, -1, dtype=np.int32)
    for pos, gate in enumerate(circuit):
        ops[pos] = GATE_TO_OPCODE.get(gate[0], 0)
        if len(gate) > 1 and gate[1] is not None:
            q1[pos] = lin_order[gate[1]]
        if len(gate) > 2 and gate[2] is not None:
            q2[pos] = lin_order[gate[2]]
    return ops, q1, q2


def build_check_arrays(checks, lin_order):
    """CSR-style (indices, ptrs): one qubit index per check."""
    idx = np.fromiter((lin_order[c] for c in checks), dtype=np.int32, count=len(checks))
    return idx, np.arange(len(checks) + 1, dtype=np.int32)


def build_syndrome_map_arrays(checks, circuit, meas_type):
    """For each check, the running indices (among `meas_type` gates) at which it is measured, as CSR (positions, ptrs)."""
    slot = {c: k for k, c in enumerate(checks)}
    per_check = [[] for _ in checks]
    seen = 0
    for gate in circuit:
        if gate[0] != meas_type:
            continue
        k = slot.get(gate[1])
        if k is not None:
            per_check[k].append(seen)
        seen += 1
    ptrs = np.zeros(len(checks) + 1, dtype=np.int32)
    ptrs[1:] = np.cumsum([len(p) for p in per_check])
    flat = [x for p in per_check for x in p]
    return np.array(flat, dtype=np.int32), ptrs


def count_error_locations(circuit):
    return sum(1 for gate in circuit if gate[0] in ERROR_LOCATION_GATES)


class CompiledCircuit:
    """Array form of (noisy base circuit, noiseless suffix) + the lookup tables the noise kernels need.
    Same constructor and attributes as the reference class (src/noise/compiled.py:116-173)."""

    def __init__(self, base_circuit, noiseless_suffix, lin_order, data_qubits, Xchecks, Zchecks):
        self.lin_order = lin_order
        self.total_qubits = len(lin_order)
        self.base_ops, self.base_q1, self.base_q2 = circuit_to_arrays(base_circuit, lin_order)
        self.suffix_ops, self.suffix_q1, self.suffix_q2 = circuit_to_arrays(noiseless_suffix, lin_order)
        whole = list(base_circuit) + list(noiseless_suffix)
        self.x_syn_positions, self.x_syn_ptrs = build_syndrome_map_arrays(Xchecks, whole, "MeasX")
        self.z_syn_positions, self.z_syn_ptrs = build_syndrome_map_arrays(Zchecks, whole, "MeasZ")
        self.x_check_indices, self.x_check_ptrs = build_check_arrays(Xchecks, lin_order)
        self.z_check_indices, self.z_check_ptrs = build_check_arrays(Zchecks, lin_order)
        self.data_qubit_indices = np.array([lin_order[q] for q in data_qubits], dtype=np.int32)
        self.num_error_locs = count_error_locations(base_circuit)
        self.max_circuit_size = len(base_circuit) + len(noiseless_suffix) + self.num_error_locs
        self.out_ops = np.empty(self.max_circuit_size, dtype=np.int32)
        self.out_q1 = np.empty(self.max_circuit_size, dtype=np.int32)
        self.out_q2 = np.empty(self.max_circuit_size, dtype=np.int32)
        self.max_syndromes_x = int(np.count_nonzero(np.concatenate([self.base_ops, self.suffix_ops]) == GATE_TO_OPCODE["MeasX"])) + 100
        self.max_syndromes_z = int(np.count_nonzero(np.concatenate([self.base_ops, self.suffix_ops]) == GATE_TO_OPCODE["MeasZ"])) + 100
        self.num_x_checks = len(Xchecks)
        self.num_z_checks = len(Zchecks)
