"""Lowering of the tuple circuit into the flat int32 tables the device kernels index.

Interface parity: `CompiledCircuit` takes the constructor arguments and exposes the attribute names of the reference
class (src/noise/compiled.py:116-173) because `run_trial_fast`, the builder and `_lib.make_circuit_desc` read them by
name.  The lowering itself is table driven: one (count, 3) int32 table per gate list, measurement bookkeeping done with
numpy on the opcode column instead of a second walk over the tuples.
"""
import numpy as np

from .constants import ERROR_LOCATION_GATES, GATE_TO_OPCODE

_NOISY_OPCODES = np.array(sorted(GATE_TO_OPCODE[name] for name in ERROR_LOCATION_GATES), dtype=np.int32)
_SYNDROME_SLACK = 100  # head-room the reference leaves on its syndrome scratch sizes


def _lower(gates, where):
    """Gate tuples -> int32 table with columns (opcode, first qubit, second qubit); absent operands are -1 and a
    gate name outside the opcode table lowers to opcode 0 (the kernels skip it)."""
    table = np.full((len(gates), 3), -1, dtype=np.int32)
    for row, gate in zip(table, gates):
        row[0] = GATE_TO_OPCODE.get(gate[0], 0)
        for col, operand in enumerate(gate[1:3], start=1):
            if operand is not None:
                row[col] = where[operand]
    return table


def circuit_to_arrays(circuit, lin_order):
    """(ops, q1, q2) contiguous int32 views of the lowered table."""
    table = _lower(circuit, lin_order)
    return tuple(np.ascontiguousarray(table[:, c]) for c in range(3))


def build_check_arrays(checks, lin_order):
    """Every stabiliser check owns exactly one ancilla, so the CSR has unit-length rows."""
    owners = np.array([lin_order[c] for c in checks], dtype=np.int32).reshape(-1)
    return owners, np.arange(owners.size + 1, dtype=np.int32)


def build_syndrome_map_arrays(checks, circuit, meas_type):
    """CSR (positions, ptrs): for check k, the ordinals -- counted among `meas_type` gates only -- of its measurements."""
    rank_of = {c: k for k, c in enumerate(checks)}
    measured = [gate[1] for gate in circuit if gate[0] == meas_type]
    owner = np.array([rank_of.get(q, -1) for q in measured], dtype=np.int64).reshape(-1)
    ordinals = np.flatnonzero(owner >= 0)
    grouped = ordinals[np.argsort(owner[ordinals], kind="stable")]
    ptrs = np.zeros(len(checks) + 1, dtype=np.int32)
    np.cumsum(np.bincount(owner[ordinals], minlength=len(checks)), out=ptrs[1:])
    return grouped.astype(np.int32), ptrs


def count_error_locations(circuit):
    """Number of gates after which the noise model may insert a fault."""
    return int(np.isin(_lower(circuit, _Anywhere())[:, 0], _NOISY_OPCODES).sum())


class _Anywhere(dict):
    """Operand lookup that accepts any qubit label (used when only the opcode column matters)."""

    def __missing__(self, key):
        return 0


class CompiledCircuit:
    """Noisy base circuit + noiseless suffix in array form, with the check / measurement lookup tables."""

    def __init__(self, base_circuit, noiseless_suffix, lin_order, data_qubits, Xchecks, Zchecks):
        self.lin_order = lin_order
        self.total_qubits = len(lin_order)

        for prefix, gates in (("base", base_circuit), ("suffix", noiseless_suffix)):
            for field, column in zip(("ops", "q1", "q2"), circuit_to_arrays(gates, lin_order)):
                setattr(self, f"{prefix}_{field}", column)

        opcodes = np.concatenate([self.base_ops, self.suffix_ops])
        every_gate = list(base_circuit) + list(noiseless_suffix)
        for tag, checks, gate_name in (("x", Xchecks, "MeasX"), ("z", Zchecks, "MeasZ")):
            positions, ptrs = build_syndrome_map_arrays(checks, every_gate, gate_name)
            owners, owner_ptrs = build_check_arrays(checks, lin_order)
            setattr(self, f"{tag}_syn_positions", positions)
            setattr(self, f"{tag}_syn_ptrs", ptrs)
            setattr(self, f"{tag}_check_indices", owners)
            setattr(self, f"{tag}_check_ptrs", owner_ptrs)
            setattr(self, f"num_{tag}_checks", len(checks))
            setattr(self, f"max_syndromes_{tag}", int((opcodes == GATE_TO_OPCODE[gate_name]).sum()) + _SYNDROME_SLACK)

        self.data_qubit_indices = np.array([lin_order[q] for q in data_qubits], dtype=np.int32)
        self.num_error_locs = int(np.isin(self.base_ops, _NOISY_OPCODES).sum())
        # worst case: every error location receives an inserted fault gate
        self.max_circuit_size = opcodes.size + self.num_error_locs
        for field in ("ops", "q1", "q2"):
            setattr(self, f"out_{field}", np.empty(self.max_circuit_size, dtype=np.int32))
