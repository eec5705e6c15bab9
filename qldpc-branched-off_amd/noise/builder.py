"""``build_decoding_matrices`` with the reference's signature and result dict (src/noise/builder.py:69-176).

The reference simulates every single Z-type / X-type fault of the noisy circuit one by one on a process pool and merges
faults with identical (detector, logical) signatures into one column, in first-seen order, summing their probabilities.
Here ONE batched GPU frame simulation (qldpc_circuit_fault_signatures) yields the signature of a flip on each qubit slot
of every location; the per-gate fault list, the merge order and the probability sums follow the reference exactly, so the
result equals the reference's cached matrices (matrix_cache/*.npz) column for column.
"""
import numpy as np

from .. import _lib
from .compiled import CompiledCircuit
from .constants import GATE_TO_OPCODE

_OP = GATE_TO_OPCODE


def _sector(compiled, Lx, Lz, is_x, error_rate, num_syndrome_bits, k):
    ptr, idx, logmask = _lib.circuit_fault_signatures(compiled, Lx, Lz, is_x)
    ops = np.asarray(compiled.base_ops)
    meas, prep = (_OP["MeasZ"], _OP["PrepZ"]) if is_x else (_OP["MeasX"], _OP["PrepX"])

    def sig(e):
        return frozenset(int(v) for v in idx[ptr[e]:ptr[e + 1]]), int(logmask[e])

    columns = {}                    # signature -> probability terms in fault-enumeration order (dict keeps first-seen order)
    for i, op in enumerate(ops):                         # fault list of builder.py:85-102 / 132-149
        if op == meas or op == prep:
            faults, pr = [sig(2 * i)], error_rate
        elif op == _OP["IDLE"]:
            faults, pr = [sig(2 * i)], error_rate * 2 / 3
        elif op == _OP["CNOT"]:
            a, b = sig(2 * i), sig(2 * i + 1)
            faults, pr = [a, b, (a[0] ^ b[0], a[1] ^ b[1])], error_rate * 4 / 15      # control, target, both (linearity)
        else:
            continue
        for f in faults:
            columns.setdefault(f, []).append(pr)
    ncol = len(columns)
    H_full = np.zeros((num_syndrome_bits + k, ncol), dtype=int)
    probs = np.zeros(ncol)
    for c, ((dets, lm), terms) in enumerate(columns.items()):
        if dets:
            H_full[sorted(dets), c] = 1
        for r in range(k):
            if (lm >> r) & 1:
                H_full[num_syndrome_bits + r, c] = 1
        probs[c] = sum(terms)                             # builder.py:125 / 164: Python sum, in enumeration order
    return H_full, probs


def build_decoding_matrices(circuit_builder, Lx, Lz, error_rate, verbose=True, num_workers=None):
    """Decoding matrices of a circuit-level noise model -> dict with the reference's keys (builder.py:165-176)."""
    cb = circuit_builder
    k = np.asarray(Lx).shape[0]
    num_syndrome_bits = cb.n2 * (cb.num_cycles + 2)
    compiled = CompiledCircuit(base_circuit=cb.get_full_circuit(), noiseless_suffix=cb.cycle * 2, lin_order=cb.lin_order,
                               data_qubits=cb.data_qubits, Xchecks=cb.Xchecks, Zchecks=cb.Zchecks)
    if verbose:
        print("Building Z-error decoding matrix...")
    HZ_full, probsZ = _sector(compiled, Lx, Lz, False, error_rate, num_syndrome_bits, k)
    if verbose:
        print("Building X-error decoding matrix...")
    HX_full, probsX = _sector(compiled, Lx, Lz, True, error_rate, num_syndrome_bits, k)
    return {
        "HdecZ": HZ_full[:num_syndrome_bits], "HdecX": HX_full[:num_syndrome_bits],
        "channel_probsZ": probsZ, "channel_probsX": probsX,
        "HZ_full": HZ_full, "HX_full": HX_full,
        "first_logical_rowZ": num_syndrome_bits, "first_logical_rowX": num_syndrome_bits,
        "num_cycles": cb.num_cycles, "k": k,
    }
