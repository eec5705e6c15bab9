"""Integer op codes of the compiled-circuit wire format (values fixed by the reference, src/noise/constants.py:8-29)."""
import numpy as np

OP_CNOT, OP_PREP_X, OP_PREP_Z, OP_MEAS_X, OP_MEAS_Z, OP_IDLE = 1, 2, 3, 4, 5, 6
OP_X, OP_Y, OP_Z = 10, 11, 12
OP_XX, OP_XY, OP_XZ, OP_YX, OP_YY, OP_YZ, OP_ZX, OP_ZY, OP_ZZ = 20, 21, 22, 23, 24, 25, 26, 27, 28

_GATES = ["CNOT", "PrepX", "PrepZ", "MeasX", "MeasZ", "IDLE"]
_PAULI1 = ["X", "Y", "Z"]
GATE_TO_OPCODE = {g: i + 1 for i, g in enumerate(_GATES)}
GATE_TO_OPCODE.update({p: 10 + i for i, p in enumerate(_PAULI1)})
GATE_TO_OPCODE.update({a + b: 20 + 3 * i + j for i, a in enumerate(_PAULI1) for j, b in enumerate(_PAULI1)})

# index -> op code of the 15 two-qubit Pauli faults drawn after a CNOT (order of noise/kernels.py:283-343)
TWO_QUBIT_ERROR_OPCODES = np.array([OP_X, OP_Y, OP_Z, OP_X, OP_Y, OP_Z, OP_XX, OP_YY, OP_ZZ, OP_XY, OP_YX, OP_YZ, OP_ZY, OP_XZ, OP_ZX],
                                   dtype=np.int32)
# 0 = acts on the control, 1 = on the target, 2 = on both
TWO_QUBIT_ERROR_TARGET = np.array([0] * 3 + [1] * 3 + [2] * 9, dtype=np.int32)
ERROR_LOCATION_GATES = ("MeasX", "MeasZ", "PrepX", "PrepZ", "IDLE", "CNOT")
