"""Syndrome-extraction circuit of a bivariate-bicycle code (interface of the reference's src/codes/bb_code.py).

Host-side, run-once setup.  Neighbour tables are computed arithmetically from (ell, m, monomial powers): the
component matrices are permutations x^p (row (a,b) -> column ((a+p) mod ell, b)) or y^p (-> (a, (b+p) mod m)),
so no matrix is materialised.  Output format (tuple gates, lin_order dict) matches what CompiledCircuit consumes.
"""
import numpy as np

# CNOT schedule of the 8-round cycle; None = this check type idles in that round (reference bb_code.py:152-155)
SCHEDULE_X = (None, 1, 4, 3, 5, 0, 2, None)
SCHEDULE_Z = (3, 5, 0, 1, 2, 4, None, None)


class BBCodeCircuit:
    def __init__(self, Hx, Hz, num_cycles=12, ell=None, m=None, a_x_powers=None, a_y_powers=None, b_y_powers=None, b_x_powers=None):
        self.Hx = np.asarray(Hx, dtype=int)
        self.Hz = np.asarray(Hz, dtype=int)
        self.num_cycles = num_cycles
        self.m_checks, self.n = self.Hx.shape
        self.n2 = self.n // 2
        if self.m_checks != self.n2:
            raise AssertionError(f"Expected square blocks: m={self.m_checks}, n2={self.n2}")
        self.ell, self.m_dim = ell, m
        self.a_x_powers = [] if a_x_powers is None else list(np.atleast_1d(a_x_powers))
        self.a_y_powers = [] if a_y_powers is None else list(np.atleast_1d(a_y_powers))
        self.b_y_powers = [] if b_y_powers is None else list(np.atleast_1d(b_y_powers))
        self.b_x_powers = [] if b_x_powers is None else list(np.atleast_1d(b_x_powers))
        self.has_component_params = ell is not None and m is not None
        self._order_qubits()
        self._neighbours()
        self.schedule_X = ["idle" if d is None else d for d in SCHEDULE_X]
        self.schedule_Z = ["idle" if d is None else d for d in SCHEDULE_Z]
        self._cycle()

    # linear order: X checks, left data, right data, Z checks
    def _order_qubits(self):
        n2 = self.n2
        self.Xchecks = [("Xcheck", i) for i in range(n2)]
        self.data_qubits = [("data_left", i) for i in range(n2)] + [("data_right", i) for i in range(n2)]
        self.Zchecks = [("Zcheck", i) for i in range(n2)]
        self.lin_order = {q: k for k, q in enumerate(self.Xchecks + self.data_qubits + self.Zchecks)}
        self.total_qubits = 4 * n2

    def _shift(self, kind, power, i, transpose):
        ell, m = self.ell, self.m_dim
        a, b = divmod(i, m)
        s = -int(power) if transpose else int(power)
        return ((a + s) % ell) * m + b if kind == "x" else a * m + (b + s) % m

    def _neighbours(self):
        self.nbs = {}
        n2 = self.n2
        if self.has_component_params:
            A = [("x", p) for p in self.a_x_powers] + [("y", p) for p in self.a_y_powers]
            B = [("y", p) for p in self.b_y_powers] + [("x", p) for p in self.b_x_powers]

            def col(comps, d, i, transpose):
                return self._shift(comps[d][0], comps[d][1], i, transpose) if d < len(comps) else 0
            for i in range(n2):
                for d in range(3):
                    self.nbs[(("Xcheck", i), d)] = ("data_left", col(A, d, i, False))
                    self.nbs[(("Xcheck", i), 3 + d)] = ("data_right", col(B, d, i, False))
                    self.nbs[(("Zcheck", i), d)] = ("data_left", col(B, d, i, True))
                    self.nbs[(("Zcheck", i), 3 + d)] = ("data_right", col(A, d, i, True))
        else:
            for name, H in (("Xcheck", self.Hx), ("Zcheck", self.Hz)):
                for i in range(n2):
                    left = np.flatnonzero(H[i, :n2])[:3]
                    right = np.flatnonzero(H[i, n2:])[:3]
                    for d, j in enumerate(left):
                        self.nbs[((name, i), d)] = ("data_left", int(j))
                    for d, j in enumerate(right):
                        self.nbs[((name, i), 3 + d)] = ("data_right", int(j))

    def _cycle(self):
        ops = []
        for t in range(8):
            busy = set()
            if t == 0:
                ops += [("PrepX", q) for q in self.Xchecks]
            dx, dz = SCHEDULE_X[t], SCHEDULE_Z[t]
            if dx is not None:
                for c in self.Xchecks:
                    tgt = self.nbs[(c, dx)]
                    ops.append(("CNOT", c, tgt))
                    busy.add(tgt)
            if dz is not None:
                for c in self.Zchecks:
                    ctl = self.nbs[(c, dz)]
                    ops.append(("CNOT", ctl, c))
                    busy.add(ctl)
            ops += [("IDLE", q) for q in self.data_qubits if q not in busy]
            if t == 6:
                ops += [("MeasZ", q) for q in self.Zchecks]
            if t == 7:
                ops += [("MeasX", q) for q in self.Xchecks]
                ops += [("PrepZ", q) for q in self.Zchecks]
        self.cycle = ops

    def get_full_circuit(self):
        return self.cycle * self.num_cycles

    def get_circuit_with_final_measurements(self):
        return self.cycle * self.num_cycles, self.cycle * 2
