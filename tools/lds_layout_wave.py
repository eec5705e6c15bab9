#!/usr/bin/env python3
"""Offline LDS bank-conflict model of the wave-private min-sum kernel (csrc/minsum_wave.hip): LDS-array cycles of the four access streams
of one iteration of one wave, for a given row / column assignment and buffer layout.  Rules of MI355X_MICROARCH.md (LDS):
  ds_read_b64   2 passes of 32 lanes, 32 bank pairs: (addr / 8) % 32; a pass costs max over banks of the number of DISTINCT addresses
  ds_write_b64  4 passes of 16 lanes, 16 bank pairs: (addr / 8) % 16; the instruction costs max(6, array cycles) (VGPR->LDS transfer)
  ds_read_b32   2 passes of 32 lanes, 32 banks: (addr / 4) % 32
usage: python tools/lds_layout_wave.py [bb144] [--search]"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd.data import load_code  # noqa: E402


def read64(addrs):
    tot = 0
    for g0 in (0, 32):
        banks = {}
        for a in addrs[g0:g0 + 32]:
            if a is not None:
                banks.setdefault((a // 8) % 32, set()).add(a)
        tot += max([len(v) for v in banks.values()] or [0])
    return tot


def write64(addrs):
    tot = 0
    for g0 in range(0, 64, 16):
        banks = {}
        for a in addrs[g0:g0 + 16]:
            if a is not None:
                banks.setdefault((a // 8) % 16, set()).add(a)
        tot += max([len(v) for v in banks.values()] or [0])
    return max(6, tot)


class Layout:
    """cpl checks per lane; rows: 'contig' (row = l * cpl + c) or 'strided' (row = c * LPS + l); R element (row, k) at double index
    rfun(row, k); V element col at vfun(col); team t at byte offset t * team_bytes; V region at offV"""

    def __init__(self, code, cpl, vb, rows="contig", rlayout="rowmajor", rst=7, rpad=0, vpad=0, team_pad=0, vars_="philox"):
        c = load_code(code)
        self.m, self.n, self.ip, self.ix = c["m"], c["n"], c["Hx_indptr"], c["Hx_indices"]
        self.cpl, self.vb = cpl, vb
        self.LPS = (self.m + cpl - 1) // cpl
        self.SPW = 64 // self.LPS
        self.rows, self.rlayout, self.rst, self.rpad, self.vpad = rows, rlayout, rst, rpad, vpad
        m, n = self.m, self.n
        if rlayout == "rowmajor":
            self.rsize = m * rst
            self.rfun = lambda row, k: row * rst + k
        else:                                   # k-major: plane k holds the k-th message of every row
            ms = m + rpad
            self.rsize = 6 * ms
            self.rfun = lambda row, k: k * ms + row
        self.offV = self.rsize * 8
        self.team_bytes = self.offV + (n + vpad) * 8 + 16 + team_pad
        self.vars_ = vars_
        cols = [[] for _ in range(n)]
        for i in range(m):
            for k, e in enumerate(range(self.ip[i], self.ip[i + 1])):
                cols[self.ix[e]].append((i, k))
        self.cols = cols

    def row_of(self, l, c):
        r = l * self.cpl + c if self.rows == "contig" else c * self.LPS + l
        return r if r < self.m else None

    def var_of(self, l, v):
        t, w = divmod(v, 4)
        if self.vars_ == "philox":
            q = l + self.LPS * t
        else:                                   # contiguous blocks per lane
            q = l * self.vb + t
        j = 4 * q + w
        return j if (q < (self.n + 3) // 4 and j < self.n) else None

    def streams(self):
        out = {"Vgather": [], "Rwrite": [], "Rgather": [], "Vwrite": []}
        lanes = [(ln // self.LPS, ln % self.LPS) for ln in range(64)]
        for c in range(self.cpl):
            for k in range(6):
                av, aw = [], []
                for (t, l) in lanes:
                    row = self.row_of(l, c) if t < self.SPW else None
                    if row is None:
                        av.append(None); aw.append(None)
                    else:
                        av.append(t * self.team_bytes + self.offV + 8 * self.ix[self.ip[row] + k])
                        aw.append(t * self.team_bytes + 8 * self.rfun(row, k))
                out["Vgather"].append(read64(av))
                out["Rwrite"].append(write64(aw))
        for v in range(4 * self.vb):
            aw = []
            for (t, l) in lanes:
                j = self.var_of(l, v) if t < self.SPW else None
                aw.append(None if j is None else t * self.team_bytes + self.offV + 8 * j)
            out["Vwrite"].append(write64(aw))
            for d in range(3):
                ar = []
                for (t, l) in lanes:
                    j = self.var_of(l, v) if t < self.SPW else None
                    if j is None:
                        ar.append(None)
                    else:
                        row, k = self.cols[j][d]
                        ar.append(t * self.team_bytes + 8 * self.rfun(row, k))
                out["Rgather"].append(read64(ar))
        return {k: sum(v) for k, v in out.items()}

    def total(self):
        s = self.streams()
        return sum(s.values()), s


if __name__ == "__main__":
    code = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bb144"
    ideal = None
    if "--search" not in sys.argv:
        for cpl, vb in ((6, 3), (5, 3), (4, 2)):
            for rst in (6, 7):
                L = Layout(code, cpl, vb, rst=rst)
                tot, s = L.total()
                n_r, n_w = cpl * 6 + 12 * vb, cpl * 6 + 4 * vb
                print(f"{code} cpl={cpl} rst={rst} team_bytes={L.team_bytes}: LDS cycles / wave-iteration {tot}  (conflict-free {2 * n_r + 6 * n_w}) {s}")
        sys.exit(0)
    best = []
    for cpl, vb in ((6, 3), (5, 3)):
        for rows in ("contig", "strided"):
            for rl, rsts, rpads in (("rowmajor", (6, 7, 8, 9), (0,)), ("kmajor", (0,), (0, 1, 2, 3, 4, 5, 6, 7, 8))):
                for rst in rsts:
                    for rpad in rpads:
                        for vpad in (0, 1, 2, 4):
                            for tp in range(0, 256, 8):
                                for vars_ in ("philox", "contig"):
                                    L = Layout(code, cpl, vb, rows=rows, rlayout=rl, rst=rst, rpad=rpad, vpad=vpad, team_pad=tp, vars_=vars_)
                                    tot, s = L.total()
                                    best.append((tot / L.SPW, tot, cpl, rows, rl, rst, rpad, vpad, tp, vars_, L.team_bytes, s))
    best.sort(key=lambda x: x[0])
    for b in best[:25]:
        print(b)
