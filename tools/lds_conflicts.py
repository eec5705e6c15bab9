#!/usr/bin/env python3
"""Offline model of LDS bank conflicts of the regular resident kernel's four access streams (ds_read_b64/ds_write_b64:
64 banks x 4 B, a b64 access occupies an aligned bank pair, lanes are served in two groups of 32).  Used to pick the row
stride / slot padding of minsum_regular.hip.  Prints the slow-down factor per stream relative to conflict-free."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qldpc_amd  # noqa: F401,E402
from qldpc_amd.data import load_code  # noqa: E402


def cycles(addrs, group=32):
    tot = 0
    for g0 in range(0, 64, group):
        banks = {}
        for a in addrs[g0:g0 + group]:
            if a is not None:
                banks.setdefault(a % 32, set()).add(a)      # address in units of 8 bytes -> 32 pair-banks
        tot += max([len(v) for v in banks.values()] or [0])
    return tot


def analyse(tag, S, RST, slot_pad=0, vbase_pad=0):
    c = load_code(tag)
    m, n, ip, ix = c["m"], c["n"], c["Hx_indptr"], c["Hx_indices"]
    TS = max(m, (n + 1) // 2)
    cols = [[] for _ in range(n)]
    for i in range(m):
        for k, e in enumerate(range(ip[i], ip[i + 1])):
            cols[ix[e]].append((i, k))
    Rslot, Vslot = m * RST + slot_pad, n + slot_pad
    vbase = S * Rslot + vbase_pad
    nw = (S * TS + 63) // 64
    out = {}
    for name, reps in (("Vgather", 6), ("Rgather", 6), ("Rwrite", 6), ("Vwrite", 2)):
        tot = cnt = 0
        for w in range(nw):
            for rep in range(reps):
                ad = []
                for t in range(64 * w, 64 * w + 64):
                    s, mem = divmod(t, TS)
                    a = None
                    if s < S:
                        if name == "Vgather" and mem < m:
                            a = vbase + s * Vslot + ix[ip[mem] + rep]
                        elif name == "Rwrite" and mem < m:
                            a = s * Rslot + mem * RST + rep
                        elif name == "Rgather":
                            j = mem + (rep // 3) * TS
                            if j < n:
                                i, k = cols[j][rep % 3]
                                a = s * Rslot + i * RST + k
                        elif name == "Vwrite":
                            j = mem + rep * TS
                            if j < n:
                                a = vbase + s * Vslot + j
                    ad.append(a)
                tot += cycles(ad)
                cnt += 1
        out[name] = round(tot / cnt / 2, 2)
    return out


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "bb144"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    for RST in (6, 7, 9):
        for sp in (0, 1, 3, 5):
            print("RST", RST, "slot_pad", sp, analyse(tag, S, RST, slot_pad=sp))
