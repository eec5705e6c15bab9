#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/pmc_passes.sh -> profiles/pmc.json (read by bench.py) + a readable table.

Per kernel the counted launches are the LAST ones of the workload's launch sequence (warm-ups come first).  Every entry records
the kernel symbol, the sha256 of the kernel's source files at collection time (bench.py refuses a record whose sources changed) and
the workload units one launch processed, so a count per unit carries over to other batch sizes of the same kernel.
HBM bytes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of a wide coalesced read,
MI355X_MICROARCH.md "HBM") -- an upper estimate for narrow reads."""
import csv
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qldpc-branched-off_amd", "csrc")


def digest(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def regular_args(kernel_name):
    """template arguments of minsum_regular_kernel<CDEG, VDEG, DAMP, NANFREE, MC, FIXED> as strings, or None"""
    import re
    mt = re.search(r"minsum_regular_kernel<([^>]*)>", kernel_name)
    return [x.strip() for x in mt.group(1).split(",")] if mt else None


def load_pass(d):
    """{counter: [(dispatch_id, kernel_name, value), ...]} of one pass directory, in dispatch order."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for k in out:
        out[k].sort()
    return out


def main():
    src = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(src.rstrip("/"))
    with open(os.path.join(src, "workload.json")) as fh:
        wl = json.load(fh)
    counters = {}
    for p in ("sq_a", "sq_b", "sq_c", "fetch", "write", "grbm"):
        counters.update(load_pass(os.path.join(src, p)))
    cl = wl["circuit_level"]
    regular = ["minsum_regular.hip", "minsum_common.h", "minsum_f64.h", "mc_common.h"]        # (the same lists as tools/isa_mix.py RECORDED)
    specs = {   # key: (kernel-name match, counted launches, units per launch, unit, sources)
        f"cc_{wl['code']}_fixed": (lambda k: (regular_args(k) or [""] * 6)[4:6] == ["true", "true"], 2,
                                   wl["cc_fixed"]["shots_per_launch"] * wl["cc_fixed"]["max_iter"], "shot_iteration", regular),
        # reference semantics: the bit-sliced first iteration sees every shot (the full decoder only the few it lists: cc_..._early_exit_full)
        f"cc_{wl['code']}_early_exit": (lambda k: "mc_first_kernel" in k, 2, wl["cc_early_exit"]["shots_per_launch"], "shot", ["mc_first.hip", "mc_common.h"]),
        f"cc_{wl['code']}_early_exit_full": (lambda k: (regular_args(k) or [""] * 6)[4:6] == ["true", "false"], 2,
                                             wl["cc_early_exit"]["shots_per_launch"], "shot", regular),
        f"{wl['circuit']}_bp": (lambda k: "minsum_wg2_kernel" in k, 2, (cl["iters_z"] + cl["iters_x"]) / 2.0, "decode_iteration",
                                ["minsum_wg2.hip", "minsum_common.h"]),
        f"{wl['circuit']}_osd": (lambda k: "osd0_gj_kernel" in k, 2, (cl["osd_z"] + cl["osd_x"]) / 2.0, "osd_shot",
                                 ["osd_gj.hip", "osd_gj.h", "osd_common.h"]),
    }
    entries, lines = {}, []
    for key, (match, nl, units, unit, sources) in specs.items():
        e = {"unit": unit, "units_per_launch": units, "sources": sources, "source_digest": digest(sources), "counted_launches": nl,
             "source": f"profiles/{tag}_pmc.txt (rocprofv3 --pmc, tools/pmc_passes.sh, collected {time.strftime('%Y-%m-%d')})"}
        for cname, rows in counters.items():
            vals = [(did, k, v) for did, k, v in rows if match(k)]
            if not vals:
                continue
            e["kernel"] = vals[-1][1]
            last = vals[-nl:]
            e[cname] = sum(v for _, _, v in last) / len(last)
        if "SQ_INSTS_VALU" not in e:
            continue
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in e:
                    e[c + "_frac"] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        types = {c[len("SQ_INSTS_VALU_"):]: round(e[c] / units, 3) for c in e if c.startswith("SQ_INSTS_VALU_")}
        if types:
            e["valu_types"] = types                   # wave-instructions per unit by the hardware's own type counters (cross-check of the static mix)
        # algorithmic floors in SIMD issue-cycles per unit (lane-operations x cycles of the cheapest instruction / 64)
        if unit == "decode_iteration":
            fl = 0.0
            for dm in cl["dims"]:
                # per edge: q = clip(v - r) 3 f64 (12), sign XOR (2), compare + 2 selects + sign insert (16), parity XOR (2) = 32; its share of the
                # min1 / min2 network (14 / 6 f64 per edge), the variable-side addition (4); per nonempty row two multiplications (8); per column + prior (4)
                fl += dm["nnz"] * (32.0 + 14.0 / 6.0 * 4.0 + 4.0) + dm["nonempty_rows"] * 8.0 + dm["n"] * 4.0
            e["floor_issue_cycles_per_unit"] = round(fl / len(cl["dims"]) / 64.0, 1)
        if unit == "osd_shot" and "osd_counts" in wl:
            oc = wl["osd_counts"]           # per OSD shot, from the diagnostic build (tools/kbench_circuit.py --timers): pivots, columns through phase 1 / 2, touched (row, operation) pairs
            mw = oc["mw"]
            # word = 64 bits = two 32-bit XORs (2 x 2 cycles).  sort: 8 radix passes x 3 cheap operations per key; column reduction: (deg - 1) word-XORs per
            # word; the block-local Gauss-Jordan (csrc/osd_gj.hip): each pivot is cleared from the 15 other columns of its block, a selected one takes mw
            # word-XORs (half of them on average); row updates: one bit test per (row, operation) pair and mw word-XORs per touched pair
            lane_cycles = (oc["n"] * 8 * 3 * 2.0 + oc["cols"] * (oc["cdeg"] - 1) * mw * 4.0 + oc["pivots"] * 7.5 * mw * 4.0 +
                           oc["pivots"] * (oc["m"] + 2) * 2.0 + oc["touched"] * mw * 4.0)
            e["floor_issue_cycles_per_unit"] = round(lane_cycles / 64.0, 1)
            # critical path: the pivots of a shot are found one after the other; a step is at least a compare/ballot, a scalar find-first, two lane reads,
            # a second find-first and the broadcast of the mask: ~8 dependent instructions at >= 8 cycles each
            e["floor_chain_cycles_per_unit"] = round(oc["pivots"] * 64.0, 1)
            e["osd_counts"] = oc
        entries[key] = e
        per = e["SQ_INSTS_VALU"] / units
        lines.append(f"{key:26s} {e.get('kernel', '?')[:70]:70s} VALU/launch {e['SQ_INSTS_VALU']:.4g}  per {unit} {per:.3f}  "
                     f"SALU {e.get('SQ_INSTS_SALU', 0) / units:.3f}  LDS {e.get('SQ_INSTS_LDS', 0) / units:.3f}  wait_any {e.get('SQ_WAIT_ANY_frac')}  "
                     f"wait_inst {e.get('SQ_WAIT_INST_ANY_frac')}  active {e.get('SQ_ACTIVE_INST_ANY_frac')}  lds_conflict {e.get('lds_bank_conflict_frac')}  "
                     f"HBM B/launch {e.get('hbm_bytes_per_launch')}")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "pmc.json"), "w") as fh:
        json.dump({"collected": time.strftime("%Y-%m-%d"), "tag": tag, "workload": wl, "entries": entries}, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.txt"), "w") as fh:
        fh.write(f"rocprofv3 --pmc passes of tools/pmc_workload.py (tools/pmc_passes.sh), per counted launch (average of the last launches of each kernel)\n")
        fh.write("workload: " + json.dumps(wl) + "\n\n")
        fh.write("\n".join(lines) + "\n\nraw per-launch averages:\n")
        for key, e in entries.items():
            fh.write(key + ": " + json.dumps({k: v for k, v in e.items() if k.isupper() or k.startswith("SQ_")}) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
